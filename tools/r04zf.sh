#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for lib in libsuggest_hip.so libsuggest_hip_dense.so libsuggest_hip_dnohaz.so; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config headline --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib headline', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
done
SG_PROF_LIB=libsuggest_hip_dprof.so timeout 900 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -12
