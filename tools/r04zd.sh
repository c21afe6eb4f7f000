#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for lib in libsuggest_hip_nodense.so libsuggest_hip.so libsuggest_hip_nohaz.so; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config headline --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib headline', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
done
for lib in libsuggest_hip_nodense.so libsuggest_hip.so; do
  SG_LIB_NAME=$lib timeout 1200 bash tools/pmc_run.sh r04zd_$lib --config headline --batches 1 --sub-configs none 2>&1 | grep "^search" | awk -v L=$lib '{print L, $2, $5}'
done
