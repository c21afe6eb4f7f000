#!/bin/bash
# chained rows (SG_DENSE=1 build): parity, then same-box A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
SG_LIB_NAME=libsuggest_hip_dense.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg5 and not dense_terms" 2>&1 | tail -4
for rep in 1 2; do for lib in libsuggest_hip.so libsuggest_hip_dense.so; do for c in headline cfg3 cfg2; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config $c --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
done; done; done
for lib in libsuggest_hip.so libsuggest_hip_dense.so; do
  SG_LIB_NAME=$lib timeout 600 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" 2>&1 | grep "per step" | sed "s/^/$lib /"
  SG_LIB_NAME=$lib timeout 600 python bench.py --config cfg4 --steps 5 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib cfg4', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
done
