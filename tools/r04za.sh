#!/bin/bash
# dense rows: parity (whole suite), then same-box A/B against the build without them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04za_pytest.log 2>&1; tail -12 $O/r04za_pytest.log
for rep in 1 2; do for lib in libsuggest_hip_nodense.so libsuggest_hip.so; do for c in headline cfg3 cfg2; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config $c --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
done; done; done
for lib in libsuggest_hip_nodense.so libsuggest_hip.so; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config cfg4 --steps 5 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib cfg4', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
  SG_LIB_NAME=$lib timeout 600 python bench.py --dict-variant skewed --steps 5 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib skewed', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
  SG_LIB_NAME=$lib timeout 600 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" 2>&1 | grep "per step" | sed "s/^/$lib /"
  SG_LIB_NAME=$lib timeout 600 python tools/small_dict_timing.py 2>&1 | grep "M q/s" | cut -c1-96 | sed "s/^/$lib /"
done
