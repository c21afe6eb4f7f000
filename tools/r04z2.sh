#!/bin/bash
# upper bound of what a layout without lanes behind the ends of the lists could gain: the experiment build drops a list's last row when it is
# less than half full (wrong results, timing only) -- same box A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for rep in 1 2; do for lib in libsuggest_hip.so libsuggest_hip_exp.so; do for c in headline cfg3 cfg2; do
  SG_LIB_NAME=$lib timeout 600 python bench.py --config $c --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4), 'model GB', round((d['roofline'].get('model_bytes') or 0)/1e9,3))"
done; done; done
for lib in libsuggest_hip.so libsuggest_hip_exp.so; do SG_LIB_NAME=$lib timeout 600 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" 2>&1 | grep "per step" | sed "s/^/$lib /"; done
