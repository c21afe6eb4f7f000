#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
SG_ASYNC_TRACE=1 timeout 600 python tools/async_trace.py > $O/r04h_async.txt 2>&1; grep -v amdgpu.ids $O/r04h_async.txt | tail -18
for pt in 1 0; do
for c in headline cfg2 cfg3; do
  SG_PRETILE=$pt timeout 600 python bench.py --config $c --steps 20 --no-cpu-baseline --traffic none --sub-configs none 2> $O/r04h_bench_${c}_pt$pt.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pretile=$pt $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4), 'model GB', d['roofline'].get('model_bytes'), d.get('host_buffers',{}).get('value'), d.get('host_buffers_pipelined',{}).get('value'))"
done
done
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=4,SG_PRETILE=1" "SG_FILTER_LEVEL=4,SG_PRETILE=0" > $O/r04h_spell_sweep.txt 2>&1; cat $O/r04h_spell_sweep.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04h_pytest.log 2>&1; tail -6 $O/r04h_pytest.log
