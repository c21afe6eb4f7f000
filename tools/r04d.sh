#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pages_through_every_candidate" > $O/r04d_from.log 2>&1; tail -5 $O/r04d_from.log
SG_DENSE=1 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_spell.py tests/test_lm_binary.py -m gpu -x -q > $O/r04d_pytest_dense.log 2>&1; tail -15 $O/r04d_pytest_dense.log
for d in 0 1; do
  for c in headline cfg3 cfg2; do
    SG_DENSE=$d timeout 600 python bench.py --config $c --steps 10 --no-cpu-baseline --traffic none --sub-configs none 2> $O/r04d_bench_${c}_d$d.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dense=$d $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
  done
done
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=2" "SG_FILTER_LEVEL=2,SG_DENSE=1" "SG_FILTER_LEVEL=4,SG_DENSE=0" "SG_FILTER_LEVEL=4,SG_DENSE=1" "SG_FILTER_LEVEL=5,SG_DENSE=1" > $O/r04d_spell_sweep.txt 2>&1; cat $O/r04d_spell_sweep.txt
SG_DENSE=1 timeout 600 python tools/small_dict_timing.py > $O/r04d_small_dense.log 2>&1; cat $O/r04d_small_dense.log
SG_DENSE=0 timeout 600 python tools/small_dict_timing.py > $O/r04d_small_plain.log 2>&1; cat $O/r04d_small_plain.log
