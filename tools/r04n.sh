#!/bin/bash
# big-register instantiations with a whole batch's counter updates in flight (SG_BIG): parity, then A/B on the long-list configs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04n_pytest.log 2>&1; tail -5 $O/r04n_pytest.log
b() { env $1 timeout 900 python bench.py --config $2 --steps ${3:-20} --no-cpu-baseline --traffic ${5:-none} --sub-configs none ${4} 2> $O/r04n_last.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1 $2 $4', round(d['value']), 'kernel ms', round(r['kernel_ms_avg'],4), 'traffic', r.get('traffic'), 'frac', r.get('frac'), 'bit_exact', d.get('parity_vs_oracle'))"; }
for g in 0 1 0 1; do b "SG_BIG=$g" cfg4 5; done
for g in 0 1 0 1; do b "SG_BIG=$g" headline 5 "--dict-variant skewed"; done
for g in 0 1; do b "SG_BIG=$g SG_LOG2_CNT=12" headline 10; b "SG_BIG=$g SG_LOG2_CNT=12" cfg3 10; done
b "SG_BIG=1 SG_G8=1" cfg4 5
