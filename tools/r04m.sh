#!/bin/bash
# 8-bit gaps for dense terms (SG_G8): parity with every term forced to the format, then the default (auto) suite, then A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
SG_G8=2 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_spell.py tests/test_lm_binary.py -m gpu -x -q > $O/r04m_pytest_g8_forced.log 2>&1; tail -15 $O/r04m_pytest_g8_forced.log
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04m_pytest.log 2>&1; tail -5 $O/r04m_pytest.log
b() { env $1 timeout 900 python bench.py --config $2 --steps ${3:-20} --no-cpu-baseline --traffic ${5:-none} --sub-configs none ${4} 2> $O/r04m_last.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1 $2 $4', round(d['value']), 'kernel ms', round(r['kernel_ms_avg'],4), 'traffic', r.get('traffic'), 'frac', r.get('frac'), 'model', r.get('model_bytes'), 'bit_exact', d.get('parity_vs_oracle'))"; grep -i "packed store" $O/r04m_last.err | head -2; }
for g in 0 1 0 1; do b "SG_G8=$g SG_VERBOSE=1" cfg4 5; done
for g in 0 1; do b "SG_G8=$g SG_VERBOSE=1" headline 5 "--dict-variant skewed"; done
for g in 0 1; do b "SG_G8=$g SG_VERBOSE=1" headline 20; b "SG_G8=$g" cfg2 20; done
b "SG_G8=1" cfg4 5 "" live
b "SG_G8=1" headline 5 "--dict-variant skewed" live
