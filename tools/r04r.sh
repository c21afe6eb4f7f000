#!/bin/bash
# new defaults (level 4 / counter words by skew): whole suite, the default bench line, then the remaining knobs of the long-list regime
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04r_pytest.log 2>&1; tail -5 $O/r04r_pytest.log
timeout 1500 python bench.py > $O/r04r_bench_default.json 2> $O/r04r_bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04r_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', round(d['value']), 'frac', r['frac'], 'traffic', r.get('traffic'), 'model', r.get('model_bytes'), 'host', d.get('host_buffers',{}).get('value'), 'piped', d.get('host_buffers_pipelined',{}).get('value'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), 'ms', v.get('kernel_ms_avg') or v.get('ms_per_step'), 'frac', v.get('frac'), 'traffic', v.get('traffic'), v.get('bit_exact'))
PY
timeout 900 python tools/sweep_any.py --config headline --dict-variant skewed "" "SG_SPLIT_CHUNKS=32768" "SG_SPLIT_CHUNKS=16384" "SG_SPLIT_CHUNKS=131072" "SG_SPLIT_CHUNKS=65536,SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=2,SG_TIGHTEN=0" "SG_TIGHTEN=1" "SG_TIGHTEN=2,SG_LOG2_CNT=13,SG_T_FLOOR=6" 2>&1 | grep -v amdgpu.ids | tee $O/r04r_sweep_skewed_other.txt
timeout 900 python tools/sweep_any.py --config cfg4 "" "SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=2,SG_SPLIT_CHUNKS=32768" "SG_SPLIT_CHUNKS=0" 2>&1 | grep -v amdgpu.ids | tee $O/r04r_sweep_cfg4_other.txt
timeout 900 python tools/sweep_any.py --config headline --steps 20 "" "SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=2,SG_T_FLOOR=10" "SG_T_FLOOR=9" "SG_T_FLOOR=8,SG_PRETOK=0" 2>&1 | grep -v amdgpu.ids | tee $O/r04r_sweep_headline_other.txt
