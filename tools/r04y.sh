#!/bin/bash
# Predict's small kernels (tokeniser staged + lockstep lookups, lane-parallel merge counts), q_sel loads issued together, ASCII table up front:
# whole suite, fuzzers, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04y_pytest.log 2>&1; tail -5 $O/r04y_pytest.log
timeout 500 python tools/fuzz_parity.py --seconds 300 --seed 991 > $O/r04y_fuzz_parity.log 2>&1; tail -1 $O/r04y_fuzz_parity.log
timeout 300 python tools/fuzz_spell.py --seconds 150 --seed 991 > $O/r04y_fuzz_spell.log 2>&1; tail -1 $O/r04y_fuzz_spell.log
timeout 1500 python bench.py > $O/r04y_bench_default.json 2> $O/r04y_bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04y_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', round(d['value']), 'frac', r['frac'], 'traffic', r.get('traffic'), 'model', r.get('model_bytes'), 'host', d.get('host_buffers',{}).get('value'), 'piped', d.get('host_buffers_pipelined',{}).get('value'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), 'ms', v.get('kernel_ms_avg') or v.get('ms_per_step'), 'frac', v.get('frac'), 'traffic', v.get('traffic'), 'model', v.get('model_bytes'), v.get('bit_exact'))
PY
cd /tmp; rm -rf /tmp/kt5; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/r04y_cfg5_trace.log 2>&1
python $R/tools/kernel_stats.py /tmp/kt5 --skip 3 > $O/r04y_kernel_stats_cfg5.csv 2>&1; cut -c1-150 $O/r04y_kernel_stats_cfg5.csv | head -9
cd $R; timeout 600 python bench.py --config cfg2 --sub-configs none --no-cpu-baseline --traffic none 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg2', round(d['value']), d['roofline']['kernel_ms_avg'])"
