#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python bench.py > $O/r04z_bench_default_run.json 2> $O/r04z_bench_default_run.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04z_bench_default_run.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', round(d['value']), 'frac', r['frac'], 'traffic', r.get('traffic'), 'model', r.get('model_bytes'), 'host', d.get('host_buffers',{}).get('value'), 'piped', d.get('host_buffers_pipelined',{}).get('value'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), 'ms', v.get('kernel_ms_avg') or v.get('ms_per_step'), 'frac', v.get('frac'), 'traffic', v.get('traffic'), 'model', v.get('model_bytes'), v.get('bit_exact'))
PY
for c in "cfg4" "headline --dict-variant skewed"; do n=$(echo $c | sed 's/headline --dict-variant //'); timeout 900 python bench.py --config $c --sub-configs none > $O/r04z_bench_$n.json 2> $O/r04z_bench_$n.err; python -c "
import json; d=json.loads([l for l in open('$O/r04z_bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']; print('$n', round(d['value']), r['kernel_ms_avg'], r['frac'], r['traffic'], r.get('model_bytes'), r.get('traffic_over_model'), d['parity_vs_oracle'])"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "split or synthetic or heaviest or async or replicas_and" 2>&1 | tail -2
