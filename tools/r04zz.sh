#!/bin/bash
# the round's last tree: whole suite, smoke, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04zz_pytest.log 2>&1; tail -4 $O/r04zz_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > $O/r04zz_bench_default.json 2> $O/r04zz_bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04zz_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', round(d['value']), 'ms', d['ms_per_step'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'model', r.get('model_bytes'), 'host', d.get('host_buffers',{}).get('value'), 'piped', d.get('host_buffers_pipelined',{}).get('value'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), 'ms', v.get('kernel_ms_avg') or v.get('ms_per_step'), 'frac', v.get('frac'), v.get('bit_exact'))
PY
