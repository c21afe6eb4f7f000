#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04b_pytest.log 2>&1; tail -15 $O/r04b_pytest.log
timeout 1500 python bench.py > $O/r04b_bench_default.json 2> $O/r04b_bench_default.err; tail -40 $O/r04b_bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04b_bench_default.json') if l.startswith('{')][-1])
print('headline', round(d['value']), d['roofline']['frac'], d.get('host_buffers'), d.get('host_buffers_pipelined'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), v.get('frac'), v.get('bit_exact'), v.get('kernels') and {a:(round(b['ms'],3)) for a,b in v['kernels'].items()})
PY
