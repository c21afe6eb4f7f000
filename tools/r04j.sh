#!/bin/bash
# baseline of the tree at re-entry: full -m gpu suite, default bench run (sub-records), cfg5 alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04j_pytest.log 2>&1; tail -8 $O/r04j_pytest.log
timeout 1500 python bench.py > $O/r04j_bench_default.json 2> $O/r04j_bench_default.err; tail -30 $O/r04j_bench_default.err | grep -v amdgpu.ids; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04j_bench_default.json') if l.startswith('{')][-1])
print('headline', round(d['value']), d['roofline']['frac'], d['roofline'].get('model_bytes'), d['roofline'].get('traffic'), d.get('host_buffers',{}).get('value'), d.get('host_buffers_pipelined'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), v.get('frac'), v.get('bit_exact'), v.get('kernels') and {a:(round(b['ms'],3)) for a,b in v['kernels'].items()})
PY
