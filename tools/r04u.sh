#!/bin/bash
# defaults of the round (filter level / counter words by index shape, small queue): whole suite + the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04u_pytest.log 2>&1; tail -5 $O/r04u_pytest.log
timeout 1500 python bench.py > $O/r04u_bench_default.json 2> $O/r04u_bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04u_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('headline', round(d['value']), 'frac', r['frac'], 'traffic', r.get('traffic'), 'model', r.get('model_bytes'), 'host', d.get('host_buffers',{}).get('value'), 'piped', d.get('host_buffers_pipelined',{}).get('value'))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']), 'ms', v.get('kernel_ms_avg') or v.get('ms_per_step'), 'frac', v.get('frac'), 'traffic', v.get('traffic'), v.get('bit_exact'))
PY
SG_BENCH_SINGLE_DEVICE=1 timeout 900 python bench.py --mode replicas --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline 2> $O/r04u_replicas.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); rm=d['replicas_mode']; print('replicas x2 on one GPU: multi', round(rm['value']), 'pipelined', round(rm['pipelined']['value']))"
tail -3 $O/r04u_replicas.err
