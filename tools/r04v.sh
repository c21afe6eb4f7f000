#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 bash tools/profile_config.sh r04z headline --sub-configs none > $O/r04z_profile_headline.log 2>&1; tail -4 $O/r04z_profile_headline.log
timeout 900 python bench.py --config cfg4 --sub-configs none > $O/r04z_bench_cfg4.json 2> $O/r04z_bench_cfg4.err; python -c "
import json; d=json.loads([l for l in open('$O/r04z_bench_cfg4.json') if l.startswith('{')][-1]); r=d['roofline']; print('cfg4', round(d['value']), r['kernel_ms_avg'], r['frac'], r['traffic'], r.get('model_bytes'), r.get('traffic_over_model'), d['parity_vs_oracle'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or synthetic_q2 or docid_range or knobs" 2>&1 | tail -2
