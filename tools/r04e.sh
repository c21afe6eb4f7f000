#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for d in 0 1; do
    SG_ROOMY=1 SG_DENSE=$d timeout 600 python bench.py --config headline --steps 10 --no-cpu-baseline --traffic none --sub-configs none 2> $O/r04e_bench_roomy_d$d.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('roomy dense=$d headline', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
    grep "host buffers" $O/r04e_bench_roomy_d$d.err
done
timeout 1200 bash tools/pmc_run.sh r04e_cfg2 --config cfg2 > $O/r04e_pmc_cfg2.txt 2>&1; cat $O/r04e_pmc_cfg2.txt | grep -v "^lm\|^parts"
