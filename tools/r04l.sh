#!/bin/bash
# cfg 5: tightening on / off / adaptive for the fuzzy top-up, per-kernel trace; cfg 4: the counters (how HBM-bound is it?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python tools/spell_sweep.py "SG_TIGHTEN=2" "SG_TIGHTEN=1" "SG_TIGHTEN=0" "SG_TIGHTEN=2" "SG_TIGHTEN=1" "SG_TIGHTEN=1,SG_ROOMY=0" "SG_TIGHTEN=1,SG_ROOMY=1" > $O/r04l_spell_sweep.txt 2>&1; grep -v amdgpu.ids $O/r04l_spell_sweep.txt
cd /tmp; rm -rf /tmp/kt5; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/r04l_cfg5_trace.log 2>&1
python $R/tools/kernel_stats.py /tmp/kt5 --skip 3 > $O/r04l_kernel_stats_cfg5.csv 2>&1; cut -c1-150 $O/r04l_kernel_stats_cfg5.csv | head -24
cd $R; timeout 1500 bash tools/pmc_run.sh r04l_cfg4 --config cfg4 --sub-configs none > $O/r04l_pmc_cfg4.txt 2>&1; grep -v "^lm\|^parts" $O/r04l_pmc_cfg4.txt | grep -v amdgpu.ids
