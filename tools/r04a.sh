#!/bin/bash
# round 4, first GPU call: what bounds cfg 5's two search launches (phase shares on the real query mix + PMC passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python tools/phase_timing_cfg5.py > $O/r04a_phases_cfg5.txt 2>&1; cat $O/r04a_phases_cfg5.txt
timeout 1500 bash tools/pmc_run.sh r04a_cfg5 --config cfg5 > $O/r04a_pmc_cfg5.txt 2>&1; cat $O/r04a_pmc_cfg5.txt
