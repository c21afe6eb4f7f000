#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04c_pytest.log 2>&1; tail -15 $O/r04c_pytest.log
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=2" "SG_FILTER_LEVEL=3" "SG_FILTER_LEVEL=4" "SG_FILTER_LEVEL=5" "SG_FILTER_LEVEL=6" "SG_FILTER_LEVEL=7" "SG_FILTER_LEVEL=6,SG_T_FLOOR=4" "SG_FILTER_LEVEL=7,SG_T_FLOOR=3" "SG_FILTER_LEVEL=2,SG_T_FLOOR=4" "SG_FILTER_LEVEL=2,SG_T_FLOOR=8,SG_ROOMY=1" "SG_FILTER_LEVEL=6,SG_T_FLOOR=8,SG_ROOMY=1" "SG_FILTER_LEVEL=6,SG_ROOMY=0,SG_LOG2_CNT=10" > $O/r04c_spell_sweep.txt 2>&1; cat $O/r04c_spell_sweep.txt
