#!/bin/bash
# cfg 5: the remaining knobs of the vocabulary index (tokeniser launches, counter words, t_floor, queue)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" "SG_PRETOK=0" "SG_PRETOK=2048,SG_LOG2_CNT=10" "SG_LOG2_CNT=12" "SG_LOG2_CNT=11,SG_T_FLOOR=6" "SG_T_FLOOR=10" "SG_T_FLOOR=8,SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=2,SG_TIGHTEN=0" "SG_TIGHTEN=2,SG_ORDER=0" "SG_ORDER=1,SG_FILTER_LEVEL=3" "SG_FILTER_LEVEL=5" 2>&1 | grep -v amdgpu.ids | tee $O/r04s_spell_sweep.txt
