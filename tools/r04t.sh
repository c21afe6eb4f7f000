#!/bin/bash
# skewed dictionaries: ordinary queries at 2^11 counter words (12 wavefronts per CU), the heavy ones cut into parts that run with 2^12 / 2^13
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in skewed skewed-families; do
timeout 900 python tools/sweep_any.py --config headline --dict-variant $v "" "SG_LOG2_CNT=11" "SG_PARTS_ALWAYS=1,SG_PARTS_CNT_BONUS=1" "SG_PARTS_CNT_BONUS=2" "SG_SPLIT_CHUNKS=32768" "SG_SPLIT_CHUNKS=16384" "SG_SPLIT_CHUNKS=8192" "SG_SPLIT_CHUNKS=16384,SG_PARTS_CNT_BONUS=1" "SG_SPLIT_CHUNKS=16384,SG_PARTS_CNT_BONUS=3" "SG_PARTS_CNT_BONUS=2,SG_T_FLOOR=6" 2>&1 | grep -v amdgpu.ids | tee $O/r04t_sweep_${v}_parts.txt
done
for e in "SG_ROOMY=2" "SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=0 SG_TIGHTEN=1" "SG_ROOMY=1 SG_TIGHTEN=1" "SG_ROOMY=0 SG_TIGHTEN=0"; do
  env $e timeout 600 python tools/small_dict_timing.py 2>&1 | grep "M q/s" | cut -c1-96 | sed "s/^/$e | /"
done | tee $O/r04t_small_dictionaries_roomy.txt
timeout 900 python tools/sweep_any.py --config headline --dict-variant families --steps 20 "" "SG_ROOMY=0" "SG_ROOMY=1" "SG_ROOMY=0,SG_TIGHTEN=1" "SG_ROOMY=1,SG_TIGHTEN=1" 2>&1 | grep -v amdgpu.ids | tee $O/r04t_sweep_families_roomy.txt
