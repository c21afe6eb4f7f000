#!/bin/bash
# second sweep: looser filter levels, and the index statistics the defaults are chosen from
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python tools/sweep_knobs.py --config cfg4 --levels 4,5,6,7 --floors 6,8 --cnt 10,11,12 --steps 4 2>&1 | grep -v amdgpu.ids > $O/r04p_sweep_cfg4.txt; sort -t: -k2 -n $O/r04p_sweep_cfg4.txt | head -8
timeout 1200 python tools/sweep_knobs.py --config headline --dict-variant skewed --levels 4,5,6,7 --floors 5,6,8 --cnt 12,13,14 --steps 4 2>&1 | grep -v amdgpu.ids > $O/r04p_sweep_skewed.txt; sort -t: -k2 -n $O/r04p_sweep_skewed.txt | head -8
timeout 1200 python tools/sweep_knobs.py --config headline --dict-variant skewed-families --levels 2,4,5 --floors 6,8 --cnt 11,12,13 --steps 4 2>&1 | grep -v amdgpu.ids > $O/r04p_sweep_skewed_families.txt; sort -t: -k2 -n $O/r04p_sweep_skewed_families.txt | head -8
for c in "cfg4" "headline --dict-variant skewed" "headline" "cfg3" "cfg2" "headline --dict-variant families" "headline --dict-variant skewed-families"; do
  SG_VERBOSE=1 timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --traffic none --sub-configs none 2>&1 | grep "suggest_hip\] terms" | sed "s/^/$c: /"
done
