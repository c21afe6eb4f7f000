#!/bin/bash
# sweeps of the small-dictionary regime (1 M strings, the reference's words / cars): is level 2 still right there?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python tools/sweep_knobs.py --config cfg2 --levels 1,2,3,4,5 --floors 6,8,10 --cnt 10,11 --steps 8 2>&1 | grep -v amdgpu.ids > $O/r04q_sweep_cfg2.txt; sort -t: -k2 -n $O/r04q_sweep_cfg2.txt | head -10
for e in "SG_FILTER_LEVEL=2" "SG_FILTER_LEVEL=3" "SG_FILTER_LEVEL=4" "SG_FILTER_LEVEL=1" "SG_FILTER_LEVEL=2 SG_T_FLOOR=6" "SG_FILTER_LEVEL=2 SG_T_FLOOR=10" "SG_FILTER_LEVEL=2 SG_LOG2_CNT=10"; do
  env $e timeout 600 python tools/small_dict_timing.py 2>&1 | grep "M q/s" | cut -c1-96 | sed "s/^/$e | /"
done
