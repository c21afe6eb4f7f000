#!/bin/bash
# knob sweeps of the long-list configs (never swept together before): filter level x t_floor x counter words
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python tools/sweep_knobs.py --config cfg4 --levels 1,2,3,4 --floors 6,8,10,12 --cnt 11,12,13 --steps 4 2>&1 | grep -v amdgpu.ids > $O/r04o_sweep_cfg4.txt; sort -t: -k2 -n $O/r04o_sweep_cfg4.txt | head -12
timeout 1200 python tools/sweep_knobs.py --config headline --dict-variant skewed --levels 1,2,3,4 --floors 6,8,10,12 --cnt 11,12,13 --steps 4 2>&1 | grep -v amdgpu.ids > $O/r04o_sweep_skewed.txt; sort -t: -k2 -n $O/r04o_sweep_skewed.txt | head -12
