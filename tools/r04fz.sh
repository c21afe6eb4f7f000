#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 800 python tools/fuzz_parity.py --seconds 600 --seed 5150 --scale 10 > $O/r04fz_fuzz_parity_scale10.log 2>&1; tail -1 $O/r04fz_fuzz_parity_scale10.log
timeout 800 python tools/fuzz_parity.py --seconds 600 --seed 6160 > $O/r04fz_fuzz_parity.log 2>&1; tail -1 $O/r04fz_fuzz_parity.log
timeout 400 python tools/fuzz_spell.py --seconds 300 --seed 6160 > $O/r04fz_fuzz_spell.log 2>&1; tail -1 $O/r04fz_fuzz_spell.log
