#!/bin/bash
# spell_tokenize_kernel staged through LDS: parity (Predict vectors, cfg 5 at size, spell fuzz), then the step time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_spell.py tests/test_lm_binary.py tests/test_gpu_configs.py -m gpu -x -q -k "spell or predict or Predict or lm or cfg5" 2>&1 | tail -3
timeout 300 python tools/fuzz_spell.py --seconds 150 --seed 777 2>&1 | tail -1
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" "SG_FILTER_LEVEL=4" 2>&1 | grep -v amdgpu.ids
cd /tmp; rm -rf /tmp/kt5; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/r04x_cfg5_trace.log 2>&1
python $R/tools/kernel_stats.py /tmp/kt5 --skip 3 > $O/r04x_kernel_stats_cfg5.csv 2>&1; cut -c1-150 $O/r04x_kernel_stats_cfg5.csv | head -9
