#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_spell.py tests/test_lm_binary.py tests/test_gpu_multi.py -m gpu -x -q -k "pages_through or spell or predict or Predict or async or lm" > $O/r04f_pytest.log 2>&1; tail -6 $O/r04f_pytest.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "cfg5" > $O/r04f_pytest_cfg5.log 2>&1; tail -4 $O/r04f_pytest_cfg5.log
for p in 1 0; do
  SG_ASYNC_PRIO=$p timeout 600 python bench.py --steps 10 --no-cpu-baseline --traffic none --sub-configs none 2> $O/r04f_bench_prio$p.err > /dev/null; echo "prio=$p"; grep "host buffers" $O/r04f_bench_prio$p.err
done
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=2" "SG_FILTER_LEVEL=4" > $O/r04f_spell_sweep.txt 2>&1; cat $O/r04f_spell_sweep.txt
