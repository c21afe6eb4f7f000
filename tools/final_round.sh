#!/bin/bash
# usage (through gpurun): bash tools/final_round.sh <tag>   — the measurement set a round's profiles/ are made from
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/${TAG}_bench_default_run.json 2> $O/${TAG}_bench_default_run.err; tail -c 400 $O/${TAG}_bench_default_run.err
for c in headline cfg2 cfg3 cfg4; do timeout 900 bash tools/profile_config.sh $TAG $c --sub-configs none > $O/${TAG}_profile_$c.log 2>&1; tail -4 $O/${TAG}_profile_$c.log; done
timeout 900 bash tools/pmc_run.sh ${TAG}_headline --config headline --batches 1 --sub-configs none > $O/${TAG}_pmc_headline.txt 2>&1; tail -45 $O/${TAG}_pmc_headline.txt
timeout 900 python bench.py --config cfg5 --steps 10 > $O/${TAG}_bench_cfg5.json 2> $O/${TAG}_bench_cfg5.err; tail -c 300 $O/${TAG}_bench_cfg5.err; tail -c 600 $O/${TAG}_bench_cfg5.json
for v in skewed families; do timeout 600 python bench.py --dict-variant $v --steps 10 --sub-configs none > $O/${TAG}_bench_$v.json 2> $O/${TAG}_bench_$v.err; python -c "import json,sys; d=json.loads([l for l in open('$O/${TAG}_bench_$v.json') if l.startswith('{')][-1]); print('$v', round(d['value']), d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_vs_oracle'])"; done
timeout 600 python tools/small_dict_timing.py > $O/${TAG}_small_dictionaries.txt 2>&1; tail -3 $O/${TAG}_small_dictionaries.txt
timeout 300 python tools/latency.py > $O/${TAG}_latency.txt 2>&1; tail -5 $O/${TAG}_latency.txt
timeout 900 python tools/fuzz_parity.py --seconds 600 --seed 4242 > $O/${TAG}_fuzz_parity.log 2>&1; tail -1 $O/${TAG}_fuzz_parity.log
timeout 300 python tools/fuzz_spell.py --seconds 150 --seed 4242 > $O/${TAG}_fuzz_spell.log 2>&1; tail -1 $O/${TAG}_fuzz_spell.log
timeout 900 bash tools/profile_config.sh ${TAG}_skewed headline --dict-variant skewed --sub-configs none --steps 5 > $O/${TAG}_profile_skewed.log 2>&1     # (files: <tag>_skewed_*_headline.*)
cd /tmp; rm -rf /tmp/kt5; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/${TAG}_cfg5_trace.log 2>&1
python $R/tools/kernel_stats.py /tmp/kt5 --skip 3 > $O/${TAG}_kernel_stats_cfg5.csv 2>&1; cut -c1-150 $O/${TAG}_kernel_stats_cfg5.csv | head -8
cd $R; timeout 900 bash tools/pmc_run.sh ${TAG}_cfg4 --config cfg4 --sub-configs none > $O/${TAG}_pmc_cfg4.txt 2>&1; grep "^search" $O/${TAG}_pmc_cfg4.txt | head -40
