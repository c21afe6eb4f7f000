#!/bin/bash
# usage (through gpurun): bash tools/gpu_round.sh <tag> [steps...]   steps: tests prof phases small load
TAG=$1; shift
STEPS=${@:-tests prof phases small load}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for s in $STEPS; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; tail -5 $O/${TAG}_pytest.log ;;
    prof) for c in headline cfg2 cfg3 cfg4; do timeout 900 bash tools/profile_config.sh $TAG $c; done ;;
    fullpmc) for c in headline cfg3 cfg4; do timeout 1200 bash tools/pmc_run.sh ${TAG}_$c --config $c --batches 1 > $O/${TAG}_pmc_$c.txt 2>&1; tail -40 $O/${TAG}_pmc_$c.txt; done ;;
    phases)
      (timeout 300 python tools/phase_timing.py --metric cosine --similarity 0.4 --topk 20; \
       timeout 300 python tools/phase_timing.py --ngram 2 --metric dice --similarity 0.5 --queries 16384; \
       timeout 300 python tools/phase_timing.py --golden cars --metric cosine --similarity 0.5 --topk 5; \
       timeout 300 python tools/phase_timing.py --golden words --metric cosine --similarity 0.5 --topk 5) > $O/${TAG}_phases.log 2>&1; cat $O/${TAG}_phases.log ;;
    variants) for v in skewed families skewed-families; do timeout 600 python bench.py --dict-variant $v --steps 10 > $O/${TAG}_bench_$v.json 2> $O/${TAG}_bench_$v.err; tail -c 300 $O/${TAG}_bench_$v.err; python -c "import json,sys; d=json.loads([l for l in open('$O/${TAG}_bench_$v.json') if l.startswith('{')][-1]); print('$v', round(d['value']), round(d['roofline']['frac'],3), d['cpu_baseline']['value'], d['parity_vs_oracle'])"; done ;;
    small) timeout 600 python tools/small_dict_timing.py > $O/${TAG}_small.log 2>&1; cat $O/${TAG}_small.log ;;
    load) (timeout 300 tests/cpp/_build/single_query_load 1000000 256 3; timeout 300 tests/cpp/_build/single_query_load 1000000 64 3; timeout 300 tests/cpp/_build/single_query_load 1000000 1 2) > $O/${TAG}_load.log 2>&1; cat $O/${TAG}_load.log ;;
  esac
done
