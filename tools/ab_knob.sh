# same-box comparison of one knob: bash tools/ab_knob.sh SG_ORDER 0 1
K=$1; shift
for v in "$@" "$@"; do
  echo "== $K=$v"
  env $K=$v python tools/small_dict_timing.py 2>&1 | tail -3 | cut -c1-86
  for c in headline cfg2 cfg3 cfg4; do env $K=$v python bench.py --config $c --no-cpu-baseline --traffic none --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"baseline_config\"], round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],3))"; done
  for d in families skewed; do env $K=$v python bench.py --dict-variant $d --no-cpu-baseline --traffic none --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],3))"; done
done
