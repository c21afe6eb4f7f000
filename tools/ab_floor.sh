# same-box comparison of SG_T_FLOOR values over the configs and dictionaries (through gpurun)
for v in "$@"; do
  echo "== SG_T_FLOOR=$v"
  for c in cfg2 cfg4; do env SG_T_FLOOR=$v python bench.py --config $c --no-cpu-baseline --traffic none --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['baseline_config'], round(d['value']), d['roofline']['kernel_ms_avg'])"; done
  for d in families skewed; do env SG_T_FLOOR=$v python bench.py --dict-variant $d --no-cpu-baseline --traffic none --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['value']), d['roofline']['kernel_ms_avg'])"; done
  env SG_T_FLOOR=$v python tools/small_dict_timing.py 2>&1 | tail -3 | cut -c1-90
done
