for v in 0 2048 0 2048; do
  echo "== SG_PRETOK=$v"
  for c in headline cfg2 cfg3 cfg4; do env SG_PRETOK=$v python bench.py --config $c --no-cpu-baseline --traffic none --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['baseline_config'], round(d['value']), d['roofline']['kernel_ms_avg'], d['parity_vs_oracle'].get('bit_exact') if isinstance(d.get('parity_vs_oracle'),dict) else d.get('parity_vs_oracle'))"; done
done
