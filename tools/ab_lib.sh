# same-box A/B of two library builds (suggest_amd/libsuggest_hip.so vs suggest_amd/old/libsuggest_hip.so)
for lib in libsuggest_hip.so old/libsuggest_hip.so libsuggest_hip.so old/libsuggest_hip.so; do
  echo "== $lib"
  SG_LIB_NAME=$lib python tools/small_dict_timing.py 2>&1 | tail -3 | cut -c1-86
  for c in headline cfg2 cfg3 cfg4; do SG_LIB_NAME=$lib python bench.py --config $c --no-cpu-baseline --traffic none --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"baseline_config\"], round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],3))"; done
done
for lib in libsuggest_hip.so old/libsuggest_hip.so; do for v in families skewed; do SG_LIB_NAME=$lib python bench.py --dict-variant $v --no-cpu-baseline --traffic none --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $v', round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],3))"; done; done
