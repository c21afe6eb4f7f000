#!/usr/bin/env python3
"""One screen of a bench.py JSON line: headline, roofline, per-kernel times, the sub-configs.  python tools/bench_summary.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.2f M %s  %.4f ms/step  frac %.3f  of achievable %.3f  traffic %.2f GB  model %.2f GB" % (d["value"] / 1e6, d["unit"], d["ms_per_step"], r["frac"], r.get("frac_of_achievable") or 0, (r.get("traffic") or 0) / 1e9, (r.get("model_bytes") or 0) / 1e9))
dm = r.get("dominant") or {}
print("dominant", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in dm.items()})
print("kernels ms/call", {k: round(v["ms_per_call_under_profiler"], 4) for k, v in (r.get("kernels") or {}).items()})
print("parity", d.get("parity_vs_oracle"), "pipeline", d["config"].get("pipeline"))
hb = d.get("host_buffers_pipelined") or {}
print("host buffers: sync %.2f M, pipelined %.2f M" % ((d.get("host_buffers") or {}).get("value", 0) / 1e6, hb.get("value", 0) / 1e6))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", {k: cb.get(k) for k in ("value", "unit", "cores", "kind")})
for k, v in (d.get("configs") or {}).items():
    print("%-7s %8.2f M  %.3f ms  frac %.3f  bit_exact %s" % (k, v["value"] / 1e6, v["ms_per_step"], v.get("frac") or 0, (v.get("parity_vs_oracle") or {}).get("bit_exact", v.get("bit_exact"))))
