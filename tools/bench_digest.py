#!/usr/bin/env python3
"""One screen of a bench.py line: headline, roofline, spread, CPU baseline, sub-records.   python tools/bench_digest.py <file.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.2f M q/s  %.4f ms/step  steps %d  timed region %.3f s" % (d["value"] / 1e6, d["ms_per_step"], d["steps"], r.get("timed_region_s") or 0))
for k in ("frac", "frac_of_achievable", "traffic", "model_bytes", "traffic_over_model", "dominant_kernel", "dominant_ms", "dominant_traffic", "dominant_frac",
          "dominant_frac_of_achievable", "plan_ms", "stream_ms", "verify_ms", "order_ms", "kernel_ms_avg", "kernel_ms_p50", "kernel_ms_stdev", "kernel_ms_min",
          "kernel_ms_max", "effective_frac"):
    print("  %-28s %s" % (k, r.get(k)))
c = d.get("cpu_baseline") or {}
print("cpu", {k: (v if not isinstance(v, dict) else round(v.get("value"))) for k, v in c.items() if k != "sample"})
print("parity", d.get("parity_vs_oracle"))
print("pipeline", d["config"]["pipeline"])
print("host_buffers", round((d.get("host_buffers") or {}).get("value", 0) / 1e6, 2), "pipelined", round((d.get("host_buffers_pipelined") or {}).get("value", 0) / 1e6, 2))
for n, s in (d.get("configs") or {}).items():
    print("%-9s %9.3f M  %8.4f ms  frac %s  traffic/model %s  bit_exact %s (%s rows)  results/query %s  pipeline %s %s  kernels %s" % (
        n, s["value"] / 1e6, s["ms_per_step"], None if s.get("frac") is None else round(s["frac"], 3), None if s.get("traffic_over_model") is None else round(s["traffic_over_model"], 2),
        s.get("bit_exact"), s.get("checked_queries"), None if s.get("results_per_query") is None else round(s["results_per_query"], 2),
        (s.get("pipeline") or {}).get("on"), (s.get("pipeline") or {}).get("queries_per_call"),
        {k: round(v, 3) for k, v in (s.get("kernels_ms") or {}).items()} if isinstance(s.get("kernels_ms"), dict) else ""))
