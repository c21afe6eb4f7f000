#!/bin/bash
# usage: tools/profile_config.sh <tag> <config> [--full-pmc] [bench args...]      (on the GPU box, through gpurun)
# One BASELINE config, three ways, outputs under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_bench_<config>.json          bench.py line (HIP-event kernel time, algorithmic GB/s, CPU oracle beside it)
#   <tag>_kernel_stats_<config>.csv    rocprofv3 --kernel-trace of the same command, warm-up dispatches dropped (tools/kernel_stats.py)
#   <tag>_traffic_<config>.json        HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (its own pass, --kernel-trace only),
#                                      x1024 (KB) x2 (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md §HBM)
#   --full-pmc: the other counter passes as well (tools/pmc_run.sh) -> <tag>_pmc_<config>.txt
set -u
TAG=$1; CFG=$2; shift 2
FULL=0
if [ "${1:-}" = "--full-pmc" ]; then FULL=1; shift; fi
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --config $CFG "$@" > $O/${TAG}_bench_$CFG.json 2> $O/${TAG}_bench_$CFG.err
W=3; K=6
rm -rf $O/${TAG}_trace_$CFG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_$CFG -- python $R/bench.py --config $CFG --steps $K --warmup $W --no-cpu-baseline --traffic none "$@" > $O/${TAG}_trace_$CFG.log 2>&1
python $R/tools/kernel_stats.py $O/${TAG}_trace_$CFG --skip $W > $O/${TAG}_kernel_stats_$CFG.csv
rm -rf $O/${TAG}_fetch_$CFG
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch_$CFG -- python $R/bench.py --config $CFG --steps $K --warmup $W --no-cpu-baseline --traffic none "$@" > $O/${TAG}_fetch_$CFG.log 2>&1
python - "$O/${TAG}_fetch_$CFG" "$O/${TAG}_bench_$CFG.json" "$TAG" "$CFG" > $O/${TAG}_traffic_$CFG.json <<'PY'
import csv, glob, json, sys
d, bench, tag, cfg = sys.argv[1:5]
vals = []
for p in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        if "sg_search_kernel_t<false, false," in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
parts = []
for p in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        if ("sg_search_kernel_t<true, false," in r["Kernel_Name"] or "sg_terms_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
            parts.append(float(r["Counter_Value"]))
out = {"config": cfg, "tag": tag, "dispatches": len(vals)}
if vals:
    kb = sum(vals) / len(vals) + (sum(parts) / len(vals) if parts else 0.0)
    out["fetch_size_kb_per_launch"] = kb
    out["bytes_per_launch"] = kb * 1024 * 2
    out["source"] = "rocprofv3 --pmc FETCH_SIZE --kernel-trace (tools/profile_config.sh %s %s): FETCH_SIZE %.6g KB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md); search + parts + tokeniser kernels, %d launches" % (tag, cfg, kb, len(vals))
try:
    b = json.loads([l for l in open(bench) if l.startswith("{")][-1])
    out["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_launch"]
    out["kernel_ms_avg"] = b["roofline"]["kernel_ms_avg"]
    out["value_qps"] = b["value"]
    if vals:
        out["wire_gbps"] = out["bytes_per_launch"] / (b["roofline"]["kernel_ms_avg"] * 1e-3) / 1e9
        out["traffic_over_algorithmic"] = out["bytes_per_launch"] / b["roofline"]["algorithmic_bytes_per_launch"]
except Exception as e:
    out["bench_error"] = repr(e)
print(json.dumps(out, indent=1))
PY
if [ $FULL = 1 ]; then
  bash $R/tools/pmc_run.sh ${TAG}_$CFG --config $CFG "$@" > $O/${TAG}_pmc_$CFG.txt 2>&1
fi
tail -c 600 $O/${TAG}_bench_$CFG.err
cat $O/${TAG}_traffic_$CFG.json
head -3 $O/${TAG}_kernel_stats_$CFG.csv
