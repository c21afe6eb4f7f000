#!/bin/bash
# usage: tools/inst_breakdown.sh <tag> [phase_timing args...]   (GPU box; needs `make -C suggest_amd/csrc prof`)
# Instruction counts of the search kernel with parts of it switched off (SG_DEBUG_SKIP bits of the SG_PHASE_TIMING build):
# the difference between two rows is what the part in between issues.  The kernel is issue-bound (VALU ~70 % busy), so
# instructions — not wave-time, which also counts the stalls other wavefronts fill — are what a change has to remove.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/inst_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for skip in 32 64 16 512 4 1024 0; do
  SG_DEBUG_SKIP=$skip rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY \
     --kernel-trace --output-format csv -d $OUT/skip$skip -- python $GRAFT_REPO_ROOT/tools/phase_timing.py "$@" > $OUT/skip$skip.log 2>&1
done
python - <<PY
import csv, glob, collections
names = {32: "nothing (launch + exit)", 64: "+ tokenise", 16: "+ tile rows, segment stats", 512: "+ group setup, clears (no stream)",
         4: "+ row loads (no counting)", 1024: "+ counting (flagged postings dropped)", 0: "everything"}
for skip in (32, 64, 16, 512, 4, 1024, 0):
    acc = collections.defaultdict(list)
    for p in glob.glob("$OUT/skip%d/**/*counter_collection.csv" % skip, recursive=True):
        for r in csv.DictReader(open(p)):
            if "sg_search_kernel_t<false" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n_q = 65536.0
    row = {k: sum(v) / len(v) / n_q for k, v in acc.items()}
    print("skip %4d %-40s per query: VALU %7.0f SALU %7.0f LDS %6.0f VMEM %5.0f BRANCH %6.0f  wave quad-cycles %8.0f (waiting %8.0f)" % (
        skip, names[skip], row.get("SQ_INSTS_VALU", 0), row.get("SQ_INSTS_SALU", 0), row.get("SQ_INSTS_LDS", 0), row.get("SQ_INSTS_VMEM_RD", 0),
        row.get("SQ_INSTS_BRANCH", 0), row.get("SQ_WAVE_CYCLES", 0), row.get("SQ_WAIT_ANY", 0)))
PY
