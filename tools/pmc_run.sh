#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# Collects PMC counters for the search kernel in separate passes (SQ=8, TCC=4 slots per pass) with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "FETCH_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic none "$@" > $OUT/pass$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        name = r.get("Kernel_Name", "")
        if "sg_search" in name:      # <kParts, kLM, kTight, kSlim>: the batch kernel <false,false,..>, the parts kernel <true,..>, the LM kernel <false,true,..>
            kind = "parts " if "<true" in name else "lm    " if "<false, true" in name else "search"
            acc[(kind, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kind, k), v in sorted(acc.items()):
        print("%s %-28s per-dispatch avg %.6g (n=%d)" % (kind, k, sum(v) / len(v), len(v)))
PY
