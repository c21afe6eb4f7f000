#!/bin/bash
# [r5] ONE driver for the round's GPU-box jobs (the r04*.sh one-offs of round 4 were one file per run):
#   gpurun --timeout 1500 -- 'bash tools/gpu_job.sh <job> [args...]'
# Every job writes under gpurun_out/<tag>_*; what is worth keeping is copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
job=$1; shift

benchline() {  # config -> "config value frac"
  python bench.py --config $1 --no-cpu-baseline --traffic none --steps ${2:-20} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config'].get('baseline_config'), round(d['value']), 'ms', d['ms_per_step'])"
}

case $job in
  pipe_ab)      # pipe_ab <tag> <config> <variants> [extra args]: fused vs pipeline variants on one resident index, rows compared
    tag=$1; cfg=$2; var=$3; shift 3
    timeout 1200 python tools/pipe_ab.py --config $cfg --variants "$var" "$@" 2>&1 | tee $O/${tag}_pipe_ab_${cfg}.txt | tail -40 ;;
  pipe_prof)    # pipe_prof <tag> <config> <variant>: rocprofv3 kernel stats of the pipeline launches
    tag=$1; cfg=$2; var=$3; shift 3
    cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/${tag}_prof_${cfg} -o run -- python $R/tools/pipe_ab.py --config $cfg --variants "$var" --steps 10 "$@" > $O/${tag}_prof_${cfg}.log 2>&1
    cd $R; db=$(find $O/${tag}_prof_${cfg} -name "*.db" | head -1); python tools/rocprof_summary.py $db "${tag} ${cfg} ${var}" 2>/dev/null | head -24 | cut -c1-130 | tee $O/${tag}_prof_${cfg}_summary.txt ;;
  pmc)          # pmc <tag> <kernel substrings, |-separated> <config> <variant>: PMC counters of the matching kernels, separate passes (MI355X_MICROARCH.md)
    tag=$1; pat=$2; cfg=$3; var=$4; shift 4
    OUT=$O/${tag}_pmc_${cfg}; mkdir -p $OUT; cd /tmp; i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" \
               "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr" \
               "SQ_INSTS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
      i=$((i+1))
      timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -- python $R/tools/pipe_ab.py --config $cfg --variants "$var" --steps 3 "$@" > $OUT/pass$i.log 2>&1
    done
    cd $R; python - "$OUT" "$pat" <<'PY' | tee $O/${tag}_pmc_${cfg}.txt
import csv, glob, collections, sys
out, pats = sys.argv[1], sys.argv[2].split("|")
acc = collections.defaultdict(list)
for p in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        name = r.get("Kernel_Name", "")
        for pt in pats:
            if pt in name:
                acc[(pt, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (pt, k), v in sorted(acc.items()):
    v = v[len(v) // 2:]          # the later half of the dispatches: steady state
    print("%-22s %-24s per-dispatch avg %.6g (n=%d)" % (pt, k, sum(v) / len(v), len(v)))
PY
    ;;
  membench)     # membench <tag>: the memory system on this engine's access pattern, timed and under the PMC counters (roofline calibration)
    tag=$1; shift
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip || exit 1
    {
      echo "# tools/membench.hip: random contiguous segments, 2 GiB per launch, every load instruction fully useful; timed without the profiler"
      for buf in 128 1024 8192; do for seg in 128 256 512 1024 2048 4096 16384; do for inf in 1 2 4; do /tmp/membench $buf $seg $inf 5; done; done; done
      echo "# under rocprofv3 --pmc (per-dispatch average over the 6 launches of a run): counter value, and bytes_per_launch / value"
      cd /tmp
      for buf in 128 1024; do for seg in 256 1024 4096; do
        for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
          rm -rf /tmp/mb_pmc; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/mb_pmc -- /tmp/membench $buf $seg 4 5 > /tmp/mb.log 2>&1
          python - $buf $seg <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for p in glob.glob("/tmp/mb_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "gather" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
b = float(2**31)
for k, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("buf_MiB %5s seg_B %5s %-28s %.6g  bytes/value %.3f" % (sys.argv[1], sys.argv[2], k, m, b / m if m else 0))
PY
        done
      done; done
    } 2>&1 | tee $O/${tag}_membench.txt ;;
  suite)        # suite <tag> [pytest args]: the GPU suite
    tag=$1; shift
    timeout 2400 python -m pytest tests -m gpu -x -q "$@" > $O/${tag}_pytest.log 2>&1; tail -5 $O/${tag}_pytest.log ;;
  bench)        # bench <tag> [bench args]: the default bench line
    tag=$1; shift
    timeout 1500 python bench.py "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err; tail -3 $O/${tag}_bench.err; tail -c 3000 $O/${tag}_bench.json ;;
  bench_prof)   # bench_prof <tag> [bench args]: rocprofv3 --kernel-trace --stats of a bench.py run (the summary the roofline's kernel time must agree with)
    tag=$1; shift
    cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $O/${tag}_benchprof -o run -- python $R/bench.py --no-cpu-baseline --traffic none "$@" > $O/${tag}_benchprof.json 2> $O/${tag}_benchprof.err
    cd $R; db=$(find $O/${tag}_benchprof -name "*.db" | head -1); python tools/rocprof_summary.py $db "${tag} bench.py --no-cpu-baseline --traffic none $*" 2>/dev/null | head -30 | cut -c1-140 | tee $O/${tag}_benchprof_summary.txt ;;
  plan_stages)  # plan_stages <tag> <config> <variant>: [r6] the plan / verify launches cut short behind stage n (SG_PHASE_TIMING build): where their time goes
    tag=$1; cfg=$2; var=$3; shift 3
    make -C suggest_amd/csrc prof > /dev/null 2>&1 || exit 1
    cd /tmp
    for skip in 0 1048576 2097152 3145728 4194304 16777216 33554432 50331648 67108864; do
      rm -rf /tmp/ps_prof
      SG_LIB_NAME=libsuggest_hip_prof.so SG_DEBUG_SKIP=$skip timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_prof -- python $R/tools/pipe_ab.py --config $cfg --variants "$var" --steps 10 --no-fused "$@" > /tmp/ps.log 2>&1
      python - $skip <<'PY'
import csv, glob, sys
per = {}
for p in glob.glob("/tmp/ps_prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        for key in ("sg_plan_kernel", "sg_verify_kernel", "sg_stream_kernel", "query_order_count", "query_order_scatter"):
            if key in n:
                per.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
skip = int(sys.argv[1])
print("plan stage %d verify stage %d: " % ((skip >> 20) & 15, (skip >> 24) & 15) + "  ".join("%s %.1f us (n=%d)" % (k, sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) / 1e3, len(v)) for k, v in sorted(per.items())))
PY
    done 2>&1 | tee $O/${tag}_plan_stages_${cfg}.txt ;;
  ab_libs)      # ab_libs <tag> <config> <variant> <lib> [<lib> ...]: [r6] the same workload on several builds of the library (SG_LIB_NAME), twice each, interleaved
    tag=$1; cfg=$2; var=$3; shift 3
    for rep in 1 2; do for lib in "$@"; do
      echo "== $lib (pass $rep)"; SG_LIB_NAME=$lib timeout 900 python tools/pipe_ab.py --config $cfg --variants "$var" --steps 100 2>&1 | grep -v '^\[' | tail -4
    done; done 2>&1 | tee $O/${tag}_ab_libs_${cfg}.txt ;;
  wpc_sweep)    # wpc_sweep <tag> <config> <variant>: [r6] wavefronts per CU of the persistent plan / verify launches (0 = a workgroup per query), per-kernel times
    tag=$1; cfg=$2; var=$3; shift 3
    cd /tmp
    for wp in 0 8 16 24 28 32; do for wv in 0 16 28; do
      rm -rf /tmp/ps_prof
      SG_PLAN_WPC=$wp SG_VERIFY_WPC=$wv timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_prof -- python $R/tools/pipe_ab.py --config $cfg --variants "$var" --steps 10 --no-fused "$@" > /tmp/ps.log 2>&1
      python - $wp $wv <<'PY'
import csv, glob, sys
per = {}
for p in glob.glob("/tmp/ps_prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        for key in ("sg_plan_kernel", "sg_verify_kernel", "sg_stream_kernel"):
            if key in n:
                per.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("plan wpc %s verify wpc %s: " % (sys.argv[1], sys.argv[2]) + "  ".join("%s %.1f us" % (k, sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) / 1e3) for k, v in sorted(per.items())))
PY
    done; done 2>&1 | tee $O/${tag}_wpc_sweep_${cfg}.txt ;;
  timeline)     # timeline <tag> <config> <variant> <lib>...: [r6] the launches of one call in order, start offsets and durations (kernel trace), per library build
    tag=$1; cfg=$2; var=$3; shift 3
    cd /tmp
    for lib in "$@"; do
      rm -rf /tmp/tl_prof
      SG_LIB_NAME=$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_prof -- python $R/tools/pipe_ab.py --config $cfg --variants "$var" --steps 12 --no-fused > /tmp/tl.log 2>&1
      python - $lib <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob("/tmp/tl_prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# calls = runs of launches that begin with query_order_count
starts = [i for i, r in enumerate(rows) if "query_order_count" in r[2]]
calls = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
calls = [c for c in calls if any("sg_stream_kernel" in r[2] for r in c)][-6:-1]
import os
if os.environ.get("TL_SINGLE"):                                  # (concurrent launches change places from call to call: two calls as they ran)
    for c in calls[-2:]:
        print("== %s: one call" % sys.argv[1])
        for r in c: print("  +%8.1f .. %8.1f us  dur %8.1f  %s" % ((r[0] - c[0][0]) / 1e3, (r[1] - c[0][0]) / 1e3, (r[1] - r[0]) / 1e3, r[2][:60]))
    sys.exit(0)
print("== %s: %d calls averaged" % (sys.argv[1], len(calls)))
n = min(len(c) for c in calls)
for j in range(n):
    off = sum(c[j][0] - c[0][0] for c in calls) / len(calls) / 1e3
    dur = sum(c[j][1] - c[j][0] for c in calls) / len(calls) / 1e3
    gap = sum((c[j][0] - c[j - 1][1]) if j else 0 for c in calls) / len(calls) / 1e3
    print("  +%8.1f us  gap %6.1f  dur %8.1f  %s" % (off, gap, dur, calls[0][j][2][:70]))
print("  call span %.1f us; next call starts +%.1f us after this one's first launch" % (sum(c[-1][1] - c[0][0] for c in calls) / len(calls) / 1e3,
      sum(calls[i + 1][0][0] - calls[i][0][0] for i in range(len(calls) - 1)) / max(1, len(calls) - 1) / 1e3))
PY
    done 2>&1 | tee $O/${tag}_timeline_${cfg}.txt ;;
  sh)           # sh <command...>: anything else
    bash -c "$*" ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
