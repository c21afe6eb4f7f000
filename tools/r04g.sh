#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/tools/async_trace.py > $O/r04g_async_plain.txt 2>&1; tail -2 $O/r04g_async_plain.txt
rm -rf /tmp/atr; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d /tmp/atr -- python $R/tools/async_trace.py > $O/r04g_async_traced.txt 2>&1; tail -2 $O/r04g_async_traced.txt
python - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
rows=[]
for p in glob.glob("/tmp/atr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:40], r.get("Queue_Id","")))
for p in glob.glob("/tmp/atr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name","")), ""))
rows.sort()
t0=rows[0][0]
out=open(O+"/r04g_timeline.txt","w")
for s,e,kind,name,q in rows[-160:]:
    out.write("%10.3f %10.3f %8.3f %s %s %s\n" % ((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,kind,name,q))
out.close()
print(open(O+"/r04g_timeline.txt").read()[-6000:])
PY
cd $R
timeout 900 python tools/spell_sweep.py "SG_FILTER_LEVEL=4" "SG_FILTER_LEVEL=4,SG_LOG2_CNT=10" "SG_FILTER_LEVEL=4,SG_LOG2_CNT=10,SG_W4=1" "SG_FILTER_LEVEL=2,SG_LOG2_CNT=10,SG_W4=1" "SG_FILTER_LEVEL=3,SG_LOG2_CNT=10,SG_W4=1" "SG_FILTER_LEVEL=5,SG_LOG2_CNT=10,SG_W4=1" > $O/r04g_spell_sweep.txt 2>&1; cat $O/r04g_spell_sweep.txt
for e in "SG_LOG2_CNT=11 SG_W4=0" "SG_LOG2_CNT=10 SG_W4=0" "SG_LOG2_CNT=10 SG_W4=1"; do
  for c in cfg2 headline; do
    env $e timeout 600 python bench.py --config $c --steps 10 --no-cpu-baseline --traffic none --sub-configs none 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e $c', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4))"
  done
  env $e timeout 600 python tools/small_dict_timing.py 2>&1 | grep "M q/s" | sed "s/^/$e /"
done
