#!/bin/bash
# what the counters say about the rows behind list ends: the drop-tails build (wrong results, timing only) against the product build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for lib in libsuggest_hip.so libsuggest_hip_exp.so; do
  SG_LIB_NAME=$lib timeout 1200 bash tools/pmc_run.sh r04zg_$lib --config headline --batches 1 --sub-configs none 2>&1 | grep "^search" | awk -v L=$lib '{print L, $2, $5}'
done
