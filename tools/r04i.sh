#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
SG_TRACE_REPS=24 timeout 600 python tools/async_trace.py 2>&1 | grep "depth"
SG_TRACE_REPS=24 SG_TRACE_DEPTH=2 timeout 600 python tools/async_trace.py 2>&1 | grep "depth"
SG_TRACE_REPS=24 SG_ASYNC_MEMCPY=1 timeout 600 python tools/async_trace.py 2>&1 | grep "depth" | sed 's/^/memcpy: /'
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "async" 2>&1 | tail -2
