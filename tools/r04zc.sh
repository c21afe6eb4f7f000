#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
for lib in libsuggest_hip_prof_nodense.so libsuggest_hip_prof.so; do echo "== $lib"; SG_PROF_LIB=$lib timeout 900 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -14; done
