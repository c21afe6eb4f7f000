#!/bin/bash
# quiet lists (SG_QUIET): parity on the whole GPU suite, then A/B on every config; LDS atomic microbenchmark
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic_rate lds_atomic_rate.hip && /tmp/lds_atomic_rate) > $O/r04k_lds_atomic_rate.txt 2>&1; cat $O/r04k_lds_atomic_rate.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r04k_pytest.log 2>&1; tail -5 $O/r04k_pytest.log
b() { env $1 timeout 600 python bench.py --config $2 --steps ${3:-20} --no-cpu-baseline --traffic none --sub-configs none ${4} 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 $2 $4', round(d['value']), 'kernel ms', round(d['roofline']['kernel_ms_avg'],4), 'bit_exact', d.get('parity_vs_oracle'))"; }
for q in 0 1 3 0 1; do
  for c in headline cfg3 cfg2; do b SG_QUIET=$q $c 20; done
  b SG_QUIET=$q cfg4 5
done
for q in 0 1 3; do b SG_QUIET=$q headline 5 "--dict-variant skewed"; done
timeout 900 python tools/spell_sweep.py "SG_QUIET=0" "SG_QUIET=1" "SG_QUIET=3" "SG_QUIET=0" "SG_QUIET=1" > $O/r04k_spell_sweep.txt 2>&1; grep -v amdgpu.ids $O/r04k_spell_sweep.txt
for q in 0 1; do SG_QUIET=$q timeout 600 python tools/small_dict_timing.py 2>&1 | grep "M q/s" | cut -c1-100 | sed "s/^/quiet=$q /"; done
