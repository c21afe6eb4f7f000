cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 600 python bench.py --config cfg5 --no-cpu-baseline --traffic none --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', d.get('parity_vs_oracle') or d.get('bit_exact'))"; }
run SG_PIPE=2
run SG_PIPE=1
run SG_PIPE=1 SG_PIPE_CAND_CAP=256
run SG_PIPE=1 SG_PIPE_CAND_CAP=1024
run SG_PIPE=1 SG_PIPE_CAND_CAP=256 SG_FILTER_LEVEL=4
