cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "ddc or state or graph or polydec or fullsize" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac_step'],3))"; }
for i in 1 2; do for w in ddc polydec cic_dec; do $B --workload $w 2>/dev/null | pick $w; done; done
