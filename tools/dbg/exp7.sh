cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac'],3))"; }
for i in 1 2 3 4; do for w in ddc polydec; do $B --workload $w 2>/dev/null | pick "$w pad0 run$i"; done; done
for p in 64 2048; do for i in 1 2 3; do for w in ddc polydec; do $B --workload $w --pad $p 2>/dev/null | pick "$w pad$p run$i"; done; done; done
