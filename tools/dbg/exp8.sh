cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac'],3))"; }
for w in ddc polydec fir255 mvavg; do $B --workload $w 2>/dev/null | pick "fresh $w"; done
$B --workload cic_dec 2>/dev/null | pick "cic_dec"
for w in ddc polydec fir255 mvavg; do $B --workload $w 2>/dev/null | pick "after-cic_dec $w"; done
sleep 20
for w in ddc polydec; do $B --workload $w 2>/dev/null | pick "after-20s $w"; done
