cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --workload intgdump"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['event_ms_per_step'], d['clock_settle']['steps'])"; }
for s in 0 0.05 0.3 0.3 1.0; do $B --settle $s 2>/dev/null | pick "settle=$s"; done
python tools/clock_ramp.py intgdump 0 2>/dev/null | awk 'NR<8 || NR%6==0'
