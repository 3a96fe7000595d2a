cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "intg or mvavg or mv_avg or fuzz or golden or abi or cpp" 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config'].get('kernel_path'), d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; }
for i in 1 2 3; do
ACDSP_LIB=ac_dsp_amd/lib/libacdsp_old.so $B --workload mvavg 2>/dev/null | pick "mvavg old"
$B --workload mvavg 2>/dev/null | pick "mvavg new"
$B --workload intgdump 2>/dev/null | pick "intgdump new default"
done
