cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config'].get('kernel_path'), d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; }
for i in 1 2 3; do
for w in polydec ddc; do
$B --workload $w 2>/dev/null | pick "$w base"
ACDSP_LIB=ac_dsp_amd/lib/libacdsp_gennt.so $B --workload $w 2>/dev/null | pick "$w nt"
done
$B --workload mvavg 2>/dev/null | pick "mvavg base"
ACDSP_LIB=ac_dsp_amd/lib/libacdsp_mvnt.so $B --workload mvavg 2>/dev/null | pick "mvavg nt"
done
