set -x
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel_ms_avg'], d['roofline'].get('kernel_ms_min'), d['roofline']['frac'])"; }
for r in 0 2 4 8 16 32 64 128; do ACDSP_INTG_RPW=$r $B --workload intgdump | pick "intgdump rpw=$r"; done
for r in 0 2 4 8 16 32 64; do ACDSP_MVAVG_TPW=$r $B --workload mvavg | pick "mvavg tpw=$r"; done
$B --workload intgdump --samples 4194304 | pick "intgdump 4M"
$B --workload mvavg --channels 4096 | pick "mvavg 4096obj"
tools/_bin/copy_probe 2 5
tools/_bin/copy_probe 1 5
