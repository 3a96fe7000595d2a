cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "intg or fuzz or golden or abi" 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config'].get('kernel_path'), d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; }
for i in 1 2; do
ACDSP_NO_INTG_BATCH=1 ACDSP_INTG_RPW=8 $B --workload intgdump 2>/dev/null | pick "old rpw=8"
for r in 8 16 32 64; do ACDSP_INTG_RPW=$r $B --workload intgdump 2>/dev/null | pick "batch lpw=$r"; done
done
