#!/bin/bash
# tools/lds_ab.sh <workload> <libA> <libB> [rounds] -- same-box A/B of two builds that differ in an LDS layout: alternating bench.py processes (ms per step,
# kernel ms, roofline fraction; --placement 1 so that both builds see the allocator's own blocks) and one rocprofv3 PMC pass per build with the LDS counters
# of the workload's dominant kernel.  Output: text for profiles/r6_lds_ab.txt.
W=$1; A=$2; B=$3; N=${4:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
echo "## $W: A = $A, B = $B"
for i in $(seq $N); do
  for L in $A $B; do
    ACDSP_LIB=$R/$L python bench.py --workload $W --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --placement 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $L  ms_per_step %.4f  kernel_ms %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
  done
done
for L in $A $B; do
  O=/tmp/ldsab_$$; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp ACDSP_LIB=$R/$L timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O -o pmc -- python $R/bench.py --workload $W --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --placement 1 > $O/log 2>&1)
  python - "$O" "$L" <<'PY'
import sys, glob, sqlite3, collections
d, lib = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/pmc_results.db"):
    cur = sqlite3.connect(f).cursor()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        agg[k][c].append(v)
    best = max((k for k in agg if "fill_stimulus" not in k and "rocclr" not in k and "hist_update" not in k), key=lambda k: sum(agg[k].get("SQ_INSTS_VALU", [0])), default=None)
    if best:
        c = {n: sum(v) / len(v) for n, v in agg[best].items()}
        print("  %s  %s" % (lib, best[:100]))
        print("      SQ_LDS_BANK_CONFLICT %.4g  SQ_LDS_IDX_ACTIVE %.4g  (%.1f %%)  SQ_INSTS_LDS %.4g  SQ_ACTIVE_INST_LDS %.4g  SQ_WAIT_INST_LDS %.4g  SQ_INSTS_VALU %.4g" % (
            c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
            c.get("SQ_INSTS_LDS", 0), c.get("SQ_ACTIVE_INST_LDS", 0), c.get("SQ_WAIT_INST_LDS", 0), c.get("SQ_INSTS_VALU", 0)))
PY
  rm -rf $O
done
