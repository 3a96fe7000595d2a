#!/bin/bash
# tools/lds_conflict_probe.sh -- times of tools/_bin/lds_conflict_probe and its LDS counters per kernel (one PMC pass)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
$R/tools/_bin/lds_conflict_probe
O=/tmp/ldsprobe_$$; rm -rf $O; mkdir -p $O
(cd /tmp && TMPDIR=/tmp timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O -o pmc -- $R/tools/_bin/lds_conflict_probe > $O/log 2>&1)
python - "$O" <<'PY'
import sys, glob, sqlite3, collections
for f in glob.glob(sys.argv[1] + "/pmc_results.db"):
    cur = sqlite3.connect(f).cursor()
    agg = collections.OrderedDict()
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        agg.setdefault(k, collections.defaultdict(list))[c].append(v)
    for k, cs in agg.items():
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        print("%-60s SQ_LDS_BANK_CONFLICT %.4g  SQ_LDS_IDX_ACTIVE %.4g  ratio %5.1f %%  SQ_INSTS_LDS %.4g  IDX_ACTIVE per instruction %.2f" % (
            k[:60], c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
            c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_IDX_ACTIVE", 0) / max(c.get("SQ_INSTS_LDS", 1), 1)))
PY
rm -rf $O
