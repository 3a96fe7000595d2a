#!/bin/bash
# tools/ktrace.sh <tag> [bench args...] -- kernel-trace only (pass 0 of tools/prof.sh), prints the per-kernel table
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > "$OUT/trace.log" 2>&1 < /dev/null
python "$R/tools/pmc_summary.py" "$OUT" | cut -c1-150 | head -20
grep -h '"metric"' "$OUT/trace.log" | cut -c1-200
rm -rf "$OUT"/*.db
