#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 evidence for one bench.py configuration (run on the GPU box):
#   pass 0: --kernel-trace --stats          (per-kernel durations)
#   pass 1..4: --pmc, one counter group each (separate runs, never combined with trace domains)
# Results land in gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns them into a text summary for profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*"
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1
timeout -k 5 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OUT" -o pmc1 -- python "$R/bench.py" $ARGS > "$OUT/pmc1.log" 2>&1
timeout -k 5 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT" -o pmc2 -- python "$R/bench.py" $ARGS > "$OUT/pmc2.log" 2>&1
timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o pmc3 -- python "$R/bench.py" $ARGS > "$OUT/pmc3.log" 2>&1
timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o pmc4 -- python "$R/bench.py" $ARGS > "$OUT/pmc4.log" 2>&1
grep -h '"metric"' "$OUT/trace.log" | cut -c1-200
ls "$OUT"
