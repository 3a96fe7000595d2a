#!/bin/bash
# tools/ab_multi.sh <workload> <rounds> <lib>... -- same-box comparison of several builds: round-robin bench.py runs
W=$1; N=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    ACDSP_LIB=$PWD/$L python bench.py --workload $W --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac'],4))"
  done
done
