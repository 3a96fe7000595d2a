#!/bin/bash
# tools/ab_run.sh <workload> <libA> <libB> [rounds] -- same-box A/B of two builds: alternating bench.py runs, kernel_ms_avg per run
W=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for L in $A $B; do
    ACDSP_LIB=$PWD/$L python bench.py --workload $W --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac'],4))"
  done
done
