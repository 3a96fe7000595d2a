#!/bin/bash
# tools/clock_study.sh [out.txt] -- on the GPU box: the evidence behind "what bounds the 255-tap FIR kernel".
#   (a) shader clock the waves of the product kernel see (ACDSP_DEBUG_CLOCK: s_memtime cycles / 100 MHz real-time ticks)
#   (b) the same binary on stimulus of 1 / 4 / 8 / 12 / 16 significant bits (the DVFS zero-data test of
#       MI355X_MICROARCH.md: if ms falls and the clock rises at constant instruction counts, the kernel is power-bound)
#   (c) tools/power_probe: the stream + MFMA envelope with no other work, and its MFMA-count / data-fill sweeps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/r2_fir255_clock.txt}
cd "$R"
mkdir -p "$(dirname "$OUT")"
{
  echo "== (a)+(b) product kernel, bench.py --workload W --stim-bits B, 20 steps; clock line from ACDSP_DEBUG_CLOCK=1 =="
  for w in fir255 fir255_dense; do
    for b in 1 4 8 12 16; do
      line=$(python bench.py --workload $w --stim-bits $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1)
      clk=$(ACDSP_DEBUG_CLOCK=1 python bench.py --workload $w --stim-bits $b --steps 4 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep 'shader clock' | tail -1)
      python - "$w" "$b" "$line" "$clk" <<'EOF'
import json, re, sys
w, b, line, clk = sys.argv[1:5]
j = json.loads(line)
m = re.search(r'shader clock ([0-9.]+) GHz', clk)
print("%-13s stim-bits %2s  kernel %.3f ms (min %.3f)  step %.3f ms  roofline.frac %.3f  shader clock %s GHz" % (
    w, b, j["roofline"]["kernel_ms_avg"], j["roofline"]["kernel_ms_min"], j["ms_per_step"], j["roofline"]["frac"], m.group(1) if m else "?"))
EOF
    done
  done
  echo
  echo "== (c) tools/power_probe (1024 ch x 2^20 samples, one single-wave workgroup per 64 steps, as the product kernel) =="
  tools/_bin/power_probe 1024 1048576 40
} 2>&1 | tee "$OUT"
