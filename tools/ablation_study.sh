#!/bin/bash
# tools/ablation_study.sh -- on the GPU box: the 255-tap pipelined kernel with parts of its step removed (timing only, wrong
# results; builds made by `tools/ab_build.sh abl_X fir_mfma.hip -DACDSP_ABL_X`), kernel ms + the shader clock its waves saw.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
echo "== (d) ablations of fir_mfma_pipe_body (fir255, 20 steps; clock from ACDSP_DEBUG_CLOCK=1; results are wrong by construction) =="
for v in full BREAD STAGE LOAD SL EMIT FLUSH EF ALL; do
  lib=""; [ $v != full ] && lib=ac_dsp_amd/lib/libacdsp_abl_$v.so
  line=$(ACDSP_LIB=$lib python bench.py --workload fir255 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  clk=$(ACDSP_LIB=$lib ACDSP_DEBUG_CLOCK=1 python bench.py --workload fir255 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 >/dev/null | grep 'shader clock' | tail -1)
  python - "$v" "$line" "$clk" <<'PY'
import json, re, sys
v, line, clk = sys.argv[1:4]
names = {"full": "product kernel", "BREAD": "B-fragment LDS reads: first group only", "STAGE": "no byte-plane staging (perm + ds_write)",
         "LOAD": "no global loads in the loop", "SL": "no loads, no staging", "EMIT": "no epilogue (conversion + tile writes)",
         "FLUSH": "no global stores (tile reads + stores)", "EF": "no epilogue, no stores", "ALL": "MFMAs + B-fragment reads only"}
j = json.loads(line)
m = re.search(r'shader clock ([0-9.]+) GHz', clk)
print("%-46s kernel %.3f ms (min %.3f)  shader clock %s GHz" % (names[v], j["roofline"]["kernel_ms_avg"], j["roofline"]["kernel_ms_min"], m.group(1) if m else "?"))
PY
done
