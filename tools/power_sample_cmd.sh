#!/bin/bash
# tools/power_sample_cmd.sh <tag> <seconds-to-skip> -- <command...>: socket power and shader clock sampled with rocm-smi every 0.2 s
# while <command> runs (on the GPU box).  Output: gpurun_out/power_<tag>.txt
TAG=$1; SKIP=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/power_$TAG.txt
{
echo "# tools/power_sample_cmd.sh $TAG: rocm-smi samples around '$*'"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power"
} > $OUT
"$@" > gpurun_out/power_cmd_$TAG.log 2>&1 &
BP=$!
sleep $SKIP
i=0
while kill -0 $BP 2>/dev/null && [ $i -lt 60 ]; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "package power\|sclk" | tr '\n' ' ' | sed "s/^/t=$i /" >> $OUT; echo >> $OUT
  i=$((i+1)); sleep 0.2
done
wait $BP
tail -2 gpurun_out/power_cmd_$TAG.log | sed 's/^/# /' >> $OUT
