#!/bin/bash
# tools/power_sample.sh [workload=fir255] [steps=4000] -- on the GPU box: socket power, power cap and shader clock sampled with
# rocm-smi every 0.2 s while bench.py loops one workload (idle samples before and after); evidence for DESIGN 5.2's
# "power-bound" reading.  Output: gpurun_out/power_<workload>.txt
W=${1:-fir255}; STEPS=${2:-4000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/power_$W.txt
{
echo "# tools/power_sample.sh $W $STEPS: rocm-smi samples around 'bench.py --workload $W --steps $STEPS'"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" 
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | sed 's/^/idle-before: /'
} > $OUT
python bench.py --workload $W --steps $STEPS --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/power_bench_$W.json 2>/dev/null &
BP=$!
sleep 6     # import + workload build + stimulus
i=0
while kill -0 $BP 2>/dev/null && [ $i -lt 80 ]; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "package power\|sclk" | tr '\n' ' ' | sed "s/^/t=$i /" >> $OUT; echo >> $OUT
  i=$((i+1)); sleep 0.2
done
wait $BP
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | sed 's/^/idle-after: /' >> $OUT
python -c "
import json; d=json.load(open('gpurun_out/power_bench_$W.json'))
print('# bench: ms_per_step %.4f  kernel_ms_avg %.4f  roofline.frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))" >> $OUT
cat $OUT
