#!/bin/bash
# tools/profile_all.sh [workloads...] -- on the GPU box: full GPU test suite, one bench line per workload (with the CPU
# baseline where bench.py has one) and the rocprofv3 kernel-trace + PMC evidence; everything lands in gpurun_out/
# (copy the *_rocprof.txt / ${RND}_bench_*.json you want judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RND=${ACDSP_ROUND:-r2}
cd "$R"
mkdir -p gpurun_out
WL=${*:-fir255 fir255_dense fir255_wide fir1023 cic_dec cic_intr ddc polydec polyintr intgdump mvavg}
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/gpu_tests.txt 2>&1
for w in $WL; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-secondary 2> gpurun_out/bench_$w.err | tail -1 > gpurun_out/${RND}_bench_$w.json
  bash tools/prof.sh ${RND}_$w --workload $w > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/prof_${RND}_$w gpurun_out/${RND}_${w}_rocprof.txt > /dev/null 2>&1
  rm -rf gpurun_out/prof_${RND}_$w/*.db gpurun_out/prof_${RND}_$w/*/
done
cat gpurun_out/gpu_tests.txt
for w in $WL; do cut -c1-160 gpurun_out/${RND}_bench_$w.json; done
