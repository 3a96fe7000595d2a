# tools/evidence_all.sh -- on the GPU box (about 12 minutes of box time): the whole evidence set of a round (tests, per-workload bench lines + rocprofv3, shape sweeps, default bench)
export ACDSP_ROUND=${ACDSP_ROUND:-r6}
bash tools/profile_all.sh fir255 fir255_dense fir255_wide fir1023 cic_dec cic_intr ddc polydec polyintr intgdump mvavg rtest_const_types rtest_load_types rtest_prog_types cic_dec_r7m2n4 cic_intr_r7m2n5 cic_dec_r64 > gpurun_out/profile_all.log 2>&1
{
echo "== tools/fir_shapes.py"; python tools/fir_shapes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/poly_shapes.py"; python tools/poly_shapes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/intr_shapes.py"; python tools/intr_shapes.py 2>&1 | grep -v amdgpu.ids
echo "== tools/cic_sweep.py"; python tools/cic_sweep.py large 2>&1 | grep -v amdgpu.ids
echo "== tools/misc_shapes.py"; python tools/misc_shapes.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/${ACDSP_ROUND}_shapes_body.txt
python bench.py 2>gpurun_out/bench_default.err | tail -1 > gpurun_out/${ACDSP_ROUND}_bench_default.json
tail -3 gpurun_out/gpu_tests.txt
