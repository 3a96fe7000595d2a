// tools/lds_conflict_probe.hip -- what SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE reads on accesses whose bank behaviour is known (gfx950).
// Five product kernels show 21 - 52 % "conflicts" in that ratio (profiles/r5_*_rocprof.txt); all of them move their data with 8- and 16-byte
// LDS operations.  Each kernel below is one shape of access, run by many waves (so that the counters are large against the noise): every lane
// reads (or writes) WIDTH bytes at byte address STRIDE * lane, eight instructions 1 KB apart per iteration.
//   WIDTH 4 / STRIDE 4, WIDTH 8 / STRIDE 8, WIDTH 16 / STRIDE 16: contiguous -- no two lanes of a pass share a bank: conflict-free by construction
//   WIDTH 4 / STRIDE 8, WIDTH 8 / STRIDE 16, WIDTH 16 / STRIDE 32: every other granule -- half the banks, 2-way conflicts by construction
//   hipcc --offload-arch=gfx950 -O3 tools/lds_conflict_probe.hip -o tools/_bin/lds_conflict_probe
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace ... (tools/lds_conflict_probe.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int WIDTH, int STRIDE, bool WRITE>
__global__ void __launch_bounds__(64) probe(int reps, unsigned *out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 64) { ((unsigned *)lds)[i] = i * 2654435761u; }
  __syncthreads();
  unsigned char *p = lds + STRIDE * threadIdx.x;
  unsigned acc = threadIdx.x;
  for (int r = 0; r < reps; r++) {
    asm volatile("" : "+v"(p));
#pragma unroll
    for (int k = 0; k < 8; k++) {
      unsigned char *q = p + (STRIDE * 64 * k) % (16384 - STRIDE * 64);
      if constexpr (WRITE) {
        if constexpr (WIDTH == 4) { *(unsigned *)q = acc; }
        else if constexpr (WIDTH == 8) { *(v2u *)q = (v2u){acc, acc}; }
        else { *(v4u *)q = (v4u){acc, acc, acc, acc}; }
      } else {
        if constexpr (WIDTH == 4) { acc += *(const unsigned *)q; }
        else if constexpr (WIDTH == 8) { const v2u v = *(const v2u *)q; acc += v.x ^ v.y; }
        else { const v4u v = *(const v4u *)q; acc += v.x ^ v.y ^ v.z ^ v.w; }
      }
    }
  }
  if (WRITE) { acc += ((unsigned *)lds)[threadIdx.x]; }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int WIDTH, int STRIDE, bool WRITE>
void run(const char *name, unsigned *d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 4096, reps = 200;
  hipLaunchKernelGGL((probe<WIDTH, STRIDE, WRITE>), dim3(blocks), dim3(64), 0, 0, reps, d_out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<WIDTH, STRIDE, WRITE>), dim3(blocks), dim3(64), 0, 0, reps, d_out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f ms  %6.2f ns per wave instruction and CU\n", name, ms, ms * 1e6 / ((double)blocks * reps * 8 / 256));
}

int main() {
  unsigned *d_out;
  hipMalloc(&d_out, 4096 * 64 * 4);
  run<4, 4, false>("read  b32  contiguous (stride 4)", d_out);
  run<4, 8, false>("read  b32  every other dword (stride 8)", d_out);
  run<8, 8, false>("read  b64  contiguous (stride 8)", d_out);
  run<8, 16, false>("read  b64  every other 8 bytes (stride 16)", d_out);
  run<16, 16, false>("read  b128 contiguous (stride 16)", d_out);
  run<16, 32, false>("read  b128 every other granule (stride 32)", d_out);
  run<4, 4, true>("write b32  contiguous (stride 4)", d_out);
  run<8, 8, true>("write b64  contiguous (stride 8)", d_out);
  run<8, 16, true>("write b64  every other 8 bytes (stride 16)", d_out);
  run<16, 16, true>("write b128 contiguous (stride 16)", d_out);
  run<16, 32, true>("write b128 every other granule (stride 32)", d_out);
  hipDeviceSynchronize();
  return 0;
}
