// tools/lds_align_probe.hip -- cost of LDS reads that are not aligned to their size (gfx950).  One wave; every lane reads WIDTH bytes at
// byte address STRIDE * lane + off, REPS times (dependent on nothing: the reads pipeline), cycles per read instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_align_probe.hip -o tools/_bin/lds_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int WIDTH>
__global__ void probe(int stride, int off, int reps, unsigned *out, long long *cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 64) { ((unsigned *)lds)[i] = i * 2654435761u; }
  __syncthreads();
  const unsigned char *p = lds + stride * threadIdx.x + off;
  unsigned acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    asm volatile("" : "+v"(p));   // the reads are loop-invariant otherwise
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if constexpr (WIDTH == 4) { unsigned v; __builtin_memcpy(&v, p + 16 * k, 4); acc += v; }
      else if constexpr (WIDTH == 8) { v2u v; __builtin_memcpy(&v, p + 16 * k, 8); acc += v.x ^ v.y; }
      else { v4u v; __builtin_memcpy(&v, p + 16 * k, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) { *cyc = t1 - t0; }
}
int main() {
  unsigned *d_out; long long *d_c;
  hipMalloc(&d_out, 256); hipMalloc(&d_c, 8);
  const int reps = 2000;
  for (int width : {4, 8, 16}) {
    for (int stride : {4, 8, 16, 32}) {
      if (stride < width) continue;
      for (int off : {0, 1, 2, 4, 6, 8}) {
        long long c = 0;
        for (int it = 0; it < 2; it++) {
          if (width == 4) hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, stride, off, reps, d_out, d_c);
          else if (width == 8) hipLaunchKernelGGL(probe<8>, 1, 64, 0, 0, stride, off, reps, d_out, d_c);
          else hipLaunchKernelGGL(probe<16>, 1, 64, 0, 0, stride, off, reps, d_out, d_c);
          hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
        }
        printf("width %2d B  lane stride %2d B  offset %d: %.1f cycles per wave read\n", width, stride, off, (double)c / (reps * 8));
      }
    }
  }
  return 0;
}
