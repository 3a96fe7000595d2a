// tools/fill_probe.hip -- what the memory system gives WRITE-dominant streams (the interpolator rows: ac_cic_intr_full R=8 reads 4 B and
// writes 64 B per input sample, ac_poly_intr IF=8 reads 2 B and writes 16 B).  Bare kernels, no arithmetic:
//   fill      every thread stores one 16-byte piece (plain / non-temporal)
//   elem-expand / elem-reduce: the same one-piece-per-thread geometry at 1 : R and R : 1
//   expand    a wave reads 1 KB and writes it R times to R consecutive KB (read : write = 1 : R), spans of `span` KB per wave in memory order
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fill_probe tools/fill_probe.hip ; run on the GPU box.  Prints TB/s of read + written bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) fill_kernel(v4i *dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const v4i v = {(int)i, 1, 2, 3};
    if (NT) { __builtin_nontemporal_store(v, dst + i); } else { dst[i] = v; }
  }
}

// one wave per block; block b handles `pieces` input KB starting at b * pieces
template <int R, bool NT>
__global__ void __launch_bounds__(64) expand_kernel(const v4i *src, v4i *dst, int pieces) {
  const int lane = threadIdx.x;
  const size_t p0 = (size_t)blockIdx.x * pieces;
  v4i cur = src[p0 * 64 + lane];
  for (int p = 0; p < pieces; p++) {
    const v4i nxt = src[(p0 + (p + 1 < pieces ? p + 1 : p)) * 64 + lane];
    v4i *d = dst + (p0 + p) * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; r++) {
      v4i v = cur; v.x += r;
      if (NT) { __builtin_nontemporal_store(v, d + 64 * r); } else { d[64 * r] = v; }
    }
    cur = nxt;
  }
}

// one 16-byte piece per thread, 256 threads: the workgroup reads 4 KB and writes R x 4 KB, contiguous
template <int R, bool NT>
__global__ void __launch_bounds__(256) elem_expand_kernel(const v4i *src, v4i *dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const v4i cur = src[i];
  v4i *d = dst + (size_t)blockIdx.x * 256 * R + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; r++) {
    v4i v = cur; v.x += r;
    if (NT) { __builtin_nontemporal_store(v, d + 256 * r); } else { d[256 * r] = v; }
  }
}

// read-dominant: the workgroup reads R x 4 KB (contiguous, R loads per thread) and writes 4 KB (read : write = R : 1)
template <int R, bool NT>
__global__ void __launch_bounds__(256) elem_reduce_kernel(const v4i *src, v4i *dst) {
  const v4i *s0 = src + (size_t)blockIdx.x * 256 * R + threadIdx.x;
  v4i acc = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < R; r++) { acc += s0[256 * r]; }
  if (NT) { __builtin_nontemporal_store(acc, dst + (size_t)blockIdx.x * 256 + threadIdx.x); } else { dst[(size_t)blockIdx.x * 256 + threadIdx.x] = acc; }
}

// span geometry of the product kernels: one wave per workgroup reads `pieces` KB contiguous (U loads in flight), writes pieces / R KB;
// workgroups in memory order; LNT / SNT: non-temporal loads / stores; XCD: workgroup L works on span (L % 8) * (n / 8) + L / 8
template <int R, int U, bool LNT, bool SNT, bool XCD>
__global__ void __launch_bounds__(64) span_reduce_kernel(const v4i *src, v4i *dst, int pieces) {
  const int lane = threadIdx.x;
  size_t blk = blockIdx.x;
  if (XCD) { const size_t T = gridDim.x; blk = (blk & 7) * (T >> 3) + (blk >> 3); }
  const v4i *s0 = src + blk * pieces * 64 + lane;
  v4i *d0 = dst + blk * (pieces / R) * 64 + lane;
  for (int p = 0; p < pieces; p += U) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = LNT ? __builtin_nontemporal_load(s0 + (p + u) * 64) : s0[(p + u) * 64]; }
#pragma unroll
    for (int o = 0; o < U / R; o++) {
      v4i acc = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < R; r++) { acc += v[o * R + r]; }
      if (SNT) { __builtin_nontemporal_store(acc, d0 + (p / R + o) * 64); } else { d0[(p / R + o) * 64] = acc; }
    }
  }
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) { f(); }
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) { f(); }
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const size_t out_bytes = (size_t)4 << 30;
  v4i *src, *dst;
  CK(hipMalloc(&src, out_bytes)); CK(hipMalloc(&dst, out_bytes));
  CK(hipMemset(src, 1, out_bytes)); CK(hipMemset(dst, 0, out_bytes));
  const size_t n = out_bytes / 16;
  for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(fill_kernel<false>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, dst, n); }   // clocks
  float ms = time_ms([&] { hipLaunchKernelGGL(fill_kernel<false>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, dst, n); }, reps);
  printf("fill plain        4 GiB  %.3f ms  %.2f TB/s\n", ms, out_bytes / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(fill_kernel<true>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, dst, n); }, reps);
  printf("fill non-temporal 4 GiB  %.3f ms  %.2f TB/s\n", ms, out_bytes / ms / 1e9);
#define EXPAND(R_, NT_, PIECES_)                                                                                              \
  {                                                                                                                           \
    const size_t in_kb = out_bytes / 1024 / R_;                                                                               \
    ms = time_ms([&] { hipLaunchKernelGGL((expand_kernel<R_, NT_>), dim3((unsigned)(in_kb / PIECES_)), dim3(64), 0, 0, src, dst, PIECES_); }, reps); \
    printf("expand 1:%-2d %s  %2d KB of input per wave  %.3f ms  %.2f TB/s (read + written)\n", R_, NT_ ? "nt   " : "plain", PIECES_, ms, \
           (out_bytes + out_bytes / R_) / ms / 1e9);                                                                          \
  }
  EXPAND(8, false, 1) EXPAND(8, false, 4) EXPAND(8, false, 16) EXPAND(8, true, 1) EXPAND(8, true, 4) EXPAND(8, true, 16)
  EXPAND(16, false, 1) EXPAND(16, false, 4) EXPAND(16, true, 1) EXPAND(16, true, 4)
  EXPAND(1, false, 16) EXPAND(1, true, 16)
#define ELEM(R_, NT_)                                                                                                         \
  {                                                                                                                           \
    const size_t in_pieces = out_bytes / 16 / R_;                                                                             \
    ms = time_ms([&] { hipLaunchKernelGGL((elem_expand_kernel<R_, NT_>), dim3((unsigned)(in_pieces / 256)), dim3(256), 0, 0, src, dst); }, reps); \
    printf("elem-expand 1:%-2d %s  (256-thread workgroups, 16 B per thread)  %.3f ms  %.2f TB/s (read + written)\n", R_, NT_ ? "nt   " : "plain", ms, \
           (out_bytes + out_bytes / R_) / ms / 1e9);                                                                          \
  }
  ELEM(1, false) ELEM(1, true) ELEM(2, false) ELEM(4, false) ELEM(8, false) ELEM(8, true) ELEM(16, false) ELEM(16, true)
#define REDUCE(R_, NT_)                                                                                                       \
  {                                                                                                                           \
    const size_t out_pieces = out_bytes / 16 / R_;                                                                            \
    ms = time_ms([&] { hipLaunchKernelGGL((elem_reduce_kernel<R_, NT_>), dim3((unsigned)(out_pieces / 256)), dim3(256), 0, 0, src, dst); }, reps); \
    printf("elem-reduce %2d:1 %s  (256-thread workgroups, 16 B per thread and load)  %.3f ms  %.2f TB/s (read + written)\n", R_, NT_ ? "nt   " : "plain", ms, \
           (out_bytes + out_bytes / R_) / ms / 1e9);                                                                          \
  }
  REDUCE(2, false) REDUCE(4, false) REDUCE(4, true) REDUCE(8, false) REDUCE(8, true) REDUCE(16, false) REDUCE(16, true) REDUCE(32, false)
#define SPANR(R_, U_, LNT_, SNT_, XCD_, PIECES_)                                                                              \
  {                                                                                                                           \
    const size_t in_kb = out_bytes / 1024;                                                                                    \
    ms = time_ms([&] { hipLaunchKernelGGL((span_reduce_kernel<R_, U_, LNT_, SNT_, XCD_>), dim3((unsigned)(in_kb / PIECES_)), dim3(64), 0, 0, src, dst, PIECES_); }, reps); \
    printf("span-reduce %2d:1 U=%2d lnt=%d snt=%d xcd=%d  %2d KB per wave  %.3f ms  %.2f TB/s (read + written)\n", R_, U_, LNT_, SNT_, XCD_, PIECES_, ms, \
           (out_bytes + out_bytes / R_) / ms / 1e9);                                                                          \
  }
  SPANR(8, 8, false, false, false, 16) SPANR(8, 8, false, true, false, 16) SPANR(8, 8, true, true, false, 16) SPANR(8, 16, false, true, false, 16)
  SPANR(8, 16, true, true, false, 16) SPANR(8, 16, false, true, false, 32) SPANR(8, 16, true, true, false, 32) SPANR(8, 16, false, true, true, 16)
  SPANR(8, 16, true, true, true, 16) SPANR(8, 16, true, true, true, 32) SPANR(8, 8, false, true, false, 8)
  SPANR(4, 16, false, true, false, 16) SPANR(4, 16, true, true, false, 16) SPANR(4, 16, true, true, true, 16)
  return 0;
}
