// copy_probe2.hip -- round 3: where do the 0.6 TB/s between this repo's best copy (5.66 TB/s) and the guide's float4 copy
// (6.29 TB/s, MI355X_MICROARCH.md) go?  Not part of the product.
//   copy_probe2 [reps=10]
// One kernel family (one contiguous span per wave, 16 bytes per lane and instruction, batches of U loads followed by U stores)
// with every knob the product's streaming kernels could turn:
//   U        loads / stores per batch (burst length the memory controller sees from one wave)
//   LNT/SNT  non-temporal loads / stores
//   span     bytes one wave walks (active footprint = resident waves x span)
//   wg       threads per workgroup (64 = the FIR kernel's single-wave workgroups)
//   skew     byte offset of the destination relative to the source's alignment (DRAM bank / channel aliasing of the two streams)
//   map      0: wave w -> span w;  1: XCD-affine (block b runs on XCD b % 8 and takes span (b % 8) * nb / 8 + b / 8)
//   GiB      buffer size (256 MiB Infinity Cache: small buffers flatter)
// plus the guide-style one-element-per-thread float4 copy and hipMemcpyDtoD.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, bool LNT, bool SNT>
__global__ void __launch_bounds__(256) span_copy(const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec, long span_vec, int map) {
  const int wpb = blockDim.x >> 6;
  long b = blockIdx.x;
  if (map == 1) { const long nb8 = gridDim.x / 8; b = (b % 8) * nb8 + b / 8; }
  const long wave = b * wpb + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  for (long i = s + lane; i < e; i += 64 * U) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = LNT ? __builtin_nontemporal_load(x + i + 64 * u) : x[i + 64 * u]; }
#pragma unroll
    for (int u = 0; u < U; u++) { if (SNT) { __builtin_nontemporal_store(v[u], y + i + 64 * u); } else { y[i + 64 * u] = v[u]; } }
  }
}

// two-deep software pipeline: the loads of batch k+1 are issued before the stores of batch k (what the product kernels do)
template <int U, bool LNT, bool SNT>
__global__ void __launch_bounds__(256) span_copy_pipe(const v4i *__restrict__ x, v4i *__restrict__ y, long n_vec, long span_vec, int map) {
  const int wpb = blockDim.x >> 6;
  long b = blockIdx.x;
  if (map == 1) { const long nb8 = gridDim.x / 8; b = (b % 8) * nb8 + b / 8; }
  const long wave = b * wpb + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  if (s >= e) { return; }
  v4i v[U], w[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = LNT ? __builtin_nontemporal_load(x + s + lane + 64 * u) : x[s + lane + 64 * u]; }
  for (long i = s + lane; i < e; i += 64 * U) {
    const long nx = i + 64 * U < e ? i + 64 * U : i;
#pragma unroll
    for (int u = 0; u < U; u++) { w[u] = LNT ? __builtin_nontemporal_load(x + nx + 64 * u) : x[nx + 64 * u]; }
#pragma unroll
    for (int u = 0; u < U; u++) { if (SNT) { __builtin_nontemporal_store(v[u], y + i + 64 * u); } else { y[i + 64 * u] = v[u]; } }
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = w[u]; }
  }
}

__global__ void __launch_bounds__(256) elem_copy(const float4 *__restrict__ x, float4 *__restrict__ y, long n_vec) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n_vec) { y[i] = x[i]; }
}

struct Ctx { char *x, *y; long bytes; int reps; };

template <typename F>
static float time_it(const Ctx &c, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; w++) { launch(); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < c.reps; r++) { launch(); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms / c.reps;
}

template <int U, bool LNT, bool SNT, bool PIPE = false>
static void run(const Ctx &c, long gib4, long span_kb, int wg, long skew, int map) {
  const long bytes = gib4 * (1L << 28);   // in quarters of a GiB
  const long n_vec = bytes / 16, span_vec = span_kb * 64;
  const long waves = (n_vec + span_vec - 1) / span_vec;
  const int wpb = wg / 64;
  long nb = (waves + wpb - 1) / wpb;
  if (map == 1) { nb = (nb + 7) / 8 * 8; }
  const v4i *x = (const v4i *)c.x;
  v4i *y = (v4i *)(c.y + skew);
  const float ms = time_it(c, [&]() {
    if (PIPE) { hipLaunchKernelGGL((span_copy_pipe<U, LNT, SNT>), dim3((unsigned)nb), dim3(wg), 0, 0, x, y, n_vec, span_vec, map); }
    else { hipLaunchKernelGGL((span_copy<U, LNT, SNT>), dim3((unsigned)nb), dim3(wg), 0, 0, x, y, n_vec, span_vec, map); }
  });
  printf("span-copy%s U=%2d lnt=%d snt=%d buf=%5.2f GiB span=%4ld KB wg=%3d skew=%8ld map=%d  %8.3f ms  %6.2f TB/s\n", PIPE ? "-pipe" : "     ", U, (int)LNT,
         (int)SNT, bytes / 1073741824.0, span_kb, wg, skew, map, ms, 2.0 * bytes / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  Ctx c;
  c.reps = argc > 1 ? atoi(argv[1]) : 10;
  c.bytes = 4L << 30;
  const long slack = 64L << 20;
  CK(hipMalloc((void **)&c.x, c.bytes + slack));
  CK(hipMalloc((void **)&c.y, c.bytes + slack));
  CK(hipMemset(c.x, 1, c.bytes + slack));
  CK(hipMemset(c.y, 2, c.bytes + slack));
  printf("# copy_probe2: src %p dst %p\n", (void *)c.x, (void *)c.y);

  printf("# --- references: guide-style one float4 per thread, and the runtime's own device copy ---\n");
  for (long q : {1L, 4L, 8L, 16L}) {
    const long bytes = q << 28, n_vec = bytes / 16;
    const float ms = time_it(c, [&]() { hipLaunchKernelGGL(elem_copy, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, 0, (const float4 *)c.x, (float4 *)c.y, n_vec); });
    printf("elem-copy  buf=%5.2f GiB  %8.3f ms  %6.2f TB/s\n", bytes / 1073741824.0, ms, 2.0 * bytes / ms / 1e9);
    const float ms2 = time_it(c, [&]() { CK(hipMemcpyAsync(c.y, c.x, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpy  buf=%5.2f GiB  %8.3f ms  %6.2f TB/s\n", bytes / 1073741824.0, ms2, 2.0 * bytes / ms2 / 1e9);
  }

  printf("# --- batch length U x non-temporal, 2 GiB buffers (the size of config 2's streams), span 16 KB, wg 256 ---\n");
  run<4, false, false>(c, 8, 16, 256, 0, 0);  run<8, false, false>(c, 8, 16, 256, 0, 0);  run<16, false, false>(c, 8, 16, 256, 0, 0);
  run<8, true, false>(c, 8, 16, 256, 0, 0);   run<8, false, true>(c, 8, 16, 256, 0, 0);   run<8, true, true>(c, 8, 16, 256, 0, 0);
  run<16, true, true>(c, 8, 16, 256, 0, 0);   run<16, true, false>(c, 8, 16, 256, 0, 0);
  run<2, false, false>(c, 8, 16, 256, 0, 0);  run<2, true, true>(c, 8, 16, 256, 0, 0);    run<4, true, true>(c, 8, 16, 256, 0, 0);

  printf("# --- span (active footprint), U 8 ---\n");
  for (long sp : {8L, 16L, 32L, 64L, 128L, 512L}) { run<8, false, false>(c, 8, sp, 256, 0, 0); }
  for (long sp : {8L, 32L, 128L}) { run<8, true, true>(c, 8, sp, 256, 0, 0); }

  printf("# --- workgroup size (64 = single-wave workgroups, the FIR kernel's geometry), span 16 and 128 KB ---\n");
  for (int wg : {64, 128, 512, 1024}) { run<8, false, false>(c, 8, 16, wg, 0, 0); }
  run<8, false, false>(c, 8, 128, 64, 0, 0);  run<8, true, true>(c, 8, 128, 64, 0, 0);  run<2, true, true>(c, 8, 128, 64, 0, 0);

  printf("# --- destination skew (bank / channel aliasing of the read and the write stream), U 8, span 16 KB ---\n");
  for (long sk : {256L, 1024L, 4096L, 16384L, 65536L, 262144L, 1048576L, 2097152L + 4096L, 33554432L + 8192L}) { run<8, false, false>(c, 8, 16, 256, sk, 0); }

  printf("# --- XCD-affine span assignment ---\n");
  run<8, false, false>(c, 8, 16, 256, 0, 1);  run<8, false, false>(c, 8, 128, 64, 0, 1);  run<8, true, true>(c, 8, 16, 256, 0, 1);

  printf("# --- buffer size (Infinity Cache 256 MiB) ---\n");
  for (long q : {1L, 2L, 4L, 16L}) { run<8, false, false>(c, q, 16, 256, 0, 0); }

  printf("# --- software-pipelined (loads of batch k+1 in front of the stores of batch k) ---\n");
  run<4, false, false, true>(c, 8, 16, 256, 0, 0);  run<8, false, false, true>(c, 8, 16, 256, 0, 0);  run<8, true, true, true>(c, 8, 16, 256, 0, 0);
  run<2, true, true, true>(c, 8, 128, 64, 0, 0);    run<8, true, true, true>(c, 8, 128, 64, 0, 0);   run<16, true, true, true>(c, 8, 128, 64, 0, 0);

  printf("# --- drift check: first row again ---\n");
  run<8, false, false>(c, 8, 16, 256, 0, 0);
  return 0;
}
