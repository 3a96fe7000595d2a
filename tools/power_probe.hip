// power_probe.hip -- what the MI355X lets a 255-tap int16 FIR reach: the envelope of "stream 2 B in + 2 B out
// per sample while issuing NM int8 32x32x32 MFMAs per 1024 samples" with (almost) no other work, on the
// launch geometry of fir_mfma_kernel (one single-wave workgroup per channel x time chunk, 2 waves per SIMD).
// Not part of the product.  Each line reports wall ms per launch (HIP events), shader cycles per 1024-sample
// step and wave, and the shader clock the waves saw (s_memtime cycles / 100 MHz s_memrealtime ticks).
//
//   power_probe [n_ch=1024] [n=1048576] [reps=5]
//
// Rows (NM = MFMAs per step; DATA = operand fill):
//   mfma-only    NM=28/36   DATA zero / small / random      no loads, no stores
//   stream-only  NM=0                                       loads + stores (the copy ceiling of this geometry)
//   stream+mfma  NM=8..36   DATA random                     loads feed the B operands, stores carry accumulator bits
// The gap between stream+mfma(28) and the real kernel is what staging, fragment reads and the epilogue cost;
// the gap between stream+mfma(28) and 0.767 ms (= 70 % of 8 TB/s) is what no kernel of this formulation can close.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kSteps = 64;   // 1024-sample steps per wave (as the product kernel at config 2)

// kU: load ring (loads run kU-1 steps ahead; must divide kSteps); OCC: waves per SIMD the register budget is held to
template <int NM, bool LOADS, bool STORES, int kU = 4, int OCC = 2, int AREUSE = 2>
__global__ void __launch_bounds__(64, OCC) probe(const v4i *__restrict__ frag, const short *__restrict__ x, short *__restrict__ y,
                                                long stride, long *__restrict__ dbg, int sink_flag) {
  static_assert(kSteps % kU == 0, "ring depth must divide the steps per wave");
  const int lane = threadIdx.x;
  const long ch = blockIdx.y, chunk = blockIdx.x;
  v4i A[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { A[i] = frag[i * 64 + lane]; }
  const char *xr = (const char *)(x + ch * stride + chunk * (long)kSteps * 1024) + lane * 16;
  char *yr = (char *)(y + ch * stride + chunk * (long)kSteps * 1024) + lane * 16;
  v4i R[kU][2];
#pragma unroll
  for (int u = 0; u < kU; u++) { R[u][0] = frag[(8 + u) * 64 + lane]; R[u][1] = frag[(12 + u) * 64 + lane]; }
  if (LOADS) {
#pragma unroll
    for (int u = 0; u < kU - 1; u++) {
      R[u][0] = *(const v4i *)(xr + u * 2048);
      R[u][1] = *(const v4i *)(xr + u * 2048 + 1024);
    }
  }
  v16i acc[4] = {{0}, {0}, {0}, {0}};
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  v4i o0 = R[0][0], o1 = R[0][1];
  for (int s0 = 0; s0 < kSteps; s0 += kU) {
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const int s = s0 + u;
      // program order per step: stores of step s-1, loads of step s+kU-1, MFMAs of step s -- so the wait in front of
      // the MFMAs is a counted vmcnt that leaves the younger stores and loads in flight
      if (STORES && s > 0) {
        *(v4i *)(yr + (s - 1) * 2048) = o0;
        *(v4i *)(yr + (s - 1) * 2048 + 1024) = o1;
      }
      if (LOADS) {   // fetch step s + kU - 1 into the slot consumed last step (past the chunk: stay on the last step)
        const int sn = s + kU - 1 < kSteps ? s + kU - 1 : kSteps - 1;
        R[(u + kU - 1) % kU][0] = *(const v4i *)(xr + sn * 2048);
        R[(u + kU - 1) % kU][1] = *(const v4i *)(xr + sn * 2048 + 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM; i++) {
        // AREUSE consecutive MFMAs share the A operand (register-blocking depth of the product kernels); B walks the ring
        acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(i / AREUSE) & 3], R[AREUSE == 2 ? u : (u + (i >> 1)) % kU][i & 1], acc[i & 3], 0, 0, 0);
      }
      if (NM > 0) {
        o0 = (v4i){acc[0][0], acc[1][1], acc[2][2], acc[3][3]};
        o1 = (v4i){acc[0][4], acc[1][5], acc[2][6], acc[3][7]};
      } else {
        o0 = R[u][0]; o1 = R[u][1];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STORES) {
    *(v4i *)(yr + (kSteps - 1) * 2048) = o0;
    *(v4i *)(yr + (kSteps - 1) * 2048 + 1024) = o1;
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  int t = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { t += acc[i][0] + acc[i][7] + acc[i][15]; }
#pragma unroll
  for (int u = 0; u < kU; u++) { t += R[u][0].x + R[u][1].w; }
  if (sink_flag == 12345 + t) { y[lane] = (short)t; }   // keeps everything live
  if (lane == 0) {
    const long w = blockIdx.y * (long)gridDim.x + blockIdx.x;
    dbg[2 * w] = (long)(c1 - c0);
    dbg[2 * w + 1] = (long)(r1 - r0);
  }
}

struct Ctx;
// write-only stream with W-byte stores per lane (W = 4, 8, 16), a wave's stores forming one contiguous run: what a
// store-bound kernel (the interpolators: 8 output words per input word) can expect from narrower stores
template <int W>
__global__ void __launch_bounds__(256) wprobe(char *__restrict__ y, long bytes_per_block) {
  char *base = y + (long)blockIdx.x * bytes_per_block;
  const int tid = threadIdx.x;
  for (long off = 0; off < bytes_per_block; off += 256 * W * 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      char *p = base + off + u * 256 * W + tid * W;
      if (W == 16) { *(v4i *)p = (v4i){tid, u, 3, 4}; }
      else if (W == 8) { *(long *)p = (long)tid * 77 + u; }
      else { *(int *)p = tid + u; }
    }
  }
}

static uint64_t sm64(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Ctx {
  v4i *d_frag; short *d_x, *d_y; long *d_dbg; long stride; int n_ch; long n; int reps;
  std::vector<long> h_dbg;
};

template <int W> static void run_w(Ctx &c, size_t bytes) {
  const long per_block = 1 << 20;
  const unsigned blocks = (unsigned)(bytes / per_block);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(wprobe<W>, dim3(blocks), dim3(256), 0, 0, (char *)c.d_y, per_block);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL(wprobe<W>, dim3(blocks), dim3(256), 0, 0, (char *)c.d_y, per_block); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  printf("write-only, %2d-byte stores per lane (256-thread blocks, 1 MiB per block): %7.3f ms  %6.0f GB/s\n", W, ms, bytes / (ms * 1e-3) / 1e9);
}

template <int NM, bool LOADS, bool STORES, int kU = 4, int OCC = 2, int AREUSE = 2>
static void run(Ctx &c, const char *label, int data) {
  // operand fill: 0 = zeros, 1 = small (|v| < 4, like the high-byte plane of a low-pass set), 2 = random bytes
  std::vector<uint32_t> h(16 * 64 * 4);
  uint64_t seed = 7;
  for (auto &w : h) {
    uint64_t z = sm64(seed);
    if (data == 0) { w = 0; }
    else if (data == 1) { w = (uint32_t)(z & 0x03030303u); }
    else { w = (uint32_t)z; }
  }
  CK(hipMemcpy(c.d_frag, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const dim3 grid((unsigned)(c.n / (1024L * kSteps)), (unsigned)c.n_ch);
  const size_t lds = (160 * 1024) / (4 * OCC) - 256;   // dynamic LDS request that caps the residency at OCC waves per SIMD
  const long n_waves = (long)grid.x * grid.y;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) { hipLaunchKernelGGL((probe<NM, LOADS, STORES, kU, OCC, AREUSE>), grid, dim3(64), lds, 0, c.d_frag, c.d_x, c.d_y, c.stride, c.d_dbg, 0); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < c.reps; r++) { hipLaunchKernelGGL((probe<NM, LOADS, STORES, kU, OCC, AREUSE>), grid, dim3(64), lds, 0, c.d_frag, c.d_x, c.d_y, c.stride, c.d_dbg, 0); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= c.reps;
  CK(hipMemcpy(c.h_dbg.data(), c.d_dbg, n_waves * 16, hipMemcpyDeviceToHost));
  double sc = 0, sr = 0;
  for (long i = 0; i < n_waves; i++) { sc += (double)c.h_dbg[2 * i]; sr += (double)c.h_dbg[2 * i + 1]; }
  const double samples = (double)c.n_ch * (double)c.n;
  const double gbs = ((LOADS ? 2.0 : 0.0) + (STORES ? 2.0 : 0.0)) * samples / (ms * 1e-3) / 1e9;
  const double tops = 2.0 * NM * 32768.0 * (samples / 1024.0) / (ms * 1e-3) / 1e12;
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<NM, LOADS, STORES, kU, OCC, AREUSE>, 64, lds));
  printf("%-14s NM=%2d ahead=%d waves/SIMD=%d data=%-6s  %7.3f ms  %7.0f cyc/step  clock %.3f GHz  %6.0f GB/s  %6.0f TOP/s\n", label, NM, kU - 1, occ / 4,
         data == 0 ? "zero" : data == 1 ? "small" : "random", ms, sc / n_waves / kSteps, sc / sr * 0.1, gbs, tops);
  fflush(stdout);
}

int main(int argc, char **argv) {
  Ctx c;
  c.n_ch = argc > 1 ? atoi(argv[1]) : 1024;
  c.n = argc > 2 ? atol(argv[2]) : (1L << 20);
  c.reps = argc > 3 ? atoi(argv[3]) : 5;
  c.stride = c.n;
  const size_t bytes = (size_t)c.n_ch * c.stride * 2;
  CK(hipMalloc((void **)&c.d_x, bytes));
  CK(hipMalloc((void **)&c.d_y, bytes));
  CK(hipMalloc((void **)&c.d_frag, 16 * 64 * 16));
  const long n_waves = (long)c.n_ch * (c.n / (1024L * kSteps));
  CK(hipMalloc((void **)&c.d_dbg, n_waves * 16));
  c.h_dbg.resize(2 * n_waves);
  {   // random int16 stimulus
    std::vector<uint64_t> h(1 << 20);
    uint64_t seed = 0xACD5;
    for (auto &w : h) { w = sm64(seed); }
    for (size_t off = 0; off < bytes; off += h.size() * 8) {
      const size_t nb = bytes - off < h.size() * 8 ? bytes - off : h.size() * 8;
      CK(hipMemcpy((char *)c.d_x + off, h.data(), nb, hipMemcpyHostToDevice));
    }
  }
  printf("# power_probe: %d ch x %ld samples, %ld single-wave workgroups, %d timed launches per row\n", c.n_ch, c.n, n_waves, c.reps);
  printf("# target: 70 %% of 8 TB/s on 4 B/sample = %.3f ms per launch\n", 4.0 * c.n_ch * c.n / 5.6e12 * 1e3);
  if (argc > 4 && atoi(argv[4]) == 1023) {   // envelope of the 1023-tap kernel: 76 MFMAs per step with the config-4 set's band skip, 132 dense
    printf("# A-operand reuse sweep (consecutive MFMAs sharing A): 1, 4, 8, 76\n");
    run<76, false, false, 4, 2, 1>(c, "mfma-only", 2);
    run<76, false, false, 4, 2, 4>(c, "mfma-only", 2);
    run<76, false, false, 4, 2, 8>(c, "mfma-only", 2);
    run<76, false, false, 4, 2, 76>(c, "mfma-only", 2);
    printf("# default reuse 2\n");
    run<76, false, false>(c, "mfma-only", 0);
    run<76, false, false>(c, "mfma-only", 2);
    run<132, false, false>(c, "mfma-only", 2);
    run<76, true, true, 4, 2>(c, "stream+mfma", 2);
    run<76, true, true, 4, 4>(c, "stream+mfma", 2);
    run<132, true, true, 4, 2>(c, "stream+mfma", 2);
    return 0;
  }
  run<28, false, false>(c, "mfma-only", 0);
  run<28, false, false>(c, "mfma-only", 1);
  run<28, false, false>(c, "mfma-only", 2);
  run<36, false, false>(c, "mfma-only", 2);
  run<0, true, true, 4, 2>(c, "stream-only", 2);
  run<0, true, true, 4, 4>(c, "stream-only", 2);
  run<0, true, true, 8, 4>(c, "stream-only", 2);
  run<0, true, true, 8, 8>(c, "stream-only", 2);
  run<0, true, false, 8, 8>(c, "read-only", 2);
  run<0, false, true, 4, 8>(c, "write-only", 2);
  // the envelope: 28 MFMAs per step beside the stream, pipeline depth x occupancy
  run<28, true, true, 2, 2>(c, "stream+mfma", 2);
  run<28, true, true, 4, 2>(c, "stream+mfma", 2);
  run<28, true, true, 8, 2>(c, "stream+mfma", 2);
  run<28, true, true, 2, 3>(c, "stream+mfma", 2);
  run<28, true, true, 4, 3>(c, "stream+mfma", 2);
  run<28, true, true, 8, 3>(c, "stream+mfma", 2);
  run<28, true, true, 2, 4>(c, "stream+mfma", 2);
  run<28, true, true, 4, 4>(c, "stream+mfma", 2);
  // MFMA count and operand fill at the best geometry found above (edit if another one wins)
  run<8, true, true, 4, 4>(c, "stream+mfma", 2);
  run<16, true, true, 4, 4>(c, "stream+mfma", 2);
  run<24, true, true, 4, 4>(c, "stream+mfma", 2);
  run<36, true, true, 4, 4>(c, "stream+mfma", 2);
  run<28, true, true, 4, 4>(c, "stream+mfma", 1);
  run<28, true, true, 4, 4>(c, "stream+mfma", 0);
  run<28, true, false, 4, 4>(c, "read+mfma", 2);
  run<28, false, true, 4, 4>(c, "write+mfma", 2);
  run_w<16>(c, bytes); run_w<8>(c, bytes); run_w<4>(c, bytes);
  return 0;
}
