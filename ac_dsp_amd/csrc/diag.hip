// diag.hip -- measurement kernels behind acdsp_diag_* (include/acdsp.h): what bench.py prints beside the roofline of the FIR rows.
// They are NOT part of the filter path and compute nothing a caller could use; they exist so that the two reference speeds of
// SURVEY 8(d) / DESIGN 5.2 are timed in the SAME process, on the same buffers and at the same settled clocks as the product kernel:
//
//   * copy      -- the plainest device copy there is: one 16-byte element per thread, 256-thread workgroups dispatched in memory
//                  order ("measured device-copy bandwidth", the guide's float4 copy: ~6.3 TB/s).
//   * envelope  -- the power / issue envelope of the exact int8-split formulation of a FIR row: stream 2 B in + 2 B out per sample in
//                  the best streaming geometry (4-wave workgroups, 32 KB spans in memory order, bursts of 8 non-temporal loads then
//                  8 non-temporal stores) and issue NM v_mfma_i32_32x32x32_i8 per 1024 samples, NH of them on high-byte-plane
//                  fragments -- with NOTHING else in the loop: no byte-plane split, no LDS, no epilogue.  A operands: Toeplitz
//                  fragments of the caller's coefficient set (low plane dense, high plane small); B operands: the loaded bytes.
//                  Whatever the product kernel does beyond this costs extra, so envelope_ms <= kernel_ms is the expectation and
//                  envelope_ms / kernel_ms says how much of the kernel's time the formulation itself accounts for
//                  (round 3 ran this as a separate tool on a builder-chosen box: tools/fir_envelope.hip, profiles/r3_fir255_envelope.txt).
#include "fir_kernels.hpp"

namespace acdsp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) diag_copy_kernel(const v4i *__restrict__ src, v4i *__restrict__ dst, int64_t n_vec) {
  const int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
  if (i < n_vec) { dst[i] = src[i]; }
}

hipError_t launch_diag_copy(const void *src, void *dst, int64_t bytes, hipStream_t s) {
  const int64_t n_vec = bytes / 16;
  if (n_vec <= 0) { return hipSuccess; }
  // a launch holds fewer than 2^32 threads per dimension (config 3's 68.7 GB block is 2^32 of them): rows of 2^20 workgroups
  const int64_t blocks = (n_vec + 255) / 256, gx = blocks < (1 << 20) ? blocks : (1 << 20), gy = (blocks + gx - 1) / gx;
  hipLaunchKernelGGL(diag_copy_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, (const v4i *)src, (v4i *)dst, n_vec);
  return hipGetLastError();
}

// MK = MFMAs per 1 KB wave-load (a 1024-sample step of int16 is two loads), HI of them on a high-plane fragment; U loads per burst.
// The stored words come from the accumulators as the PREVIOUS load left them (one load of pipelining, as the product's epilogue runs
// one step behind its MFMAs), so a store never waits for this load's MFMAs.
template <int MK, int HI, int U>
__global__ void __launch_bounds__(256, 2) diag_envelope_kernel(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y,
                                                             int64_t n_vec, int64_t span_vec) {
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  v4i Al[4], Ah[2];
#pragma unroll
  for (int i = 0; i < 4; i++) { Al[i] = frag[i * 64 + lane]; }
#pragma unroll
  for (int i = 0; i < 2; i++) { Ah[i] = frag[(4 + i) * 64 + lane]; }
  v16i acc[4] = {{0}, {0}, {0}, {0}};
  const int64_t s = wave * span_vec, e = s + span_vec < n_vec ? s + span_vec : n_vec;
  for (int64_t i = s + lane; i + 64 * (U - 1) < e; i += 64 * U) {
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = __builtin_nontemporal_load(x + i + 64 * u); }
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4i o = v[u];
      if (MK > 0) { o = (v4i){acc[0][u & 15], acc[1][(u + 1) & 15], acc[2][(u + 2) & 15], acc[3][(u + 3) & 15]}; }
      asm volatile("" : "+v"(o));                 // the four words are read HERE (else the scheduler keeps whole copies of the accumulators)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MK; m++) {
        const v4i a = m < HI ? Ah[m & 1] : Al[m & 3];
        acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, v[u], acc[m & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_nontemporal_store(o, y + i + 64 * u);
    }
  }
}

// The same envelope in the COPY kernel's geometry (round 6: the stream-only leg above runs 0.76 ms on the 4.3 GB of config 2 where the plain copy
// runs 0.69): one 16-byte element per thread, 256-thread workgroups in memory order, every wave lives for ONE 1 KB load, MK MFMAs and one store.
// Such a wave cannot keep Toeplitz fragments resident (six fragment loads per data load would be an L1-bound kernel), so the A operands are
// stand-ins made of the loaded bytes themselves: full-range bytes for the low-plane products, bytes masked to +-3 for the HI high-plane ones.
template <int MK, int HI>
__global__ void __launch_bounds__(256) diag_envelope_copygeom_kernel(const v4i *__restrict__ x, v4i *__restrict__ y, int64_t n_vec) {
  const int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_vec) { return; }
  const v4i v = x[i];
  v16i acc[4] = {{0}, {0}, {0}, {0}};
  const v4i al = (v4i){v.y, v.z, v.w, v.x}, ah = (v4i){v.z & 0x03030303, v.w & 0x03030303, v.x & 0x03030303, v.y & 0x03030303};
#pragma unroll
  for (int m = 0; m < MK; m++) { acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(m < HI ? ah : al, v, acc[m & 3], 0, 0, 0); }
  v4i o = v;
  if (MK > 0) { o = (v4i){acc[0][0], acc[1][5], acc[2][10], acc[3][15]}; }
  y[i] = o;
}

// ... and with the REAL Toeplitz fragments: four elements per thread (a 256-thread workgroup copies 16 KB, still in memory order, all four loads
// issued before the first product), the six fragments loaded once per wave (L1 hits: 6 KB per 4 KB of samples)
// ORD (ACDSP_DIAG_ENV_ORDER, an experiment on what the MFMA power follows): 0 = element by element, the A fragment changes with every product and B stays;
// 1 = fragment by fragment, A stays for four products and B changes; 2 = both operands change with every product
template <int MK, int HI, int ORD>
__global__ void __launch_bounds__(256) diag_envelope_copygeom4_kernel(const v4i *__restrict__ frag, const v4i *__restrict__ x, v4i *__restrict__ y, int64_t n_vec) {
  const int64_t i0 = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 1024 + threadIdx.x;
  if (i0 + 768 >= n_vec) { return; }
  v4i v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { v[u] = x[i0 + 256 * u]; }
  const int lane = threadIdx.x & 63;
  v4i Al[4], Ah[2];
#pragma unroll
  for (int i = 0; i < 4; i++) { Al[i] = frag[i * 64 + lane]; }
#pragma unroll
  for (int i = 0; i < 2; i++) { Ah[i] = frag[(4 + i) * 64 + lane]; }
  if constexpr (ORD == 0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v16i acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
      for (int m = 0; m < MK; m++) { acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(m < HI ? Ah[m & 1] : Al[m & 3], v[u], acc[m & 3], 0, 0, 0); }
      v4i o = v[u];
      if (MK > 0) { o = (v4i){acc[0][0], acc[1][5], acc[2][10], acc[3][15]}; }
      y[i0 + 256 * u] = o;
    }
  } else {
    v16i acc[4] = {{0}, {0}, {0}, {0}};     // one accumulator per element
#pragma unroll
    for (int m = 0; m < MK; m++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int mm = ORD == 1 ? m : (m + u) % (MK > 0 ? MK : 1);
        acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(mm < HI ? Ah[mm & 1] : Al[mm & 3], v[u], acc[u], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v4i o = v[u];
      if (MK > 0) { o = (v4i){acc[u][0], acc[u][5], acc[u][10], acc[u][15]}; }
      y[i0 + 256 * u] = o;
    }
  }
}

hipError_t launch_diag_envelope_copygeom(const uint32_t *d_frag, const void *x, void *y, int64_t bytes, int mfma, int mfma_hi, hipStream_t s) {
  if (d_frag) {
    const int64_t n_vec = bytes / 16, blocks = n_vec / 1024;
    if (blocks <= 0) { return hipSuccess; }
    const int64_t gx = blocks < (1 << 20) ? blocks : (1 << 20), gy = (blocks + gx - 1) / gx;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    ACDSP_TUNE_ENV(ord_env, "ACDSP_DIAG_ENV_ORDER");
    const int ord = ord_env ? atoi(ord_env) : 0;
#define ACDSP_ENVC4_CASE(NM, NH)                                                                                                 \
    if (mfma == NM && mfma_hi == NH) {                                                                                           \
      if (ord == 1) { hipLaunchKernelGGL((diag_envelope_copygeom4_kernel<NM / 2, NH / 2, 1>), grid, dim3(256), 0, s, (const v4i *)d_frag, (const v4i *)x, (v4i *)y, n_vec); } \
      else if (ord == 2) { hipLaunchKernelGGL((diag_envelope_copygeom4_kernel<NM / 2, NH / 2, 2>), grid, dim3(256), 0, s, (const v4i *)d_frag, (const v4i *)x, (v4i *)y, n_vec); } \
      else { hipLaunchKernelGGL((diag_envelope_copygeom4_kernel<NM / 2, NH / 2, 0>), grid, dim3(256), 0, s, (const v4i *)d_frag, (const v4i *)x, (v4i *)y, n_vec); } \
      return hipGetLastError();                                                                                                  \
    }
    ACDSP_ENVC4_CASE(0, 0) ACDSP_ENVC4_CASE(26, 8) ACDSP_ENVC4_CASE(36, 18) ACDSP_ENVC4_CASE(76, 10) ACDSP_ENVC4_CASE(132, 66)
#undef ACDSP_ENVC4_CASE
    return hipErrorInvalidValue;
  }
  const int64_t n_vec = bytes / 16;
  if (n_vec <= 0) { return hipSuccess; }
  const int64_t blocks = (n_vec + 255) / 256, gx = blocks < (1 << 20) ? blocks : (1 << 20), gy = (blocks + gx - 1) / gx;
  const dim3 grid((unsigned)gx, (unsigned)gy);
#define ACDSP_ENVC_CASE(NM, NH)                                                                                                  \
  if (mfma == NM && mfma_hi == NH) {                                                                                             \
    hipLaunchKernelGGL((diag_envelope_copygeom_kernel<NM / 2, NH / 2>), grid, dim3(256), 0, s, (const v4i *)x, (v4i *)y, n_vec); \
    return hipGetLastError();                                                                                                    \
  }
  ACDSP_ENVC_CASE(0, 0) ACDSP_ENVC_CASE(26, 8) ACDSP_ENVC_CASE(36, 18) ACDSP_ENVC_CASE(76, 10) ACDSP_ENVC_CASE(132, 66)
#undef ACDSP_ENVC_CASE
  return hipErrorInvalidValue;
}

// Placement probe (acdsp_diag_mix_ms, acdsp_dev_alloc_paired): a bare stream that READS one block and WRITES another at the byte ratio of the two
// blocks, in the product kernels' geometry -- one wave per span in memory order, up to eight 1 KB non-temporal loads in flight, 1 KB non-temporal
// stores.  The HBM-bound rows run 3 - 8 % apart on different (input, output) allocation pairs (profiles/r3_placement_modes.txt: same bytes, same
// TLB and L2 counters, more DRAM credit stalls on the slow pair); this kernel ranks candidate allocations the same way at a fraction of a row's time.
//   rd_per_wr >= 1: every wave reads rd_per_wr KB and writes 1 KB;  rd_per_wr < 0: reads 1 KB and writes -rd_per_wr KB
__global__ void __launch_bounds__(64) diag_mix_kernel(const v4i *__restrict__ src, v4i *__restrict__ dst, int64_t n_waves, int rd_per_wr) {
  const int64_t w = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (w >= n_waves) { return; }
  const int lane = threadIdx.x;
  if (rd_per_wr >= 1) {
    const v4i *s = src + w * rd_per_wr * 64 + lane;
    v4i acc = {0, 0, 0, 0};
    for (int r0 = 0; r0 < rd_per_wr; r0 += 8) {
      v4i v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { v[u] = __builtin_nontemporal_load(s + (r0 + u < rd_per_wr ? r0 + u : r0) * 64); }
#pragma unroll
      for (int u = 0; u < 8; u++) { acc ^= v[u]; }
    }
    __builtin_nontemporal_store(acc, dst + w * 64 + lane);
  } else {
    const int k = -rd_per_wr;
    v4i v = __builtin_nontemporal_load(src + w * 64 + lane);
    for (int r = 0; r < k; r++) { v.x += r; __builtin_nontemporal_store(v, dst + (w * k + r) * 64 + lane); }
  }
}

// reads [src, src + src_bytes), writes [dst, dst + dst_bytes); the ratio is rounded to a whole number of KB per wave
hipError_t launch_diag_mix(const void *src, int64_t src_bytes, void *dst, int64_t dst_bytes, hipStream_t s) {
  if (src_bytes < 1024 || dst_bytes < 1024) { return hipSuccess; }
  int ratio;
  int64_t n_waves;
  if (src_bytes >= dst_bytes) { int64_t r = src_bytes / dst_bytes; if (r > 256) { r = 256; } ratio = (int)r; n_waves = dst_bytes / 1024; if (n_waves * ratio * 1024 > src_bytes) { n_waves = src_bytes / (1024 * ratio); } }
  else { int64_t r = dst_bytes / src_bytes; if (r > 256) { r = 256; } ratio = -(int)r; n_waves = src_bytes / 1024; if (n_waves * r * 1024 > dst_bytes) { n_waves = dst_bytes / (1024 * r); } }
  if (n_waves <= 0) { return hipSuccess; }
  const int64_t gx = n_waves < (1 << 20) ? n_waves : (1 << 20), gy = (n_waves + gx - 1) / gx;
  hipLaunchKernelGGL(diag_mix_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(64), 0, s, (const v4i *)src, (v4i *)dst, n_waves, ratio);
  return hipGetLastError();
}

bool diag_envelope_compiled(int mfma, int mfma_hi) {
  return (mfma == 0 && mfma_hi == 0) || (mfma == 26 && mfma_hi == 8) || (mfma == 36 && mfma_hi == 18) || (mfma == 76 && mfma_hi == 10) ||
         (mfma == 132 && mfma_hi == 66);
}

// d_frag: six fragments [4 low-plane blocks][2 high-plane blocks] x 64 lanes x 16 bytes.  mfma / mfma_hi: per 1024 samples (even).
hipError_t launch_diag_envelope(const uint32_t *d_frag, const void *x, void *y, int64_t bytes, int mfma, int mfma_hi, hipStream_t s) {
  const int64_t n_vec = bytes / 16, span_vec = 32 * 64;            // 32 KB spans
  const int64_t waves = (n_vec + span_vec - 1) / span_vec, nb = (waves + 3) / 4;
  if (n_vec <= 0) { return hipSuccess; }
  const v4i *f = (const v4i *)d_frag;
#define ACDSP_ENV_CASE(NM, NH)                                                                                                   \
  if (mfma == NM && mfma_hi == NH) {                                                                                             \
    hipLaunchKernelGGL((diag_envelope_kernel<NM / 2, NH / 2, 8>), dim3((unsigned)nb), dim3(256), 0, s, f, (const v4i *)x, (v4i *)y, n_vec, span_vec); \
    return hipGetLastError();                                                                                                    \
  }
  ACDSP_ENV_CASE(0, 0)      // the stream alone
  ACDSP_ENV_CASE(26, 8)     // config 2: 255 taps, windowed-sinc set (18 low-plane + 8 high-plane products)
  ACDSP_ENV_CASE(36, 18)    // config 2, dense set
  ACDSP_ENV_CASE(76, 10)    // config 4: 1023 taps, five-block high band (66 + 10)
  ACDSP_ENV_CASE(132, 66)   // config 4, dense set
#undef ACDSP_ENV_CASE
  return hipErrorInvalidValue;
}

// Shader clock right behind a workload (bench.py: `clock_mhz_after` beside every row): one wave per XCD-sized slice of the chip spins on
// dependent VALU adds for ~40 us and compares the shader-clock counter with the constant 100 MHz real-time counter; launched on the
// workload's stream immediately behind its last timed step, before the power management has moved the clock (it reacts in milliseconds).
__global__ void diag_clock_kernel(float *mhz) {
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned v = threadIdx.x;
  uint64_t r1;
  do {
#pragma unroll
    for (int i = 0; i < 256; i++) { v = v * 3u + 1u; }
    r1 = __builtin_amdgcn_s_memrealtime();
  } while (r1 - r0 < 4000);
  const uint64_t c1 = __builtin_readcyclecounter();
  if (threadIdx.x == (v & 0u)) { mhz[blockIdx.x] = (float)((double)(c1 - c0) / (double)(r1 - r0) * 100.0); }
}
hipError_t launch_diag_clock(float *d_mhz, int n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(diag_clock_kernel, dim3((unsigned)n_blocks), dim3(64), 0, s, d_mhz);
  return hipGetLastError();
}

}  // namespace acdsp
