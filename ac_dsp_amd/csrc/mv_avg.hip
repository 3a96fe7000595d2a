// mv_avg.hip -- ac_mv_avg on many objects (second half).
//
// Replaces the per-sample loop of ac_mv_avg_core::mvAvgCore (reference include/ac_dsp/ac_mv_avg.h:111-123) driven by
// ac_mv_avg::run (:146-190): for every valid window position of a frame
//     acc = 0;  for j = -TAPS/2 .. TAPS/2:  acc = ACC_TYPE(acc + ACC_TYPE(w[j]) * coeffs[j + TAPS/2]);  out = OUT_TYPE(acc)
// with w the ac_window_1d_flag over the frame (AC_WIN: interior positions only; AC_CLIP / AC_MIRROR: every position,
// samples outside the frame replaced by the edge sample / the sample mirrored about it -- include/ac_types/ac_window.h).
//
// Mapping: one thread per output, a 256-thread block per tile of one (object, frame); the block stages the
// 256 + TAPS - 1 window samples -- boundary rule applied, already cast to ACC_TYPE (the cast depends on the sample only) --
// and the coefficients in LDS with coalesced loads, then every thread walks its taps in the reference's order.
// Two arithmetic classes: per-tap requantisation in 128 bits (any Q / O), and, for AC_TRN / AC_RND + AC_WRAP accumulators
// whose products fit 63 bits, the order-free form acc = sum_j ((xq_j * c_j + rnd) >> F_c) mod 2^W_acc in int64.
// Streaming op: 2 + 2 bytes per sample at 16-bit containers; bound by HBM for short windows.
#include "fir_kernels.hpp"

namespace acdsp {

namespace {

constexpr int kTile = 256;

__device__ inline i128 shl128_(i128 v, int s) { return (i128)((u128)v << s); }

// line position of window element `pos` after the boundary rule (n samples per frame)
__device__ inline int64_t fold_pos(int64_t pos, int64_t n, int mode) {
  if (mode == 2) { return pos < 0 ? 0 : (pos > n - 1 ? n - 1 : pos); }
  if (n == 1) { return 0; }
  while (pos < 0 || pos > n - 1) {
    if (pos < 0) { pos = -pos; }
    if (pos > n - 1) { pos = 2 * (n - 1) - pos; }
  }
  return pos;
}

template <bool FAST>
__global__ void __launch_bounds__(kTile) mv_avg_kernel(MvAvgParams p) {
  extern __shared__ int64_t smem[];
  int64_t *win = smem;                       // [kTile + taps - 1]: ACC raw words of line positions m0 - h + j
  int64_t *cf = smem + kTile + p.taps - 1;   // [taps]
  const int tid = threadIdx.x, h = p.taps / 2;
  for (int i = tid; i < p.taps; i += kTile) { cf[i] = p.coeffs[i]; }
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  const int64_t first = p.win_mode == 0 ? h : 0;   // line position of a frame's first output
  for (int64_t pr = blockIdx.y; pr < pairs; pr += gridDim.y) {
    const int64_t obj = pr / p.n_frames, fr = pr % p.n_frames;
    const int64_t xbase = obj * p.in_stride + fr * p.n_sample, ybase = obj * p.out_stride + fr * p.out_per_frame;
    const int64_t m0 = first + (int64_t)blockIdx.x * kTile;   // line position of this tile's first output
    __syncthreads();   // previous pair's readers are done with win[]
    for (int j = tid; j < kTile + p.taps - 1; j += kTile) {
      int64_t pos = m0 - h + j;
      int64_t v = 0;
      if (p.win_mode != 0) { pos = fold_pos(pos, p.n_sample, p.win_mode); }
      if (pos >= 0 && pos < p.n_sample) { v = requant64(load_raw(p.x, xbase + pos, p.in_eb, p.in.S), p.in.F, p.acc); }   // (ACC_TYPE) w[j]
      win[j] = v;
    }
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * kTile + tid;   // output index within the frame
    if (k < p.out_per_frame) {
      const int64_t *w = win + tid;   // w[j + h] = element j of the window centred on this output
      int64_t acc = 0;
      if (FAST) {
        uint64_t s = 0;
        const int sh = p.cf.F;   // >= 0 in this class
        const int64_t rnd = (p.acc.Q == ACDSP_RND && sh > 0) ? (int64_t(1) << (sh - 1)) : 0;
        for (int j = 0; j < p.taps; j++) { s += (uint64_t)((w[j] * cf[j] + rnd) >> sh); }
        acc = wrap64((int64_t)s, p.acc.W, p.acc.S);
      } else {
        const int fp = p.acc.F + p.cf.F, f = fp > p.acc.F ? fp : p.acc.F;
        for (int j = 0; j < p.taps; j++) {
          const i128 sum = shl128_((i128)acc, f - p.acc.F) + shl128_((i128)w[j] * (i128)cf[j], f - fp);
          acc = requant128(sum, f, p.acc);
        }
      }
      store_raw(p.y, ybase + k, p.out_eb, requant64(acc, p.acc.F, p.out));
    }
  }
}

}  // namespace

hipError_t launch_mv_avg(const MvAvgParams &p, hipStream_t s) {
  if (p.out_per_frame <= 0 || p.n_frames <= 0) { return hipSuccess; }
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  dim3 grid((unsigned)((p.out_per_frame + kTile - 1) / kTile), (unsigned)(pairs < 65535 ? pairs : 65535));
  const size_t lds = (size_t)(kTile + 2 * p.taps - 1) * sizeof(int64_t);
  if (p.fast) { hipLaunchKernelGGL(mv_avg_kernel<true>, grid, dim3(kTile), lds, s, p); }
  else { hipLaunchKernelGGL(mv_avg_kernel<false>, grid, dim3(kTile), lds, s, p); }
  return hipGetLastError();
}

}  // namespace acdsp
