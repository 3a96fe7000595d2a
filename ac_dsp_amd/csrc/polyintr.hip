// polyintr.hip -- exact, order-preserving kernel for the polyphase interpolator (second half).
//
// Restates the three cores of ac_poly_intr_core (reference include/ac_dsp/ac_poly_intr.h:126-258) for one output per
// thread.  Per input sample n and phase j the reference forms the sub-filter sum acc_n[j] in ACC_TYPE from the shift
// register (= a window of the input stream here; samples before the call come from the handle's history):
//   FOLD_EVEN  i = N/2-1 .. 0      fold = ACC(taps[i] + tp),  tp = sign[j] ? taps[N-1-i] : IN(-taps[N-1-i]),
//                                  acc += coeffs[i + (j*N)/2] * fold                                    (:141-151)
//   FOLD_ODD   i = 0 .. (N-1)/2    centre tap alone, acc += coeffs[i + (N/2+1)*j] * fold                (:194-209)
//   FOLD_ANTI  i = N-1 .. 0        acc += taps[i] * coeffs[i + N*j], written out at once                (:246-256)
// The folded cores hold the sums in two banks and emit them ONE SAMPLE LATER (flip / init, :153-175): the group of
// sample n carries  t1 = acc_{n-1}[j], t2 = acc_{n-1}[corr[j]]  and writes  (t1 + (sign[j] ? ACC(-t2) : t2)) >> 1  when
// corr[j] != j, else t1; the very first sample emits nothing.  acc_{n-1} of a call's first sample was formed with the
// coefficients of its own time, so it is carried in the handle (`saved`), not recomputed.
#include <type_traits>

#include "fir_kernels.hpp"

namespace acdsp {

__device__ inline i128 pi_shl128(i128 v, int s) { return (i128)((u128)v << s); }
__device__ inline int64_t pi_mac(int64_t acc, i128 prod, int f_prod, const DFmt &A) {
  const int f = f_prod > A.F ? f_prod : A.F;
  return requant128(pi_shl128((i128)acc, f - A.F) + pi_shl128(prod, f - f_prod), f, A);
}

// sub-filter sum of phase j at local sample m (taps[i] = x[m - i])
__device__ __forceinline__ int64_t polyintr_acc(const PolyIntrParams &p, int ch, int64_t m, int j) {
  const int N = p.n_taps;
  auto xs = [&](int64_t t) -> int64_t {
    if (t >= 0) { return load_raw(p.x, (int64_t)ch * p.in_stride + t, p.in_eb, p.in.S); }
    if (t >= -(int64_t)p.hl) { return load_raw(p.hist, (int64_t)ch * p.hl + p.hl + t, p.in_eb, p.in.S); }
    return 0;
  };
  int64_t acc = 0;
  if (p.ftype == 2) {
    for (int i = N - 1; i >= 0; i--) { acc = pi_mac(acc, (i128)xs(m - i) * p.coeffs[i + N * j], p.in.F + p.cf.F, p.acc); }
    return acc;
  }
  const bool sg = p.sign[j] != 0;
  if (p.ftype == 0) {
    for (int i = (N / 2) - 1; i >= 0; i--) {
      const int64_t far = xs(m - (N - 1 - i));
      const int64_t tp = sg ? far : requant128(-(i128)far, p.in.F, p.in);
      const int64_t fold = requant128((i128)xs(m - i) + (i128)tp, p.in.F, p.acc);
      acc = pi_mac(acc, (i128)p.coeffs[i + j * N / 2] * (i128)fold, p.cf.F + p.acc.F, p.acc);
    }
  } else {
    const int mid = (N - 1) / 2;
    for (int i = 0; i <= mid; i++) {
      int64_t fold;
      if (i == mid) { fold = requant128((i128)xs(m - i), p.in.F, p.acc); }
      else {
        const int64_t far = xs(m - (N - 1 - i));
        const int64_t tp = sg ? far : requant128(-(i128)far, p.in.F, p.in);
        fold = requant128((i128)xs(m - i) + (i128)tp, p.in.F, p.acc);
      }
      acc = pi_mac(acc, (i128)p.coeffs[i + (N / 2 + 1) * j] * (i128)fold, p.cf.F + p.acc.F, p.acc);
    }
  }
  return acc;
}

// Lossless class (host-checked: signed IN, signed wrapping ACC with F_acc >= F_in + F_coeff and room for the fold): every
// `acc +=` is exact, so the sub-filter sum is an integer dot product mod 2^64 wrapped once to ACC_TYPE.  The only
// non-linear step left is the IN_TYPE negation of the most negative word, reproduced by a select.
__device__ __forceinline__ int64_t polyintr_acc_fast(const PolyIntrParams &p, int ch, int64_t m, int j) {
  const int N = p.n_taps;
  auto xs = [&](int64_t t) -> int64_t {
    if (t >= 0) { return load_raw(p.x, (int64_t)ch * p.in_stride + t, p.in_eb, 1); }
    if (t >= -(int64_t)p.hl) { return load_raw(p.hist, (int64_t)ch * p.hl + p.hl + t, p.in_eb, 1); }
    return 0;
  };
  uint64_t acc = 0;
  if (p.ftype == 2) {
    for (int i = N - 1; i >= 0; i--) { acc += (uint64_t)xs(m - i) * (uint64_t)p.coeffs[i + N * j]; }
  } else {
    const bool sg = p.sign[j] != 0;
    const int64_t neg_min = p.in.O == ACDSP_WRAP ? p.in.lo : ((p.in.O == ACDSP_SAT || p.in.O == ACDSP_SAT_SYM) ? p.in.hi : 0);
    const int mid = (N - 1) / 2;
    const int cnt = p.ftype == 0 ? N / 2 : mid + 1;
    const int cbase = p.ftype == 0 ? j * N / 2 : (N / 2 + 1) * j;
    for (int i = 0; i < cnt; i++) {
      int64_t fold = xs(m - i);
      if (p.ftype == 0 || i != mid) {
        const int64_t far = xs(m - (N - 1 - i));
        fold += sg ? far : (far == p.in.lo ? neg_min : -far);
      }
      acc += (uint64_t)p.coeffs[cbase + i] * (uint64_t)fold;
    }
  }
  return wrap64((int64_t)(acc << p.lossless_shift), p.acc.W, 1);
}

__global__ void polyintr_kernel(PolyIntrParams p) {
  const int ch = blockIdx.y;
  const int64_t o = p.o_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // output index of this call
  if (o >= p.o_end) { return; }
  const int IF = p.ifac;
  const int j = (int)(o % IF);
  const int64_t grp = o / IF + p.skip;                                 // local index of the sample that emits this group
  int64_t y;
  if (p.ftype == 2) {
    y = requant64(p.lossless ? polyintr_acc_fast(p, ch, grp, j) : polyintr_acc(p, ch, grp, j), p.acc.F, p.out);
  } else {
    const int cj = p.corr[j];
    int64_t t1, t2;
    if (grp == 0) { t1 = p.saved[(int64_t)ch * IF + j]; t2 = p.saved[(int64_t)ch * IF + cj]; }   // sums of the previous call's last sample
    else if (p.lossless) { t1 = polyintr_acc_fast(p, ch, grp - 1, j); t2 = (cj == j) ? t1 : polyintr_acc_fast(p, ch, grp - 1, cj); }
    else { t1 = polyintr_acc(p, ch, grp - 1, j); t2 = (cj == j) ? t1 : polyintr_acc(p, ch, grp - 1, cj); }
    if (cj != j) {
      const int64_t tn = p.sign[j] ? requant128(-(i128)t2, p.acc.F, p.acc) : t2;
      y = requant128(((i128)t1 + (i128)tn) >> 1, p.acc.F, p.out);
    } else {
      y = requant64(t1, p.acc.F, p.out);
    }
  }
  store_raw(p.y, (int64_t)ch * p.out_stride + o, p.out_eb, y);
}

// Tiled form of the lossless class: a workgroup owns kPiTile consecutive outputs of one channel, stages the input window
// and the coefficient table in LDS once, and each thread produces its outputs with int64 MACs from LDS (no per-thread
// 64-bit divisions, no global gathers).  The thread-per-output kernel above ran 203 ms on the bench row (0.02 TB/s).
constexpr int kPiTile = 2048;

// NARROW: IN_TYPE and COEFF_TYPE fit 31 / 32 bits: LDS words and the MAC (v_mad_i64_i32) are 32-bit.
template <bool NARROW, int FT>
__global__ void __launch_bounds__(256) polyintr_fast_kernel(PolyIntrParams p, uint32_t rcp, int n_win_max) {
  typedef typename std::conditional<NARROW, int32_t, int64_t>::type WT;
  extern __shared__ __attribute__((aligned(8))) unsigned char pi_lds[];
  WT *cf = (WT *)pi_lds;                           // [coeff_sz]
  WT *xw = cf + (p.coeff_sz + 1) / 2 * 2;          // [n_win_max]
  unsigned char *sgn = (unsigned char *)(xw + n_win_max), *cor = sgn + 256;   // [IF] each
  const int ch = blockIdx.y;
  const int IF = p.ifac, N = p.n_taps;
  const int64_t o0 = p.o_begin + (int64_t)blockIdx.x * kPiTile;
  const int64_t g0 = o0 / IF;                      // wave-uniform, once
  const int rem0 = (int)(o0 - g0 * IF);
  const int lag = FT == 2 ? 0 : 1;            // the folded cores emit the sums of the previous sample
  // window: local samples m_lo - (N-1) .. m_lo + groups, m_lo = g0 + skip - lag
  const int64_t m_lo = g0 + p.skip - lag;
  const int n_groups = (rem0 + kPiTile - 1) / IF + 1;
  const int n_win = N - 1 + n_groups;
  for (int i = threadIdx.x; i < p.coeff_sz; i += 256) { cf[i] = (WT)p.coeffs[i]; }
  if ((int)threadIdx.x < p.ifac) { sgn[threadIdx.x] = p.sign[threadIdx.x]; cor[threadIdx.x] = p.corr[threadIdx.x]; }
  for (int i = threadIdx.x; i < n_win; i += 256) {
    const int64_t t = m_lo - (N - 1) + i;
    int64_t v = 0;
    if (t >= 0) { if (t < p.n) { v = load_raw(p.x, (int64_t)ch * p.in_stride + t, p.in_eb, 1); } }
    else if (t >= -(int64_t)p.hl) { v = load_raw(p.hist, (int64_t)ch * p.hl + p.hl + t, p.in_eb, 1); }
    xw[i] = (WT)v;
  }
  __syncthreads();
  const int64_t neg_min = p.in.O == ACDSP_WRAP ? p.in.lo : ((p.in.O == ACDSP_SAT || p.in.O == ACDSP_SAT_SYM) ? p.in.hi : 0);
  const int mid = (N - 1) / 2;
  const int cnt = FT == 0 ? N / 2 : mid + 1;
  auto sub = [&](const WT *w, int j) -> int64_t {   // w[-i] = taps[i] of the sample
    uint64_t acc = 0;
    if (FT == 2) {
      for (int i = N - 1; i >= 0; i--) {
        if (NARROW) { acc = (uint64_t)((int64_t)acc + (int64_t)(int32_t)w[-i] * (int64_t)(int32_t)cf[i + N * j]); }
        else { acc += (uint64_t)w[-i] * (uint64_t)cf[i + N * j]; }
      }
    } else {
      const bool sg = sgn[j] != 0;
      const int cbase = FT == 0 ? j * N / 2 : (N / 2 + 1) * j;
      for (int i = 0; i < cnt; i++) {
        int64_t fold = w[-i];
        if (FT == 0 || i != mid) {
          const int64_t far = w[-(N - 1 - i)];
          fold += sg ? far : (far == p.in.lo ? neg_min : -far);
        }
        if (NARROW) { acc = (uint64_t)((int64_t)acc + (int64_t)(int32_t)cf[cbase + i] * (int64_t)(int32_t)fold); }   // |fold| < 2^31
        else { acc += (uint64_t)cf[cbase + i] * (uint64_t)fold; }
      }
    }
    return wrap64((int64_t)(acc << p.lossless_shift), p.acc.W, 1);
  };
  for (int it = 0; it < kPiTile / 256; it++) {
    const int64_t o = o0 + threadIdx.x + 256 * it;
    if (o >= p.o_end) { return; }
    const unsigned t = (unsigned)rem0 + threadIdx.x + 256u * it;     // < IF + kPiTile < 2^16
    const unsigned dg = IF == 1 ? t : __umulhi(t, rcp);              // ceil(2^32 / 1) does not fit the 32-bit reciprocal
    const int j = (int)(t - dg * (unsigned)IF);
    const int64_t m = m_lo + dg;                                     // local sample whose sums this output carries
    const WT *w = xw + (N - 1) + dg;
    int64_t y;
    if (FT == 2) {
      y = requant64(sub(w, j), p.acc.F, p.out);
    } else {
      const int cj = cor[j];
      int64_t t1, t2;
      if (m < 0) { t1 = p.saved[(int64_t)ch * IF + j]; t2 = p.saved[(int64_t)ch * IF + cj]; }
      else { t1 = sub(w, j); t2 = (cj == j) ? t1 : sub(w, cj); }
      if (cj != j) {
        const int64_t tn = sgn[j] ? wrap64(-t2, p.acc.W, 1) : t2;
        y = requant64((t1 + tn) >> 1, p.acc.F, p.out);
      } else {
        y = requant64(t1, p.acc.F, p.out);
      }
    }
    store_raw(p.y, (int64_t)ch * p.out_stride + o, p.out_eb, y);
  }
}

// sums of the call's last sample -> the handle (they are emitted by the next call's first sample)
__global__ void polyintr_save_kernel(PolyIntrParams p, int64_t *saved_next) {
  const int ch = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.ifac) { return; }
  saved_next[(int64_t)ch * p.ifac + j] = p.lossless ? polyintr_acc_fast(p, ch, p.n - 1, j) : polyintr_acc(p, ch, p.n - 1, j);
}

hipError_t launch_polyintr(const PolyIntrParams &p, int64_t *saved_next, hipStream_t s) {
  const int64_t span = p.o_end - p.o_begin;
  if (span > 0) {
    const int n_win_max = p.n_taps + kPiTile / p.ifac + 4;
    const size_t lds = ((size_t)p.coeff_sz + 2 + (size_t)n_win_max) * sizeof(int64_t) + 512;
    if (p.lossless && lds <= 60 * 1024) {
      dim3 grid((unsigned)((span + kPiTile - 1) / kPiTile), (unsigned)p.n_ch);
      const uint32_t rcp = (uint32_t)((0x100000000ull + p.ifac - 1) / p.ifac);
      const bool narrow = p.in.W <= 30 && p.cf.W <= 32;
#define ACDSP_PI_LAUNCH(NV, FV) hipLaunchKernelGGL((polyintr_fast_kernel<NV, FV>), grid, dim3(256), lds, s, p, rcp, n_win_max)
      if (narrow) {
        if (p.ftype == 0) { ACDSP_PI_LAUNCH(true, 0); } else if (p.ftype == 1) { ACDSP_PI_LAUNCH(true, 1); } else { ACDSP_PI_LAUNCH(true, 2); }
      } else {
        if (p.ftype == 0) { ACDSP_PI_LAUNCH(false, 0); } else if (p.ftype == 1) { ACDSP_PI_LAUNCH(false, 1); } else { ACDSP_PI_LAUNCH(false, 2); }
      }
#undef ACDSP_PI_LAUNCH
    } else {
      dim3 grid((unsigned)((span + 255) / 256), (unsigned)p.n_ch);
      hipLaunchKernelGGL(polyintr_kernel, grid, dim3(256), 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return e; }
  }
  if (p.ftype != 2 && p.n > 0 && saved_next) {
    dim3 grid((unsigned)((p.ifac + 63) / 64), (unsigned)p.n_ch);
    hipLaunchKernelGGL(polyintr_save_kernel, grid, dim3(64), 0, s, p, saved_next);
    return hipGetLastError();
  }
  return hipSuccess;
}

}  // namespace acdsp
