// wide.hip -- exact-order kernels for types wider than 64 bits (accumulators / outputs / CIC INT_TYPE up to 128 bits).
//
// The reference's templates take ac_fixed of any width: ac_cic_dec_full derives INT_TYPE = W + N log2(R M) bits (reference
// include/ac_dsp/ac_cic_dec_full.h:116-137, e.g. R 32, N 8 on <32,16> = 72 bits; the interpolator likewise,
// ac_cic_intr_full.h:107-127), and the FIR cores accumulate into whatever ACC_TYPE the user names (its own testbenches
// already form 96-bit products, tests/rtest_ac_fir_const_coeffs.cpp:50-74).  These kernels are the engine's path for any
// handle with a format of more than 64 bits: IN / COEFF <= 64 bits, ACC / OUT / INT_TYPE <= 128 bits, raw words in 16-byte
// containers (little endian: low quadword first).  They replay the reference's loops in its own order with 256-bit exact
// intermediates (wide_int.hpp); they are a correctness path (one thread per output, VALU only), not a roofline path.
//
//   FIR:  the loops of fir_generic.hip (reference ac_fir_const_coeffs.h:190-296 and twins) on i128 accumulators
//   CIC:  both directions through the FIR identity  out = sum_k h[k] x[.-k] mod 2^W_int  with h = z^-(N-1) boxcar(R M')^N
//         (every tap < 2^31, so a 128-bit modular sum is exact mod 2^W_int for W_int <= 128)
#include "cic_kernels.hpp"
#include "fir_kernels.hpp"
#include "wide_int.hpp"
#include "wide_kernels.hpp"

namespace acdsp {

// low W bits of q, extended per S
__host__ __device__ inline i128 wrap_w128(const i256 &q, int W, int S) {
  const i128 v = q.low128();
  if (W >= 128) { return v; }
  const int sh = 128 - W;
  return S ? (i128)((u128)v << sh) >> sh : (i128)(((u128)v << sh) >> sh);
}

__host__ __device__ inline i128 overflow_w(const i256 &q, const WFmt &d) {
  const i256 lo(d.lo), hi(d.hi);
  const bool under = q < lo, over = q > hi;
  switch (d.O) {
    case ACDSP_SAT: return under ? d.lo : (over ? d.hi : q.low128());
    case ACDSP_SAT_ZERO: return (under || over) ? (i128)0 : q.low128();
    case ACDSP_SAT_SYM:
      if (d.S) {
        if (under || over) { return q.neg() ? d.lo + 1 : d.hi; }
        return (q.low128() == d.lo && d.W > 1) ? d.lo + 1 : q.low128();
      }
      return under ? d.lo : (over ? d.hi : q.low128());
    default: return wrap_w128(q, d.W, d.S);
  }
}

// exact x * 2^-f_src -> raw word of d (quantise with d.Q, then overflow with d.O; the rounding carry takes part in the overflow)
__host__ __device__ inline i128 requant_w(const i256 &x, int f_src, const WFmt &d) {
  const int sh = f_src - d.F;
  if (sh <= 0) { return overflow_w(shl(x, -sh), d); }
  const int qb = x.bit(sh - 1), r = x.any_below(sh - 1) ? 1 : 0;
  i256 q = sar(x, sh);
  if (q_increment(d.Q, qb, r, x.neg() ? 1 : 0, q.bit(0))) { q = q + i256(1); }
  return overflow_w(q, d);
}

// acc = ACC_TYPE(acc + prod), prod exact with f_prod fractional bits
__device__ inline i128 mac_w(i128 acc, const i256 &prod, int f_prod, const WFmt &A) {
  const int f = f_prod > A.F ? f_prod : A.F;
  return requant_w(shl(i256(acc), f - A.F) + shl(prod, f - f_prod), f, A);
}

__device__ inline void store_w(void *p, int64_t idx, int eb, i128 v) {
  if (eb == 16) {
    uint64_t *q = (uint64_t *)p + 2 * idx;
    q[0] = (uint64_t)(u128)v; q[1] = (uint64_t)((u128)v >> 64);
  } else {
    store_raw(p, idx, eb, (int64_t)v);
  }
}
__device__ inline i128 load_w(const void *p, int64_t idx) {
  const uint64_t *q = (const uint64_t *)p + 2 * idx;
  return (i128)(((u128)q[1] << 64) | q[0]);
}

constexpr int kWTile = 128;

// One thread = one output; the block stages its input window and the taps in LDS (as in fir_direct_kernel).
__global__ void __launch_bounds__(kWTile) fir_wide_kernel(FirWideParams pw) {
  extern __shared__ int64_t smem[];
  const FirParams &p = pw.p;
  const int N = p.n_taps;
  int64_t *win = smem;                  // [kWTile + N - 1]: win[j] = x[t0 - (N-1) + j]
  int64_t *cf = smem + kWTile + N - 1;  // [N]
  const int ch = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kWTile;
  const int tid = threadIdx.x;
  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  for (int i = tid; i < N; i += kWTile) { cf[i] = cg[i]; }
  for (int j = tid; j < kWTile + N - 1; j += kWTile) {
    const int64_t g = t0 - (N - 1) + j;
    int64_t v = 0;
    if (g >= 0) {
      if (g < p.n) { v = load_raw(p.x, (int64_t)ch * p.in_stride + g, p.in_eb, p.in.S); }
    } else if (!p.use_rt && g >= -(int64_t)p.hl) {
      v = load_raw(p.hist, (int64_t)ch * p.hl + p.hl + g, p.in_eb, p.in.S);
    }
    win[j] = v;
  }
  __syncthreads();
  const int64_t t = t0 + tid;
  if (t >= p.n) { return; }
  const int64_t *w = win + tid + (N - 1);  // w[-k] = x[t-k]
  const int fp = p.in.F + p.cf.F;
  const WFmt &A = pw.acc;
  i128 acc = 0;
  switch (p.ftype) {
    case ACDSP_SHIFT_REG:
    case ACDSP_ROTATE_SHIFT:   // reference ac_fir_const_coeffs.h:190-220: i = N-1 .. 0
      for (int i = N - 1; i >= 0; i--) { acc = mac_w(acc, i256(w[-i]) * i256(cf[i]), fp, A); }
      break;
    case ACDSP_C_BUFF:         // :226-237 ascending
    case kRsShiftReg:          // ac_fir_reg_share.h:136-150
      for (int i = 0; i < N; i++) { acc = mac_w(acc, i256(w[-i]) * i256(cf[i]), fp, A); }
      break;
    case kRsFoldEven:          // ac_fir_reg_share.h:157-171
    case kRsFoldEvenAnti:      // :178-192
      for (int i = 0; i < N / 2; i++) {
        const i256 pre = p.ftype == kRsFoldEven ? i256(w[-i]) + i256(w[-(N - 1 - i)]) : i256(w[-i]) - i256(w[-(N - 1 - i)]);
        acc = mac_w(acc, i256(cf[i]) * pre, fp, A);
      }
      break;
    case ACDSP_FOLD_EVEN:      // ac_fir_const_coeffs.h:244-253: i = N/2-1 .. 0, exact pre-add
      for (int i = N / 2 - 1; i >= 0; i--) { acc = mac_w(acc, i256(cf[i]) * (i256(w[-i]) + i256(w[-(N - 1 - i)])), fp, A); }
      break;
    case ACDSP_FOLD_ODD:       // :260-275: `fold` is an ACC_TYPE variable
    case kRsFoldOdd:           // ac_fir_reg_share.h:199-219
    case kRsFoldOddAnti: {     // :226-246
      const int mid = (N - 1) / 2;
      for (int i = 0; i <= mid; i++) {
        const i256 pre = (i == mid) ? i256(w[-i])
                                    : (p.ftype == kRsFoldOddAnti ? i256(w[-i]) - i256(w[-(N - 1 - i)]) : i256(w[-i]) + i256(w[-(N - 1 - i)]));
        const i128 fold = requant_w(pre, p.in.F, A);
        acc = mac_w(acc, i256(cf[i]) * i256(fold), p.cf.F + A.F, A);
      }
      break;
    }
    case ACDSP_TRANSPOSED: {   // :281-296, unrolled in time as in fir_direct_kernel
      int jstart = N - 1;
      if (p.use_rt && t < N - 1) {
        jstart = (int)t;
        acc = load_w(pw.rt, (int64_t)ch * N + (N - 2 - (int)t));
      }
      for (int j = jstart; j >= 0; j--) { acc = mac_w(acc, i256(w[-j]) * i256(cf[j]), fp, A); }
      break;
    }
    default: break;
  }
  store_w(p.y, (int64_t)ch * p.out_stride + t, p.out_eb, requant_w(i256(acc), A.F, pw.out));   // data_out = acc
}

// reg_trans[i] after the last sample of the call (see fir_rt_update_kernel), ACC_TYPE words of 128 bits
__global__ void fir_wide_rt_update_kernel(FirWideParams pw, void *rt_next) {
  const FirParams &p = pw.p;
  const int ch = blockIdx.y;
  const int N = p.n_taps;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) { return; }
  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  const int fp = p.in.F + p.cf.F;
  i128 r = (i >= p.n) ? load_w(pw.rt, (int64_t)ch * N + (i - p.n)) : (i128)0;
  const int64_t m0 = (i < p.n - 1) ? i : p.n - 1;
  for (int64_t m = m0; m >= 0; m--) {
    const int64_t x = load_raw(p.x, (int64_t)ch * p.in_stride + (p.n - 1 - m), p.in_eb, p.in.S);
    r = mac_w(r, i256(x) * i256(cg[N - 1 - i + m]), fp, pw.acc);
  }
  store_w(rt_next, (int64_t)ch * N + i, 16, r);
}

hipError_t launch_fir_wide(const FirWideParams &pw, hipStream_t s) {
  if (pw.p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((pw.p.n + kWTile - 1) / kWTile), (unsigned)pw.p.n_ch);
  const size_t lds = sizeof(int64_t) * (size_t)(kWTile + 2 * pw.p.n_taps - 1);
  hipLaunchKernelGGL(fir_wide_kernel, grid, dim3(kWTile), lds, s, pw);
  return hipGetLastError();
}

hipError_t launch_fir_wide_rt_update(const FirWideParams &pw, void *rt_next, hipStream_t s) {
  if (pw.p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((pw.p.n_taps + 63) / 64), (unsigned)pw.p.n_ch);
  hipLaunchKernelGGL(fir_wide_rt_update_kernel, grid, dim3(64), 0, s, pw, rt_next);
  return hipGetLastError();
}

// ---- CIC, both directions, direct form of the FIR identity; one thread per output ----
// decimator:     output j of the call sits at local input time t = first + j R:  sum_k h[k] x[t - k]
// interpolator:  iteration q (global): n = q / R, r = q % R:  sum_k h[r + R k] x[n - k]   (x[n] enters at q = R n)
__global__ void __launch_bounds__(256) cic_wide_kernel(CicWideParams pw, const int64_t *__restrict__ taps, int n_taps, int64_t n_out) {
  const CicParams &p = pw.p;
  const int ch = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) { return; }
  u128 acc = 0;
  auto x_at = [&](int64_t n_local) -> int64_t {
    if (n_local >= 0) { return n_local < p.n_in ? load_raw(p.x, (int64_t)ch * p.in_stride + n_local, p.in_eb, p.in.S) : 0; }
    return n_local >= -(int64_t)p.hl ? load_raw(p.hist, (int64_t)ch * p.hl + p.hl + n_local, p.in_eb, p.in.S) : 0;
  };
  if (!p.interp) {
    const int64_t t = p.first + j * p.R;
    for (int k = 0; k < n_taps; k++) { acc += (u128)(i128)taps[k] * (u128)(i128)x_at(t - k); }
  } else {
    const int64_t lo = p.q_begin > p.q_skip ? p.q_begin : p.q_skip;
    const int64_t q = lo + j, n = q / p.R;
    const int r = (int)(q - n * p.R);
    for (int d = r, k = 0; d < n_taps; d += p.R, k++) { acc += (u128)(i128)taps[d] * (u128)(i128)x_at(n - k - p.t_prev); }
  }
  // INT_TYPE (signed, AC_TRN, AC_WRAP: ac_cic_dec_full.h:132-136) then `data_out_final = data_out_t` (:219)
  const int sh = 128 - p.w_int;
  const i128 v = (i128)(acc << sh) >> sh;
  store_w(p.y, (int64_t)ch * p.out_stride + j, p.out_eb, requant_w(i256(v), p.in.F, pw.out));
}

hipError_t launch_cic_wide(const CicWideParams &pw, const int64_t *d_taps, int n_taps, int64_t n_out, hipStream_t s) {
  if (n_out <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)pw.p.n_ch);
  hipLaunchKernelGGL(cic_wide_kernel, grid, dim3(256), 0, s, pw, d_taps, n_taps, n_out);
  return hipGetLastError();
}

}  // namespace acdsp
