// fir_generic.hip -- exact, order-preserving FIR kernels (VALU) and the state-carry kernels.
//
// One thread computes one output sample y[ch][t] by replaying the reference's
// MAC loop for that sample in the reference's own tap order, so that every
// accumulator quantisation/overflow (`acc += ...`) happens on the same partial
// sum as in the reference cores:
//   SHIFT_REG / ROTATE_SHIFT  i = N-1..0            ac_fir_const_coeffs.h:190-220
//   C_BUFF                    i = 0..N-1            ac_fir_const_coeffs.h:226-237
//   FOLD_EVEN                 i = N/2-1..0, exact pre-add            :244-253
//   FOLD_ODD                  i = 0..(N-1)/2, pre-add held in ACC_TYPE :260-275
//   TRANSPOSED                partial-sum chain through reg_trans[]   :281-296
//   ac_fir_reg_share cores    ascending, blocks of BLK_SZ; anti-symmetric folds   ac_fir_reg_share.h:136-260
// (the load_/prog_ cores are the same loops: ac_fir_load_coeffs.h:180-278,
// ac_fir_prog_coeffs.h:147-247).  The shift register of the reference becomes
// a window of the input stream: reg[k] after firShiftReg(x[t]) is x[t-k], with
// samples before t = 0 taken from the handle's history buffer (zeros after
// construction, ac_fir_const_coeffs.h:145).
//
// Data movement: a 256-thread block stages 256+N-1 inputs of one channel and
// the N coefficients in LDS (coalesced loads), then threads read the window
// with unit stride (conflict-free) and the coefficient as an LDS broadcast.
#include "fir_kernels.hpp"

namespace acdsp {

constexpr int kTile = 256;

__device__ inline i128 shl128(i128 v, int s) { return (i128)((u128)v << s); }

// acc = ACC_TYPE(acc + prod), prod exact with f_prod fractional bits
__device__ inline int64_t mac_q(int64_t acc, i128 prod, int f_prod, const DFmt &A) {
  int f = f_prod > A.F ? f_prod : A.F;
  i128 s = shl128((i128)acc, f - A.F) + shl128(prod, f - f_prod);
  return requant128(s, f, A);
}

template <bool LOSSLESS>
__global__ void __launch_bounds__(kTile) fir_direct_kernel(FirParams p) {
  extern __shared__ int64_t smem[];
  const int N = p.n_taps;
  int64_t *win = smem;                 // [kTile + N - 1]: win[j] = x[t0 - (N-1) + j]
  int64_t *cf = smem + kTile + N - 1;  // [N]
  const int ch = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kTile;
  const int tid = threadIdx.x;

  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  for (int i = tid; i < N; i += kTile) { cf[i] = cg[i]; }
  for (int j = tid; j < kTile + N - 1; j += kTile) {
    int64_t g = t0 - (N - 1) + j;
    int64_t v = 0;
    if (g >= 0) {
      if (g < p.n) { v = load_raw(p.x, (int64_t)ch * p.in_stride + g, p.in_eb, p.in.S); }
    } else if (!p.use_rt && g >= -(int64_t)p.hl) {
      v = load_raw(p.hist, (int64_t)ch * p.hl + p.hl + g, p.in_eb, p.in.S);
    }
    win[j] = v;
  }
  __syncthreads();

  // small calls (hist_next set): block 0 of a channel also writes the next history -- hist_next[j] = sample at local time
  // n - hl + j, from this call's input or the old history -- so a one-sample run() is ONE launch (cf. fir_hist_update_kernel)
  if (p.hist_next && blockIdx.x == 0) {
    for (int j = tid; j < p.hl; j += kTile) {
      const int64_t g = p.n - p.hl + j;
      const int64_t v = (g >= 0) ? load_raw(p.x, (int64_t)ch * p.in_stride + g, p.in_eb, p.in.S)
                                 : load_raw(p.hist, (int64_t)ch * p.hl + p.hl + g, p.in_eb, p.in.S);
      store_raw(p.hist_next, (int64_t)ch * p.hl + j, p.in_eb, v);
    }
  }
  const int64_t t = t0 + tid;
  if (t >= p.n) { return; }
  const int64_t *w = win + tid + (N - 1);  // w[-k] = x[t-k]
  const int fp = p.in.F + p.cf.F;
  int64_t acc = 0;

  if (LOSSLESS) {
    // All partial sums are exact and AC_WRAP is a ring homomorphism: sum mod 2^64, wrap once.
    uint64_t s = 0;
    switch (p.ftype) {
      case ACDSP_FOLD_EVEN:
        for (int i = 0; i < N / 2; i++) { s += (uint64_t)cf[i] * (uint64_t)(w[-i] + w[-(N - 1 - i)]); }
        break;
      case ACDSP_FOLD_ODD:
      case kRsFoldOdd: {
        const int mid = (N - 1) / 2;
        for (int i = 0; i < mid; i++) { s += (uint64_t)cf[i] * (uint64_t)(w[-i] + w[-(N - 1 - i)]); }
        s += (uint64_t)cf[mid] * (uint64_t)w[-mid];
        break;
      }
      case kRsFoldEven:
        for (int i = 0; i < N / 2; i++) { s += (uint64_t)cf[i] * (uint64_t)(w[-i] + w[-(N - 1 - i)]); }
        break;
      case kRsFoldEvenAnti:
        for (int i = 0; i < N / 2; i++) { s += (uint64_t)cf[i] * (uint64_t)(w[-i] - w[-(N - 1 - i)]); }
        break;
      case kRsFoldOddAnti: {
        const int mid = (N - 1) / 2;
        for (int i = 0; i < mid; i++) { s += (uint64_t)cf[i] * (uint64_t)(w[-i] - w[-(N - 1 - i)]); }
        s += (uint64_t)cf[mid] * (uint64_t)w[-mid];
        break;
      }
      default:
        for (int i = 0; i < N; i++) { s += (uint64_t)cf[i] * (uint64_t)w[-i]; }
        break;
    }
    acc = wrap64((int64_t)(s << p.lossless_shift), p.acc.W, p.acc.S);
  } else {
    switch (p.ftype) {
      case ACDSP_SHIFT_REG:
      case ACDSP_ROTATE_SHIFT:
        for (int i = N - 1; i >= 0; i--) { acc = mac_q(acc, (i128)w[-i] * cf[i], fp, p.acc); }
        break;
      case ACDSP_C_BUFF:
      case kRsShiftReg:        // ac_fir_reg_share.h:136-150: ascending
        for (int i = 0; i < N; i++) { acc = mac_q(acc, (i128)w[-i] * cf[i], fp, p.acc); }
        break;
      case kRsFoldEven:        // :157-171
      case kRsFoldEvenAnti:    // :178-192
        for (int i = 0; i < N / 2; i++) {
          i128 pre = p.ftype == kRsFoldEven ? (i128)w[-i] + (i128)w[-(N - 1 - i)] : (i128)w[-i] - (i128)w[-(N - 1 - i)];
          acc = mac_q(acc, (i128)cf[i] * pre, fp, p.acc);
        }
        break;
      case kRsFoldOdd:         // :199-219 (same loop as ACDSP_FOLD_ODD)
      case kRsFoldOddAnti: {   // :226-246
        const int mid = (N - 1) / 2;
        for (int i = 0; i <= mid; i++) {
          i128 pre = (i == mid) ? (i128)w[-i]
                                : (p.ftype == kRsFoldOdd ? (i128)w[-i] + (i128)w[-(N - 1 - i)] : (i128)w[-i] - (i128)w[-(N - 1 - i)]);
          int64_t fold = requant128(pre, p.in.F, p.acc);  // ACC_TYPE fold
          acc = mac_q(acc, (i128)cf[i] * (i128)fold, p.cf.F + p.acc.F, p.acc);
        }
        break;
      }
      case ACDSP_FOLD_EVEN:
        for (int i = N / 2 - 1; i >= 0; i--) {
          i128 pre = (i128)w[-i] + (i128)w[-(N - 1 - i)];
          acc = mac_q(acc, (i128)cf[i] * pre, fp, p.acc);
        }
        break;
      case ACDSP_FOLD_ODD: {
        const int mid = (N - 1) / 2;
        for (int i = 0; i <= mid; i++) {
          i128 pre = (i == mid) ? (i128)w[-i] : (i128)w[-i] + (i128)w[-(N - 1 - i)];
          int64_t fold = requant128(pre, p.in.F, p.acc);  // ACC_TYPE fold
          acc = mac_q(acc, (i128)cf[i] * (i128)fold, p.cf.F + p.acc.F, p.acc);
        }
        break;
      }
      case ACDSP_TRANSPOSED: {
        // y[t] = reg_trans[N-1] after sample t: a chain that starts at reg_trans[N-2-t] of the
        // previous call (or 0) and adds x[t-j]*c[j] for j = min(t, N-1) .. 0.
        int jstart = N - 1;
        if (p.use_rt && t < N - 1) {
          jstart = (int)t;
          acc = p.rt[(int64_t)ch * N + (N - 2 - (int)t)];
        }
        for (int j = jstart; j >= 0; j--) { acc = mac_q(acc, (i128)w[-j] * cf[j], fp, p.acc); }
        break;
      }
      default: break;
    }
  }
  int64_t y = requant64(acc, p.acc.F, p.out);
  store_raw(p.y, (int64_t)ch * p.out_stride + t, p.out_eb, y);
}

static size_t fir_smem_bytes(const FirParams &p) { return sizeof(int64_t) * (size_t)(kTile + 2 * p.n_taps - 1); }

hipError_t launch_fir_generic(const FirParams &p, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((p.n + kTile - 1) / kTile), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_direct_kernel<false>, grid, dim3(kTile), fir_smem_bytes(p), s, p);
  return hipGetLastError();
}

hipError_t launch_fir_lossless64(const FirParams &p, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((p.n + kTile - 1) / kTile), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_direct_kernel<true>, grid, dim3(kTile), fir_smem_bytes(p), s, p);
  return hipGetLastError();
}

// hist_next[ch][j] = sample at local time n - hl + j (from this call's input, or the old history).
__global__ void fir_hist_update_kernel(FirParams p, void *hist_next) {
  const int ch = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p.hl; j += gridDim.x * blockDim.x) {
    int64_t g = p.n - p.hl + j;
    int64_t v = (g >= 0) ? load_raw(p.x, (int64_t)ch * p.in_stride + g, p.in_eb, p.in.S)
                         : load_raw(p.hist, (int64_t)ch * p.hl + p.hl + g, p.in_eb, p.in.S);
    store_raw(hist_next, (int64_t)ch * p.hl + j, p.in_eb, v);
  }
}

// the same for calls of at least hl samples whose tail is 16-byte aligned: a plain copy of the row tails, 16 bytes per lane
__global__ void fir_hist_copy16_kernel(const char *__restrict__ x, char *__restrict__ hist_next, int64_t row_bytes, int64_t tail_off, int vecs_per_row) {
  const int ch = blockIdx.y;
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  const v4i_ *src = (const v4i_ *)(x + (int64_t)ch * row_bytes + tail_off);
  v4i_ *dst = (v4i_ *)hist_next + (int64_t)ch * vecs_per_row;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < vecs_per_row; j += gridDim.x * blockDim.x) { dst[j] = src[j]; }
}

hipError_t launch_fir_hist_update(const FirParams &p, void *hist_next, hipStream_t s) {
  if (p.n <= 0 || p.hl <= 0) { return hipSuccess; }
  const int64_t hb = (int64_t)p.hl * p.in_eb, tail = (int64_t)(p.n - p.hl) * p.in_eb, rb = (int64_t)p.in_stride * p.in_eb;
  if (p.n >= p.hl && hb >= 4096 && hb % 16 == 0 && tail % 16 == 0 && rb % 16 == 0 && ((uintptr_t)p.x % 16) == 0 && ((uintptr_t)hist_next % 16) == 0) {
    const int vpr = (int)(hb / 16);
    dim3 grid((unsigned)((vpr + 255) / 256), (unsigned)p.n_ch);
    hipLaunchKernelGGL(fir_hist_copy16_kernel, grid, dim3(256), 0, s, (const char *)p.x, (char *)hist_next, rb, tail, vpr);
    return hipGetLastError();
  }
  dim3 grid((unsigned)((p.hl + 255) / 256), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_hist_update_kernel, grid, dim3(256), 0, s, p, hist_next);
  return hipGetLastError();
}

// reg_trans[i] after the last sample of this call (ac_fir_const_coeffs.h:286-294 unrolled in time):
//   r = (i >= n) ? rt_prev[i - n] : 0;  for m = min(i, n-1) .. 0:  r = ACC(x[n-1-m] * c[N-1-i+m] + r)
__global__ void fir_rt_update_kernel(FirParams p, int64_t *rt_next) {
  const int ch = blockIdx.y;
  const int N = p.n_taps;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) { return; }
  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  const int fp = p.in.F + p.cf.F;
  int64_t r = (i >= p.n) ? p.rt[(int64_t)ch * N + (i - p.n)] : 0;
  int64_t m0 = (i < p.n - 1) ? i : p.n - 1;
  for (int64_t m = m0; m >= 0; m--) {
    int64_t x = load_raw(p.x, (int64_t)ch * p.in_stride + (p.n - 1 - m), p.in_eb, p.in.S);
    r = mac_q(r, (i128)x * cg[N - 1 - i + m], fp, p.acc);
  }
  rt_next[(int64_t)ch * N + i] = r;
}

hipError_t launch_fir_rt_update(const FirParams &p, int64_t *rt_next, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((p.n_taps + 63) / 64), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_rt_update_kernel, grid, dim3(64), 0, s, p, rt_next);
  return hipGetLastError();
}

}  // namespace acdsp
