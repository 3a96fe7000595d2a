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
  const int64_t t0 = p.t_begin + (int64_t)blockIdx.x * kTile;
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
  if (p.n <= p.t_begin) { return hipSuccess; }
  dim3 grid((unsigned)((p.n - p.t_begin + kTile - 1) / kTile), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_direct_kernel<false>, grid, dim3(kTile), fir_smem_bytes(p), s, p);
  return hipGetLastError();
}

hipError_t launch_fir_lossless64(const FirParams &p, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((p.n + kTile - 1) / kTile), (unsigned)p.n_ch);
  hipLaunchKernelGGL(fir_direct_kernel<true>, grid, dim3(kTile), fir_smem_bytes(p), s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Class B on 16-bit types (lossy accumulator, AC_TRN / AC_RND into AC_WRAP): acc = sum_k Q(x[t-k] c[k]) mod 2^W_acc with
// Q(p) = (p + rnd) >> s, s = F_in + F_c - F_acc in 1 .. 30.  `acc += p` quantises acc 2^s + p, and acc is a multiple of the quantum, so
// every product is quantised on its own and the additions are exact: the sum has no order (SHIFT_REG, ROTATE_SHIFT, C_BUFF and the
// ascending ac_fir_reg_share core alike) and AC_WRAP is applied once at the end.  The products fit int32 (16 x 16 bits).
// One lane = 8 consecutive outputs; per 8 taps it takes ONE aligned 16-byte LDS read (the 8 samples in front of its window; the other
// half of the 16-sample register window is the previous read), 64 x { v_mad_i32_i24, v_ashrrev_i32, v_add } and one 64-bit accumulate
// per output (G8: eight quantised products cannot leave int32 for s >= 3; s = 1, 2 accumulate in 64 bits per tap).  Coefficients are
// wave-uniform scalar loads.  A 256-thread block covers 2048 outputs of one channel.
constexpr int kLossyOut = 8, kLossyTile = 256 * kLossyOut;

bool fir_lossy_fast_ok(const FirParams &p) {
  const int s = p.in.F + p.cf.F - p.acc.F;
  const bool order_free = p.ftype == ACDSP_SHIFT_REG || p.ftype == ACDSP_ROTATE_SHIFT || p.ftype == ACDSP_C_BUFF || p.ftype == kRsShiftReg ||
                          (p.ftype == ACDSP_TRANSPOSED && !p.use_rt);
  return order_free && p.in_eb == 2 && (p.in.S ? p.in.W <= 16 : p.in.W <= 15) && (p.cf.S ? p.cf.W <= 16 : p.cf.W <= 15) &&
         p.acc.O == ACDSP_WRAP && (p.acc.Q == ACDSP_TRN || p.acc.Q == ACDSP_RND) && p.acc.W <= 64 && s >= 1 && s <= 30 && p.n_taps <= 2048 &&
         !p.hist_next;
}

template <bool G8>
__global__ void __launch_bounds__(256) fir_lossy_kernel(FirParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int16_t *win = (int16_t *)smem_raw;                     // win[j] = x[t0 - NP + j], j in [0, NP + kLossyTile)
  const int N = p.n_taps, N8 = (N + 7) & ~7, NP = N8 + 8;
  const int ch = blockIdx.y, tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * kLossyTile;
  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;
  for (int j = tid; j < NP + kLossyTile; j += 256) {
    const int64_t g = t0 - NP + j;
    int16_t v = 0;
    if (g >= 0) { if (g < p.n) { v = xrow[g]; } }
    else if (!p.use_rt && g >= -(int64_t)p.hl) { v = hrow[g]; }
    win[j] = v;
  }
  __syncthreads();
  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  const int s = p.in.F + p.cf.F - p.acc.F;
  const int rnd = p.acc.Q == ACDSP_RND ? (1 << (s - 1)) : 0;
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  const int base = NP + kLossyOut * tid;                  // window index of this lane's first output
  int64_t acc[kLossyOut];
#pragma unroll
  for (int j = 0; j < kLossyOut; j++) { acc[j] = 0; }
  int xs[16];                                             // samples win[base - i0 - 8 .. base - i0 + 7]
  {
    const v4i_ hi = *(const v4i_ *)(win + base);
#pragma unroll
    for (int q = 0; q < 4; q++) { xs[8 + 2 * q] = (int)(int16_t)hi[q]; xs[9 + 2 * q] = hi[q] >> 16; }
  }
  for (int i0 = 0; i0 < N8; i0 += 8) {
    const v4i_ lo = *(const v4i_ *)(win + base - i0 - 8);
#pragma unroll
    for (int q = 0; q < 4; q++) { xs[2 * q] = (int)(int16_t)lo[q]; xs[2 * q + 1] = lo[q] >> 16; }
    int part[kLossyOut];
#pragma unroll
    for (int j = 0; j < kLossyOut; j++) { part[j] = 0; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int c = (i0 + k < N) ? (int)cg[i0 + k] : 0;   // wave-uniform
      const int r = (i0 + k < N) ? rnd : 0;
#pragma unroll
      for (int j = 0; j < kLossyOut; j++) {
        const int q = (xs[8 + j - k] * c + r) >> s;       // x[t + j - (i0 + k)]
        if (G8) { part[j] += q; } else { acc[j] += q; }
      }
    }
#pragma unroll
    for (int j = 0; j < kLossyOut; j++) {
      if (G8) { acc[j] += part[j]; }
      xs[8 + j] = xs[j];                                   // the new low half is the next iteration's high half
    }
  }
  const int64_t t = t0 + kLossyOut * tid;
#pragma unroll
  for (int j = 0; j < kLossyOut; j++) {
    if (t + j < p.n) {
      const int64_t a = wrap64(acc[j], p.acc.W, p.acc.S);
      store_raw(p.y, (int64_t)ch * p.out_stride + t + j, p.out_eb, requant64(a, p.acc.F, p.out));
    }
  }
}

// Class C on 16-bit types, the common case of it: a SATURATING accumulator of up to 32 bits (AC_SAT; AC_TRN / AC_RND).  acc = sat(acc + Q(p)) after
// every tap, in the reference's tap order -- i = N-1 .. 0 for SHIFT_REG / ROTATE_SHIFT (ac_fir_const_coeffs.h:190-220), ascending for C_BUFF
// (:226-237) and ac_fir_reg_share's SHIFT_REG core -- so the chain of one output is serial, and a lane runs its 8 outputs' chains side by side:
// v_mad_i32_i24, v_ashrrev_i32, v_add_i32 clamp (the sum of two in-range words cannot be told from a wrapped one otherwise), v_med3_i32.
// Same register window as fir_lossy_kernel; DESC walks it from the oldest sample up.
bool fir_satacc_fast_ok(const FirParams &p) {
  const int s = p.in.F + p.cf.F - p.acc.F;
  const bool order_ok = p.ftype == ACDSP_SHIFT_REG || p.ftype == ACDSP_ROTATE_SHIFT || p.ftype == ACDSP_C_BUFF || p.ftype == kRsShiftReg;
  return order_ok && p.in_eb == 2 && (p.in.S ? p.in.W <= 16 : p.in.W <= 15) && (p.cf.S ? p.cf.W <= 16 : p.cf.W <= 15) &&
         p.acc.O == ACDSP_SAT && (p.acc.Q == ACDSP_TRN || p.acc.Q == ACDSP_RND) && p.acc.W <= (p.acc.S ? 32 : 31) && s >= 0 && s <= 30 &&
         p.n_taps <= 2048 && !p.hist_next && !p.use_rt;
}

template <bool DESC>
__global__ void __launch_bounds__(256) fir_satacc_kernel(FirParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int16_t *win = (int16_t *)smem_raw;
  const int N = p.n_taps, N8 = (N + 7) & ~7, NP = N8 + 8;
  const int ch = blockIdx.y, tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * kLossyTile;
  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;
  for (int j = tid; j < NP + kLossyTile; j += 256) {
    const int64_t g = t0 - NP + j;
    int16_t v = 0;
    if (g >= 0) { if (g < p.n) { v = xrow[g]; } }
    else if (g >= -(int64_t)p.hl) { v = hrow[g]; }
    win[j] = v;
  }
  __syncthreads();
  const int64_t *cg = p.coeffs + (p.coeffs_per_channel ? (int64_t)ch * N : 0);
  const int s = p.in.F + p.cf.F - p.acc.F;
  const int rnd = (p.acc.Q == ACDSP_RND && s > 0) ? (1 << (s - 1)) : 0;
  const int lo = (int)p.acc.lo, hi = (int)p.acc.hi;
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  const int base = NP + kLossyOut * tid;
  int acc[kLossyOut];
#pragma unroll
  for (int j = 0; j < kLossyOut; j++) { acc[j] = 0; }
  int xs[16];                                             // samples win[base - i0 - 8 .. base - i0 + 7]
  auto unpack = [&](const v4i_ &v, int at) {
#pragma unroll
    for (int q = 0; q < 4; q++) { xs[at + 2 * q] = (int)(int16_t)v[q]; xs[at + 2 * q + 1] = v[q] >> 16; }
  };
  auto tap = [&](int i0, int k) {
    const int c = (i0 + k < N) ? (int)cg[i0 + k] : 0;     // wave-uniform; padded taps add (0 + rnd) >> s = 0
    const int r = (i0 + k < N) ? rnd : 0;
#pragma unroll
    for (int j = 0; j < kLossyOut; j++) {
      const int q = (xs[8 + j - k] * c + r) >> s;
      int a = __builtin_elementwise_add_sat(acc[j], q);
      acc[j] = a < lo ? lo : (a > hi ? hi : a);
    }
  };
  if (DESC) {
    unpack(*(const v4i_ *)(win + base - (N8 - 8) - 8), 0);
    for (int i0 = N8 - 8; i0 >= 0; i0 -= 8) {
      unpack(*(const v4i_ *)(win + base - i0), 8);
#pragma unroll
      for (int k = 7; k >= 0; k--) { tap(i0, k); }
#pragma unroll
      for (int j = 0; j < 8; j++) { xs[j] = xs[8 + j]; }   // the high half is the next (newer) iteration's low half
    }
  } else {
    unpack(*(const v4i_ *)(win + base), 8);
    for (int i0 = 0; i0 < N8; i0 += 8) {
      unpack(*(const v4i_ *)(win + base - i0 - 8), 0);
#pragma unroll
      for (int k = 0; k < 8; k++) { tap(i0, k); }
#pragma unroll
      for (int j = 0; j < 8; j++) { xs[8 + j] = xs[j]; }
    }
  }
  const int64_t t = t0 + kLossyOut * tid;
#pragma unroll
  for (int j = 0; j < kLossyOut; j++) {
    if (t + j < p.n) { store_raw(p.y, (int64_t)ch * p.out_stride + t + j, p.out_eb, requant64((int64_t)acc[j], p.acc.F, p.out)); }
  }
}

hipError_t launch_fir_satacc(const FirParams &p, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  const int N8 = (p.n_taps + 7) & ~7;
  const size_t lds = sizeof(int16_t) * (size_t)(N8 + 8 + kLossyTile);
  dim3 grid((unsigned)((p.n + kLossyTile - 1) / kLossyTile), (unsigned)p.n_ch);
  if (p.ftype == ACDSP_SHIFT_REG || p.ftype == ACDSP_ROTATE_SHIFT) { hipLaunchKernelGGL(fir_satacc_kernel<true>, grid, dim3(256), lds, s, p); }
  else { hipLaunchKernelGGL(fir_satacc_kernel<false>, grid, dim3(256), lds, s, p); }
  return hipGetLastError();
}

hipError_t launch_fir_lossy(const FirParams &p, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  const int N8 = (p.n_taps + 7) & ~7;
  const size_t lds = sizeof(int16_t) * (size_t)(N8 + 8 + kLossyTile);
  dim3 grid((unsigned)((p.n + kLossyTile - 1) / kLossyTile), (unsigned)p.n_ch);
  if (p.in.F + p.cf.F - p.acc.F >= 3) { hipLaunchKernelGGL(fir_lossy_kernel<true>, grid, dim3(256), lds, s, p); }
  else { hipLaunchKernelGGL(fir_lossy_kernel<false>, grid, dim3(256), lds, s, p); }
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
