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
#include "fir_kernels.hpp"

namespace acdsp {

__device__ inline i128 pi_shl128(i128 v, int s) { return (i128)((u128)v << s); }
__device__ inline int64_t pi_mac(int64_t acc, i128 prod, int f_prod, const DFmt &A) {
  const int f = f_prod > A.F ? f_prod : A.F;
  return requant128(pi_shl128((i128)acc, f - A.F) + pi_shl128(prod, f - f_prod), f, A);
}

// sub-filter sum of phase j at local sample m (taps[i] = x[m - i])
__device__ int64_t polyintr_acc(const PolyIntrParams &p, int ch, int64_t m, int j) {
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

__global__ void polyintr_kernel(PolyIntrParams p) {
  const int ch = blockIdx.y;
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // output index of this call
  if (o >= p.n_out) { return; }
  const int IF = p.ifac;
  const int j = (int)(o % IF);
  const int64_t grp = o / IF + p.skip;                                 // local index of the sample that emits this group
  int64_t y;
  if (p.ftype == 2) {
    y = requant64(polyintr_acc(p, ch, grp, j), p.acc.F, p.out);
  } else {
    const int cj = p.corr[j];
    int64_t t1, t2;
    if (grp == 0) { t1 = p.saved[(int64_t)ch * IF + j]; t2 = p.saved[(int64_t)ch * IF + cj]; }   // sums of the previous call's last sample
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

// sums of the call's last sample -> the handle (they are emitted by the next call's first sample)
__global__ void polyintr_save_kernel(PolyIntrParams p, int64_t *saved_next) {
  const int ch = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.ifac) { return; }
  saved_next[(int64_t)ch * p.ifac + j] = polyintr_acc(p, ch, p.n - 1, j);
}

hipError_t launch_polyintr(const PolyIntrParams &p, int64_t *saved_next, hipStream_t s) {
  if (p.n_out > 0) {
    dim3 grid((unsigned)((p.n_out + 255) / 256), (unsigned)p.n_ch);
    hipLaunchKernelGGL(polyintr_kernel, grid, dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { return e; }
  }
  if (p.ftype != 2 && p.n > 0) {
    dim3 grid((unsigned)((p.ifac + 63) / 64), (unsigned)p.n_ch);
    hipLaunchKernelGGL(polyintr_save_kernel, grid, dim3(64), 0, s, p, saved_next);
    return hipGetLastError();
  }
  return hipSuccess;
}

}  // namespace acdsp
