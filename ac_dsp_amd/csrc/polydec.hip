// polydec.hip -- exact, order-preserving kernel for the polyphase decimator (correctness path).
//
// Restates the loop nest of ac_poly_dec::run (reference include/ac_dsp/ac_poly_dec.h:109-128) for one
// output per thread: for df = DF-1 .. 0 the sub-filter accumulator acc1[df] takes NTAPS quantised MACs
// (`acc1[df] = acc1[df] + taps[tp*DF] * coeffs[tp + NTAPS*df]`, tp ascending) and is then added into
// `acc` (ACC_TYPE again); the shift register `taps[]` becomes a window of the input stream with the
// handle's history for samples before t = 0.  Output g belongs to the input group [g*DF, g*DF+DF).
// The lossless accumulator class does not come here: it is a decimating FIR
//   y[g] = sum_k h[k] x[g*DF + DF-1 - k],  h[df + tp*DF] = c[tp + NTAPS*df]
// and runs on the matrix cores (fir_gen.hip).
#include "fir_kernels.hpp"

namespace acdsp {

__device__ inline i128 pd_shl128(i128 v, int s) { return (i128)((u128)v << s); }

__global__ void polydec_generic_kernel(FirParams p, int ntaps, int df_n, int64_t n_out) {
  const int ch = blockIdx.y;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_out) { return; }
  const int fp = p.in.F + p.cf.F;
  auto xs = [&](int64_t t) -> int64_t {
    if (t >= 0) { return load_raw(p.x, (int64_t)ch * p.in_stride + t, p.in_eb, p.in.S); }
    if (t >= -(int64_t)p.hl) { return load_raw(p.hist, (int64_t)ch * p.hl + p.hl + t, p.in_eb, p.in.S); }
    return 0;
  };
  int64_t acc = 0;
  for (int df = df_n - 1; df >= 0; df--) {
    const int64_t newest = g * df_n + (df_n - 1 - df);   // taps[0] during this df iteration
    int64_t acc1 = 0;
    for (int tp = 0; tp < ntaps; tp++) {
      const i128 prod = (i128)xs(newest - (int64_t)tp * df_n) * p.coeffs[tp + ntaps * df];
      const int f = fp > p.acc.F ? fp : p.acc.F;
      acc1 = requant128(pd_shl128((i128)acc1, f - p.acc.F) + pd_shl128(prod, f - fp), f, p.acc);
    }
    acc = requant128((i128)acc + (i128)acc1, p.acc.F, p.acc);
  }
  store_raw(p.y, (int64_t)ch * p.out_stride + g, p.out_eb, requant64(acc, p.acc.F, p.out));
}

hipError_t launch_polydec_generic(const FirParams &p, int ntaps, int df, int64_t n_out, hipStream_t s) {
  if (n_out <= 0) { return hipSuccess; }
  dim3 grid((unsigned)((n_out + 127) / 128), (unsigned)p.n_ch);
  hipLaunchKernelGGL(polydec_generic_kernel, grid, dim3(128), 0, s, p, ntaps, df, n_out);
  return hipGetLastError();
}

}  // namespace acdsp
