// wide_kernels.hpp -- launch interface of wide.hip: formats of up to 128 bits.
#pragma once
#include "cic_kernels.hpp"
#include "fir_kernels.hpp"

namespace acdsp {

// Format of up to 128 bits as the wide kernels see it.
struct WFmt {
  int32_t W, F, S, Q, O;
  i128 lo, hi;  // representable raw range
};

inline WFmt make_wfmt(const acdsp_fmt_t &f) {
  WFmt d;
  d.W = f.W; d.F = f.W - f.I; d.S = f.S; d.Q = f.Q; d.O = f.O;
  if (f.S) {
    d.hi = (i128)(((u128)1 << (f.W - 1)) - 1);
    d.lo = -d.hi - 1;                      // (not -(1 << (W-1)): negating the 128-bit minimum is signed overflow)
  } else {
    d.lo = 0;
    d.hi = (i128)(((u128)1 << f.W) - 1);   // unsigned W = 128 is rejected at create
  }
  return d;
}

// FirParams (IN / COEFF formats, geometry, pointers; its acc / out DFmt members are unused here) + the wide formats.
struct FirWideParams {
  FirParams p;
  WFmt acc, out;
  const void *rt;   // [n_ch][n_taps] ACC raw words, 16 bytes each (use_rt)
};
hipError_t launch_fir_wide(const FirWideParams &pw, hipStream_t s);
hipError_t launch_fir_wide_rt_update(const FirWideParams &pw, void *rt_next, hipStream_t s);

struct CicWideParams {
  CicParams p;      // p.out is unused; p.w_int up to 128
  WFmt out;
};
// taps = z^-(N-1) * boxcar(R*me)^N (every tap < 2^31); n_out outputs per channel
hipError_t launch_cic_wide(const CicWideParams &pw, const int64_t *d_taps, int n_taps, int64_t n_out, hipStream_t s);

}  // namespace acdsp
