// acdsp_dev.hpp -- device-side ac_fixed conversion rules shared by every kernel.
//
// Semantics restated (not copied) from the AC Datatypes rules the reference
// relies on at `acc += reg[i] * coeffs[i]` (reference
// include/ac_dsp/ac_fir_const_coeffs.h:196) and `data_out = acc` (:198):
// quantise with the destination's Q mode, then apply its O mode; the rounding
// carry takes part in the overflow decision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/acdsp.h"

namespace acdsp {

typedef __int128 i128;
typedef unsigned __int128 u128;

// Format as the kernels see it (wave-uniform, lives in SGPRs / kernarg).
struct DFmt {
  int32_t W, F, S, Q, O;
  int64_t lo, hi;  // representable raw range
};

inline DFmt make_dfmt(const acdsp_fmt_t &f) {
  DFmt d;
  d.W = f.W; d.F = f.W - f.I; d.S = f.S; d.Q = f.Q; d.O = f.O;
  if (f.S) {
    d.lo = (f.W >= 64) ? INT64_MIN : -(int64_t(1) << (f.W - 1));
    d.hi = (f.W >= 64) ? INT64_MAX : (int64_t(1) << (f.W - 1)) - 1;
  } else {
    d.lo = 0;
    d.hi = (f.W >= 63) ? INT64_MAX : (int64_t(1) << f.W) - 1;  // unsigned W = 64 is rejected at create
  }
  return d;
}

// Wrap to W bits and extend.
__host__ __device__ inline int64_t wrap64(int64_t x, int W, int S) {
  if (W >= 64) { return x; }
  int sh = 64 - W;
  return S ? (int64_t)((uint64_t)x << sh) >> sh : (int64_t)(((uint64_t)x << sh) >> sh);
}

// Increment decided by the quantisation mode.
// qb: most significant dropped bit, r: OR of the other dropped bits,
// neg: source is negative, lsb: LSB of the kept part.
__host__ __device__ inline int q_increment(int Q, int qb, int r, int neg, int lsb) {
  switch (Q) {
    case ACDSP_TRN: return 0;
    case ACDSP_RND: return qb;
    case ACDSP_TRN_ZERO: return neg & (qb | r);
    case ACDSP_RND_ZERO: return qb & (r | neg);
    case ACDSP_RND_INF: return qb & (r | (neg ^ 1));
    case ACDSP_RND_MIN_INF: return qb & r;
    case ACDSP_RND_CONV: return qb & (r | lsb);
    case ACDSP_RND_CONV_ODD: return qb & (r | (lsb ^ 1));
    default: return 0;
  }
}

// Overflow handling of an exact (possibly out-of-range) quotient q.
template <typename T /* int64_t or i128 */>
__host__ __device__ inline int64_t apply_overflow(T q, const DFmt &d) {
  bool under = q < (T)d.lo, over = q > (T)d.hi;
  switch (d.O) {
    case ACDSP_SAT: return under ? d.lo : (over ? d.hi : (int64_t)q);
    case ACDSP_SAT_ZERO: return (under || over) ? 0 : (int64_t)q;
    case ACDSP_SAT_SYM:
      if (d.S) {
        if (under || over) { return (q < 0) ? d.lo + 1 : d.hi; }
        return ((int64_t)q == d.lo && d.W > 1) ? d.lo + 1 : (int64_t)q;
      }
      return under ? d.lo : (over ? d.hi : (int64_t)q);
    default: return wrap64((int64_t)q, d.W, d.S);  // ACDSP_WRAP: low W bits
  }
}

// Exact x * 2^-f_src (128-bit) -> raw word of d.
__host__ __device__ inline int64_t requant128(i128 x, int f_src, const DFmt &d) {
  int sh = f_src - d.F;
  if (sh <= 0) {
    i128 q = (-sh >= 128) ? (i128)0 : (i128)((u128)x << (-sh));
    return apply_overflow<i128>(q, d);
  }
  int neg = x < 0, sticky = 0;
  if (sh > 126) {
    int k = sh - 126;
    i128 xs = (k >= 128) ? (neg ? (i128)-1 : (i128)0) : (x >> k);
    sticky = (k >= 128) ? (x != 0) : ((x - (i128)((u128)xs << k)) != 0);
    x = xs;
    sh = 126;
  }
  i128 q = x >> sh;
  i128 rem = x - (i128)((u128)q << sh);
  i128 half = ((i128)1) << (sh - 1);
  int qb = rem >= half;
  int r = ((rem & (half - 1)) != 0) | sticky;
  q += q_increment(d.Q, qb, r, neg, (int)(q & 1));
  return apply_overflow<i128>(q, d);
}

// 64-bit source, sh = f_src - f_dst.  Falls back to 128-bit only for left shifts.
__host__ __device__ inline int64_t requant64(int64_t x, int f_src, const DFmt &d) {
  int sh = f_src - d.F;
  if (sh <= 0) {
    if (sh == 0) { return apply_overflow<int64_t>(x, d); }
    return requant128((i128)x, f_src, d);
  }
  if (sh >= 64) { return requant128((i128)x, f_src, d); }
  int64_t q = x >> sh;
  uint64_t rem = (uint64_t)x & ((uint64_t(1) << sh) - 1);
  uint64_t half = uint64_t(1) << (sh - 1);
  int qb = (rem & half) != 0;
  int r = (rem & (half - 1)) != 0;
  int inc = q_increment(d.Q, qb, r, x < 0, (int)(q & 1));
  if (inc && q == INT64_MAX) { return requant128((i128)x, f_src, d); }
  return apply_overflow<int64_t>(q + inc, d);
}

// Load / store a raw word from an IN / OUT container of eb bytes.
__device__ inline int64_t load_raw(const void *p, int64_t idx, int eb, int S) {
  if (eb == 2) { return S ? (int64_t)((const int16_t *)p)[idx] : (int64_t)((const uint16_t *)p)[idx]; }
  if (eb == 4) { return S ? (int64_t)((const int32_t *)p)[idx] : (int64_t)((const uint32_t *)p)[idx]; }
  return ((const int64_t *)p)[idx];
}
__device__ inline void store_raw(void *p, int64_t idx, int eb, int64_t v) {
  if (eb == 2) { ((int16_t *)p)[idx] = (int16_t)v; }
  else if (eb == 4) { ((int32_t *)p)[idx] = (int32_t)v; }
  else { ((int64_t *)p)[idx] = v; }
}

// XCD-affine work order.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup L runs on XCD L % 8: observed rule, only
// speed depends on it).  In plain launch order every XCD touches every 8th piece of a stream; remapped, XCD k walks the k-th
// contiguous eighth of the launch in memory order (tools/copy_probe2.hip: +3 - 5 % on a copy with 16 KB spans).  `on` is set by the
// host only when the launch has a multiple of 8 workgroups.  Returns the remapped (x, y) block index.
__device__ __forceinline__ void xcd_remap(int on, int &bx, int &by) {
  bx = blockIdx.x; by = blockIdx.y;
  if (on) {
    const int64_t T = (int64_t)gridDim.x * gridDim.y, L = (int64_t)by * gridDim.x + bx;
    const int64_t L2 = (L & 7) * (T >> 3) + (L >> 3);
    by = (int)(L2 / gridDim.x);
    bx = (int)(L2 - (int64_t)by * gridDim.x);
  }
}
// Tuning knobs are environment variables read ONCE per process; with ACDSP_TUNE_LIVE set they are re-read at every launch, so that one
// process (tools/knob_sweep.py) can compare variants on the same allocations -- the streaming rows move 3 - 8 % with the placement of
// their buffers from process to process (profiles/r3_placement_modes.txt), more than most knobs.
inline bool tune_live() {
  static const bool l = getenv("ACDSP_TUNE_LIVE") != nullptr;
  return l;
}
#define ACDSP_TUNE_ENV(var, name)                    \
  static const char *var##_once = getenv(name);      \
  const char *var = ::acdsp::tune_live() ? getenv(name) : var##_once
// default of a kernel family, overridden by ACDSP_XCD_MAP=0 / 1 (A/B knob)
inline bool xcd_map_wanted(bool family_default) {
  ACDSP_TUNE_ENV(e, "ACDSP_XCD_MAP");
  return e ? atoi(e) != 0 : family_default;
}

}  // namespace acdsp
