// engine_common.hpp -- what the translation units of the C-ABI layer share (round 5: engine.hip, 2 700 lines of every operator family,
// split per family: engine.hip = entry points common to all + diagnostics + state blobs + stream files, engine_fir.hip, engine_cic.hip,
// engine_ddc.hip, engine_poly.hip, engine_misc.hip).  Host-side object model: one handle = n independent reference filter objects whose
// state lives in HBM and carries across run() calls.  There is no CPU compute path: every run() launches HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "wide_kernels.hpp"
#include "cic_kernels.hpp"
#include "fir_kernels.hpp"

namespace acdsp {
namespace eng {

// sets the calling thread's acdsp_last_error() message and returns `code` (engine.hip)
int fail(int code, const char *fmt, ...);
// Device check of every entry point (engine.hip)
int check_device(int device);

// Is `s` recording a HIP graph?  A replayed graph re-runs the kernels with the HOST-side bookkeeping of capture time baked into
// their arguments (decimation / interpolation phase, "first call of the stream" special cases), so calls whose bookkeeping would
// not return to the captured value are refused while capturing instead of replaying the wrong phase silently.
inline bool stream_is_capturing(hipStream_t s) {
  // The legacy NULL stream cannot be captured, and asking about it while ANOTHER stream is in a global-mode capture returns an
  // error that may invalidate that capture and stays behind as the thread's last error (the next launch's hipGetLastError would
  // report it as a kernel failure): the host-buffer paths, which run on the NULL stream, never ask.
  if (s == nullptr) { return false; }
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing(s, &st);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) { return fail(ACDSP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
  } while (0)

// history buffer the state kernel of a call writes: the current one (in place) when the call's input alone defines the new
// history, else the other one
inline int hist_next_index(int cur, bool in_place) { return in_place ? cur : (cur ^ 1); }

// max_w: 64 for IN / COEFF and every class without a wide path; 128 for ACC / OUT of the FIR classes and OUT of the CIC classes
// (wide.hip).  Unsigned types of the full container width are not representable in the signed raw words and are refused.
inline int check_fmt(const acdsp_fmt_t &f, const char *name, int max_w = 64) {
  if (f.W < 1 || f.W > max_w) { return fail(ACDSP_EUNSUPPORTED, "%s: W=%d outside 1..%d", name, f.W, max_w); }
  if (!f.S && (f.W == 64 || f.W == 128)) { return fail(ACDSP_EUNSUPPORTED, "%s: unsigned W=%d not supported", name, f.W); }
  if (f.Q < 0 || f.Q > ACDSP_RND_CONV_ODD) { return fail(ACDSP_EINVAL, "%s: bad Q mode %d", name, f.Q); }
  if (f.O < 0 || f.O > ACDSP_SAT_SYM) { return fail(ACDSP_EINVAL, "%s: bad O mode %d", name, f.O); }
  if (f.S != 0 && f.S != 1) { return fail(ACDSP_EINVAL, "%s: S must be 0 or 1", name); }
  return ACDSP_OK;
}

// ACDSP_TRACE=1: one stderr line per handle created and destroyed (with the number of run() calls it launched kernels for) -- how
// tests/test_cpp_gpu.py tells a testbench that reached the HIP kernels from one the header's host loop served (include/ac_dsp/acdsp_engine.h)
inline bool trace_handles() {
  static const bool t = getenv("ACDSP_TRACE") != nullptr;
  return t;
}

inline int elem_bytes(int W) { return W <= 16 ? 2 : (W <= 32 ? 4 : (W <= 64 ? 8 : 16)); }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// HIP-event timing of the main kernel of each run(), recorded on the launch stream.
// A ring of event pairs so that a whole timed region can be read back afterwards.
struct Timer {
  static const int kRing = 64;
  hipEvent_t e0[kRing], e1[kRing];
  int64_t count = 0;  // runs recorded so far
  bool ok = false;
  int init() {
    for (int i = 0; i < kRing; i++) { e0[i] = nullptr; e1[i] = nullptr; }
    for (int i = 0; i < kRing; i++) {
      HIP_TRY(hipEventCreate(&e0[i]));
      HIP_TRY(hipEventCreate(&e1[i]));
    }
    ok = true;
    return ACDSP_OK;
  }
  void destroy() {
    if (!ok) { return; }
    for (int i = 0; i < kRing; i++) {
      if (e0[i]) { (void)hipEventDestroy(e0[i]); }
      if (e1[i]) { (void)hipEventDestroy(e1[i]); }
    }
  }
  hipEvent_t start() { return e0[count % kRing]; }
  hipEvent_t stop() { return e1[count % kRing]; }
  void commit() { count++; }
  // average / minimum over the last k runs
  int stats(int k, float *avg, float *mn) {
    if (count == 0) { return fail(ACDSP_ESTATE, "no run() recorded yet"); }
    if (k < 1) { k = 1; }
    if (k > kRing) { k = kRing; }
    if (k > count) { k = (int)count; }
    double sum = 0;
    float lo = 1e30f;
    for (int i = 0; i < k; i++) {
      const int64_t idx = (count - 1 - i) % kRing;
      float ms = 0;
      HIP_TRY(hipEventSynchronize(e1[idx]));
      HIP_TRY(hipEventElapsedTime(&ms, e0[idx], e1[idx]));
      sum += ms;
      if (ms < lo) { lo = ms; }
    }
    if (avg) { *avg = (float)(sum / k); }
    if (mn) { *mn = lo; }
    return ACDSP_OK;
  }
};

struct Staging {
  void *d_in = nullptr, *d_out = nullptr;
  size_t cap_in = 0, cap_out = 0;
  // Small calls (the drop-in run() of one channel, ac_fir_prog_coeffs: ONE sample per call, reference ac_fir_prog_coeffs.h:281):
  // a pinned, device-mapped host buffer the kernels read and write directly -- no H2D / D2H copy calls, one synchronisation.
  static const size_t kPinBytes = 64 * 1024;
  void *pin_in = nullptr, *pin_out = nullptr;
  int ensure_pinned() {
    if (!pin_in) { HIP_TRY(hipHostMalloc(&pin_in, kPinBytes, hipHostMallocMapped)); }
    if (!pin_out) { HIP_TRY(hipHostMalloc(&pin_out, kPinBytes, hipHostMallocMapped)); }
    return ACDSP_OK;
  }
  int ensure(size_t bin, size_t bout) {
    if (bin > cap_in) {
      if (d_in) { (void)hipFree(d_in); }
      HIP_TRY(hipMalloc(&d_in, bin));
      cap_in = bin;
    }
    if (bout > cap_out) {
      if (d_out) { (void)hipFree(d_out); }
      HIP_TRY(hipMalloc(&d_out, bout));
      cap_out = bout;
    }
    return ACDSP_OK;
  }
  void destroy() {
    if (d_in) { (void)hipFree(d_in); }
    if (d_out) { (void)hipFree(d_out); }
    if (pin_in) { (void)hipHostFree(pin_in); }
    if (pin_out) { (void)hipHostFree(pin_out); }
  }
};

}  // namespace eng
}  // namespace acdsp

using namespace acdsp;   // (the handle structs live in the global namespace: they are the opaque types of include/acdsp.h)
using acdsp::eng::Timer;
using acdsp::eng::Staging;

struct acdsp_fir {
  acdsp_fir_desc_t d;
  int in_eb, out_eb, hl;
  bool use_rt, lossless, coeffs_set;
  // AC_SAT / AC_SAT_SYM / AC_SAT_ZERO accumulators (round 5): lossless_shape = the exact-sum conditions with the overflow mode left out (create);
  // sat_free = no partial sum of the CURRENT coefficient set can reach the type's bounds, so the saturation is dead code and the handle
  // runs the classes of a wrapping accumulator (set_coeffs; every FirParams of the handle then carries AC_WRAP: fir_acc_fmt)
  bool lossless_shape = false, sat_free = false;
  bool wide = false;   // ACC_TYPE or OUT_TYPE wider than 64 bits: wide.hip (reg_trans words are then 16 bytes)
  bool small_call = false;   // set by run_host around a call that fits the pinned buffers (launch-bound: see acdsp_fir_run)
  int rt_eb = 8;
  int path;
  void *d_hist[2] = {nullptr, nullptr};
  int64_t *d_rt[2] = {nullptr, nullptr};
  int cur = 0;
  // TRANSPOSED with loadable coefficients, exact-sum class (rt_hybrid): reg_trans[] differs from an input history only while partial sums
  // of an EARLIER coefficient set are still in it -- for the n_taps - 1 samples behind a coefficient change (or a loaded state blob).  Those
  // samples run the exact-order kernel on reg_trans; everything else is the same dot product as SHIFT_REG and runs the matrix-core kernels
  // on the input history, which is kept up to date by every call.  reg_trans is rebuilt from the history (rt_from_hist) when it is asked for.
  // unsigned 16-bit samples on the int8 MFMA kernel: x_u = (x_u ^ 0x8000 as int16) + 32768 -- the kernel flips the top bit as it splits
  // the samples into byte planes (FirParams::in_flip) and 32768 * sum(c) rides in the correction constant; rows and state stay raw
  bool in_flip = false;
  bool rt_hybrid = false, rt_valid = true;
  int64_t rt_since = 0;         // samples since the last coefficient change / state load, saturating at n_taps - 1
  int cur_rt = 0;               // rt_hybrid: index of the current reg_trans buffer (the history has `cur`)
  int64_t *d_coeffs = nullptr;
  uint32_t *d_frag = nullptr;   // [n_sets][2][nb][64][4] Toeplitz byte-plane fragments
  int64_t *d_corr = nullptr;    // [n_sets] 128 * sum(c)
  FirMfmaPlan plan;             // worst case over the coefficient sets (bounds for the epilogue choice)
  bool mfma_ok = false;
  int mfma_cshift = 0;          // the fragments hold the coefficients scaled by 2^mfma_cshift (narrow types: engine_fir.hip, set_coeffs)
  uint32_t *d_gfrag = nullptr;  // fragments of the generalised (wide-input) MFMA kernel
  FirGenPlan gplan;
  bool gen_ok = false;
  // class B on the matrix cores (fir_gen.hip, LZ ring shapes): gplan / d_gfrag hold the plan of the effective taps, lzp the residue table
  bool lz_ok = false;
  FirLossyPlan lzp;
  uint32_t *d_lzcl = nullptr;
  int64_t n_runs = 0;             // run() calls that launched kernels (ACDSP_TRACE)
  int kclass = 0;                 // acdsp_fir_kernel_class
  std::vector<int64_t> h_coeffs;  // last coefficient set (for clone)
  Timer tm;
  Staging st;
};

struct acdsp_cic {
  acdsp_cic_desc_t d;
  acdsp_fmt_t it;
  int in_eb, out_eb, hl, me;
  // decimator through its FIR identity on the matrix cores (fir_gen.hip): taps, and per (first mod 16) plans / fragments
  std::vector<int64_t> h_taps;
  bool gen_ok = false;
  bool gen_have[16] = {false};
  FirGenPlan gen_plan[16];
  uint32_t *d_gfrag = nullptr;   // [16][3*8*64*4]
  // decimator in two stages (cic2.hip): R = c2_R1 * c2_R2, stage-1 taps z^-(N-1) boxcar(R1)^N with their per (first mod 16) plans / fragments
  bool c2_ok = false;
  int c2_R1 = 0, c2_R2 = 0, c2_wu = 0;
  std::vector<int64_t> c2_taps;
  bool c2_have[16] = {false};
  FirGenPlan c2_plan[16];
  uint32_t *d_c2frag = nullptr;  // [16][3*8*64*4]
  int warm = 0;                  // inputs the recurrence kernel simulates in front of a chunk (the filter memory; hl may be longer: cic2.hip)
  int64_t *d_taps = nullptr;     // interpolator: the identity's taps for the polyphase kernel
  // interpolator on the matrix cores (fir_up.hip): per-phase taps E_r[k] = h[r + R k]
  bool up_ok = false;
  int up_px = 0;
  FirUpPlan up_plan;
  uint32_t *d_upfrag = nullptr;
  int64_t *d_upcorr = nullptr;
  int last_path = 0;
  bool wide = false;    // INT_TYPE or OUT_TYPE wider than 64 bits: both directions through cic_wide_kernel (wide.hip)
  int64_t t_total = 0;  // inputs consumed so far (all calls)
  void *d_hist[2] = {nullptr, nullptr};
  int cur = 0;
  Timer tm;
  Staging st;
};

// DDC cascade handle (engine_ddc.hip; the state blobs of engine.hip read it)
struct acdsp_ddc {
  acdsp_cic_t cic = nullptr;     // stage A: parameter checks, INT_TYPE, FIR-identity taps; runs the stage in two-kernel mode
  acdsp_fir_t fir = nullptr;     // stage B: coefficient checks; runs the stage in two-kernel mode
  bool fused = false;            // decided at creation / coefficient load; a handle never switches modes mid-stream
  // fused mode: the only state is the input history (stage B's window is recomputed from it) and the input count
  int hl = 0;
  void *d_hist[2] = {nullptr, nullptr};
  int cur = 0;
  int64_t t_total = 0;
  bool haveA[16] = {false};
  FirGenPlan planA[16], planB;
  uint32_t *d_fragA = nullptr, *d_fragB = nullptr;
  bool coeffs_set = false;
  // two-kernel mode: intermediate stream
  void *d_mid = nullptr;
  int64_t mid_cap = 0;
  Timer tm;
};


namespace acdsp {
namespace eng {
// FIR helpers other families use (engine_fir.hip)
std::vector<int64_t> effective_coeffs(const int64_t *c, int N, int ftype);
int internal_ftype(int kind, int ftype);
// rt_hybrid: rebuild reg_trans[] from the input history (engine_fir.hip; the state blobs need it)
int32_t fir_rt_from_hist(acdsp_fir *h);
}  // namespace eng
}  // namespace acdsp
