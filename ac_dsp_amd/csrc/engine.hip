// engine.hip -- the C ABI of libacdsp.so (declared in include/acdsp.h).
//
// Host-side object model: one handle = n_channels independent reference filter
// objects (reference: one ac_fir_* / ac_cic_* instance each) whose state lives
// in HBM and carries across run() calls.  There is no CPU compute path here:
// every run() launches HIP kernels, and creation fails without a gfx950 device.
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

using namespace acdsp;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// Is `s` recording a HIP graph?  A replayed graph re-runs the kernels with the HOST-side bookkeeping of capture time baked into
// their arguments (decimation / interpolation phase, "first call of the stream" special cases), so calls whose bookkeeping would
// not return to the captured value are refused while capturing instead of replaying the wrong phase silently.
static bool stream_is_capturing(hipStream_t s) {
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
static inline int hist_next_index(int cur, bool in_place) { return in_place ? cur : (cur ^ 1); }

// Device check of every entry point.  The architecture test (hipGetDeviceProperties: ~100 us) runs once per device and
// process; later calls only make `device` the calling thread's current device when it is not already.
int check_device(int device) {
  static std::atomic<uint64_t> verified{0};   // bit d: device d has been seen to be a gfx950
  if (device >= 0 && device < 64 && ((verified.load(std::memory_order_relaxed) >> device) & 1)) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) { return ACDSP_OK; }
    HIP_TRY(hipSetDevice(device));
    return ACDSP_OK;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { return fail(ACDSP_ENODEVICE, "no HIP device visible"); }
  if (device < 0 || device >= n) { return fail(ACDSP_EINVAL, "device %d out of range (%d devices)", device, n); }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    return fail(ACDSP_ENODEVICE, "device %d is %s; this engine is built for gfx950 only", device, prop.gcnArchName);
  }
  HIP_TRY(hipSetDevice(device));
  if (device < 64) { verified.fetch_or(uint64_t(1) << device, std::memory_order_relaxed); }
  return ACDSP_OK;
}

// max_w: 64 for IN / COEFF and every class without a wide path; 128 for ACC / OUT of the FIR classes and OUT of the CIC classes
// (wide.hip).  Unsigned types of the full container width are not representable in the signed raw words and are refused.
int check_fmt(const acdsp_fmt_t &f, const char *name, int max_w = 64) {
  if (f.W < 1 || f.W > max_w) { return fail(ACDSP_EUNSUPPORTED, "%s: W=%d outside 1..%d", name, f.W, max_w); }
  if (!f.S && (f.W == 64 || f.W == 128)) { return fail(ACDSP_EUNSUPPORTED, "%s: unsigned W=%d not supported", name, f.W); }
  if (f.Q < 0 || f.Q > ACDSP_RND_CONV_ODD) { return fail(ACDSP_EINVAL, "%s: bad Q mode %d", name, f.Q); }
  if (f.O < 0 || f.O > ACDSP_SAT_SYM) { return fail(ACDSP_EINVAL, "%s: bad O mode %d", name, f.O); }
  if (f.S != 0 && f.S != 1) { return fail(ACDSP_EINVAL, "%s: S must be 0 or 1", name); }
  return ACDSP_OK;
}

int elem_bytes(int W) { return W <= 16 ? 2 : (W <= 32 ? 4 : (W <= 64 ? 8 : 16)); }
int round_up(int x, int m) { return (x + m - 1) / m * m; }

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

}  // namespace

namespace {
// dst[ch][j] = src[ch][j] ^ 0x8000 for the n samples of a call (row stride ds, multiple of 16: the tail up to ds is zero-filled in the
// flipped domain's zero = 0x8000 ^ 0 ... it is never used by an output that exists) and for the hl history samples
__global__ void flip16_kernel(const uint16_t *x, int64_t xs, int64_t n, uint16_t *dx, int64_t ds, const uint16_t *hist, uint16_t *dh, int hl) {
  const int ch = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ds + hl; j += (int64_t)gridDim.x * blockDim.x) {
    if (j < ds) { dx[(int64_t)ch * ds + j] = (uint16_t)((j < n ? x[(int64_t)ch * xs + j] : 0) ^ 0x8000u); }
    else { dh[(int64_t)ch * hl + (j - ds)] = (uint16_t)(hist[(int64_t)ch * hl + (j - ds)] ^ 0x8000u); }
  }
}
hipError_t launch_flip16(const void *x, int64_t xs, int64_t n, void *dx, int64_t ds, const void *hist, void *dh, int hl, int n_ch, hipStream_t s) {
  int64_t blocks = (ds + hl + 1023) / 1024;
  if (blocks > 4096) { blocks = 4096; }
  hipLaunchKernelGGL(flip16_kernel, dim3((unsigned)blocks, (unsigned)n_ch), dim3(256), 0, s, (const uint16_t *)x, xs, n, (uint16_t *)dx, ds,
                     (const uint16_t *)hist, (uint16_t *)dh, hl);
  return hipGetLastError();
}
}  // namespace

struct acdsp_fir {
  acdsp_fir_desc_t d;
  int in_eb, out_eb, hl;
  bool use_rt, lossless, coeffs_set;
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
  // unsigned 16-bit samples on the int8 MFMA kernel (round 4): x_u = (x_u ^ 0x8000 as int16) + 32768, so a flipped copy of the call's samples
  // and of the history goes through the signed kernel and 32768 * sum(c) rides in the correction constant; the state stays raw
  bool in_flip = false;
  Staging st_u;
  bool rt_hybrid = false, rt_valid = true;
  int64_t rt_since = 0;         // samples since the last coefficient change / state load, saturating at n_taps - 1
  int cur_rt = 0;               // rt_hybrid: index of the current reg_trans buffer (the history has `cur`)
  int64_t *d_coeffs = nullptr;
  uint32_t *d_frag = nullptr;   // [n_sets][2][nb][64][4] Toeplitz byte-plane fragments
  int64_t *d_corr = nullptr;    // [n_sets] 128 * sum(c)
  FirMfmaPlan plan;             // worst case over the coefficient sets (bounds for the epilogue choice)
  bool mfma_ok = false;
  uint32_t *d_gfrag = nullptr;  // fragments of the generalised (wide-input) MFMA kernel
  FirGenPlan gplan;
  bool gen_ok = false;
  // class B on the matrix cores (fir_gen.hip, LZ ring shapes): gplan / d_gfrag hold the plan of the effective taps, lzp the residue table
  bool lz_ok = false;
  FirLossyPlan lzp;
  uint32_t *d_lzcl = nullptr;
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

extern "C" {

int32_t acdsp_abi_version(void) { return ACDSP_ABI_VERSION; }
const char *acdsp_last_error(void) { return g_err.c_str(); }
int32_t acdsp_elem_bytes(int32_t W) { return elem_bytes(W); }

int32_t acdsp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { return 0; }
  int k = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) { k++; }
  }
  return k;
}

int32_t acdsp_dev_alloc(int32_t device, uint64_t bytes, void **d_ptr) {
  if (!d_ptr) { return fail(ACDSP_EINVAL, "null output pointer"); }
  int rc = check_device(device);
  if (rc) { return rc; }
  HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
  return ACDSP_OK;
}
int32_t acdsp_dev_free(int32_t device, void *d_ptr) {
  int rc = check_device(device);
  if (rc) { return rc; }
  HIP_TRY(hipFree(d_ptr));
  return ACDSP_OK;
}
int32_t acdsp_copy_h2d(int32_t device, void *d_dst, const void *h_src, uint64_t bytes) {
  int rc = check_device(device);
  if (rc) { return rc; }
  HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return ACDSP_OK;
}
int32_t acdsp_copy_d2h(int32_t device, void *h_dst, const void *d_src, uint64_t bytes) {
  int rc = check_device(device);
  if (rc) { return rc; }
  HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return ACDSP_OK;
}
int32_t acdsp_sync(int32_t device, void *stream) {
  int rc = check_device(device);
  if (rc) { return rc; }
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// stimulus generator
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void fill_stimulus_kernel(void *d, int eb, int64_t n, int64_t stride, uint64_t seed, int bits, uint64_t ch0,
                                     uint64_t t0) {
  const int64_t ch = blockIdx.y;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    uint64_t idx = ((ch0 + (uint64_t)ch) << 32) | ((t0 + (uint64_t)t) & 0xffffffffull);
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    store_raw(d, ch * stride + t, eb, ((int64_t)z) >> (64 - bits));
  }
}
}  // namespace

extern "C" int32_t acdsp_fill_stimulus(int32_t device, void *d_ptr, int32_t eb, int64_t n_ch, int64_t n, int64_t stride,
                                       uint64_t seed, int32_t bits, uint64_t ch0, uint64_t t0, void *stream) {
  if (!d_ptr || (eb != 2 && eb != 4 && eb != 8) || n_ch < 1 || n < 0 || stride < n || bits < 1 || bits > 8 * eb) {
    return fail(ACDSP_EINVAL, "fill_stimulus: bad arguments");
  }
  int rc = check_device(device);
  if (rc) { return rc; }
  if (n == 0) { return ACDSP_OK; }
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 4096) { gx = 4096; }
  for (int64_t c0 = 0; c0 < n_ch; c0 += 65535) {
    int64_t nc = n_ch - c0 < 65535 ? n_ch - c0 : 65535;
    hipLaunchKernelGGL(fill_stimulus_kernel, dim3(gx, (unsigned)nc), dim3(256), 0, (hipStream_t)stream,
                       (char *)d_ptr + c0 * stride * eb, eb, n, stride, seed, bits, ch0 + (uint64_t)c0, t0);
  }
  HIP_TRY(hipGetLastError());
  return ACDSP_OK;
}

namespace acdsp {
int set_error(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }
}  // namespace acdsp

// ---------------------------------------------------------------------------------------------
// diagnostics (bench.py: roofline.copy_GBps, roofline.envelope_ms)
// ---------------------------------------------------------------------------------------------
namespace {
template <typename F>
int time_launches(F launch, int warmup, int reps, hipStream_t s, float *ms_avg) {
  if (!ms_avg || reps < 1 || warmup < 0) { return fail(ACDSP_EINVAL, "diag: bad repetition counts"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipError_t e = hipSuccess;
  for (int i = 0; i < warmup && e == hipSuccess; i++) { e = launch(); }
  if (e == hipSuccess) { e = hipEventRecord(e0, s); }
  for (int i = 0; i < reps && e == hipSuccess; i++) { e = launch(); }
  if (e == hipSuccess) { e = hipEventRecord(e1, s); }
  if (e == hipSuccess) { e = hipEventSynchronize(e1); }
  float ms = 0;
  if (e == hipSuccess) { e = hipEventElapsedTime(&ms, e0, e1); }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "diag launch failed: %s", hipGetErrorString(e)); }
  *ms_avg = ms / reps;
  return ACDSP_OK;
}
}  // namespace

extern "C" int32_t acdsp_diag_copy_ms(int32_t device, const void *d_src, void *d_dst, uint64_t bytes, int32_t warmup, int32_t reps,
                                      void *stream, float *ms_avg) {
  if (!d_src || !d_dst || bytes < 16 || bytes % 16 || ((uintptr_t)d_src | (uintptr_t)d_dst) % 16) { return fail(ACDSP_EINVAL, "diag_copy: 16-byte aligned buffers of a multiple of 16 bytes"); }
  int rc = check_device(device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  return time_launches([&] { return launch_diag_copy(d_src, d_dst, (int64_t)bytes, s); }, warmup, reps, s, ms_avg);
}

extern "C" int32_t acdsp_diag_fir_envelope_ms(int32_t device, const int64_t *coeffs, int32_t n_taps, int32_t mfma_per_step, int32_t mfma_hi_per_step,
                                              const void *d_x, void *d_y, uint64_t bytes, int32_t warmup, int32_t reps, void *stream, float *ms_avg) {
  if (!d_x || !d_y || bytes < 16 || bytes % 16 || ((uintptr_t)d_x | (uintptr_t)d_y) % 16) { return fail(ACDSP_EINVAL, "diag_fir_envelope: 16-byte aligned buffers of a multiple of 16 bytes"); }
  if (mfma_per_step < 0 || mfma_hi_per_step < 0 || mfma_hi_per_step > mfma_per_step || (mfma_per_step > 0 && (!coeffs || n_taps < 1 || n_taps > 1025))) {
    return fail(ACDSP_EINVAL, "diag_fir_envelope: bad coefficient set or MFMA counts");
  }
  if (!diag_envelope_compiled(mfma_per_step, mfma_hi_per_step)) {
    return fail(ACDSP_EUNSUPPORTED, "diag_fir_envelope: (%d, %d) MFMAs per step is not a compiled count", mfma_per_step, mfma_hi_per_step);
  }
  int rc = check_device(device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  // A operands with the statistics of the product's: Toeplitz fragments of the caller's set (fir_mfma_build_fragments: [2 planes][nb][64][4]
  // dwords), four blocks of the low-byte plane spread over the taps and two non-zero blocks of the high-byte plane (the centre of the band)
  std::vector<uint32_t> six((size_t)6 * 64 * 4, 0u);
  if (mfma_per_step > 0) {
    for (int i = 0; i < n_taps; i++) {
      if (coeffs[i] < -32768 || coeffs[i] > 32767) { return fail(ACDSP_EINVAL, "diag_fir_envelope: coefficient %d is not a 16-bit word", i); }
    }
    FirMfmaPlan plan;
    const int nb = fir_mfma_plan_blocks(n_taps);
    std::vector<uint32_t> frag((size_t)2 * nb * 64 * 4);
    if (!fir_mfma_build_fragments(coeffs, n_taps, &plan, frag.data()) || plan.nb != nb) { return fail(ACDSP_EUNSUPPORTED, "diag_fir_envelope: set not splittable into two signed bytes"); }
    auto take = [&](int slot, int plane, int b) { memcpy(&six[(size_t)slot * 256], &frag[((size_t)plane * nb + b) * 256], 256 * sizeof(uint32_t)); };
    const int lo_pick[4] = {nb / 8 + (nb > 8 ? 1 : 0), (3 * nb) / 8, nb / 2, (3 * nb) / 4};
    for (int i = 0; i < 4; i++) { take(i, 1, lo_pick[i] < nb ? lo_pick[i] : nb - 1); }
    int h0 = -1, h1 = -1;    // the two non-zero high-plane blocks nearest the centre
    for (int d = 0; d < nb && h1 < 0; d++) {
      const int cand[2] = {nb / 2 - d, nb / 2 + d + 1};
      for (int k = 0; k < 2 && h1 < 0; k++) {
        const int b = cand[k];
        if (b >= 0 && b < nb && ((plan.hi_mask >> b) & 1)) { if (h0 < 0) { h0 = b; } else if (b != h0) { h1 = b; } }
      }
    }
    if (h0 < 0) { h0 = nb / 2; }
    if (h1 < 0) { h1 = h0; }
    take(4, 0, h0); take(5, 0, h1);
  }
  uint32_t *d_frag = nullptr;
  HIP_TRY(hipMalloc((void **)&d_frag, six.size() * sizeof(uint32_t)));
  hipError_t ce = hipMemcpy(d_frag, six.data(), six.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (ce != hipSuccess) { (void)hipFree(d_frag); return fail(ACDSP_EHIP, "diag_fir_envelope: upload failed: %s", hipGetErrorString(ce)); }
  int out = time_launches([&] { return launch_diag_envelope(d_frag, d_x, d_y, (int64_t)bytes, mfma_per_step, mfma_hi_per_step, s); }, warmup, reps, s, ms_avg);
  (void)hipFree(d_frag);
  return out;
}

// ---------------------------------------------------------------------------------------------
// FIR
// ---------------------------------------------------------------------------------------------
namespace {

// Effective direct-form coefficients of the folded architectures (lossless paths only):
// FOLD_EVEN uses c[0..N/2-1] on both halves (ac_fir_const_coeffs.h:248-251), FOLD_ODD uses
// c[0..mid] with the centre tap alone (:265-273).  Taps the reference never reads become 0.
// ftype: kernel-side value (internal_ftype): the anti-symmetric folds of ac_fir_reg_share negate the mirrored half.
std::vector<int64_t> effective_coeffs(const int64_t *c, int N, int ftype) {
  std::vector<int64_t> e(N, 0);
  if (ftype == ACDSP_FOLD_EVEN || ftype == kRsFoldEven || ftype == kRsFoldEvenAnti) {
    const int64_t sg = ftype == kRsFoldEvenAnti ? -1 : 1;
    for (int i = 0; i < N / 2; i++) { e[i] += c[i]; e[N - 1 - i] += sg * c[i]; }
  } else if (ftype == ACDSP_FOLD_ODD || ftype == kRsFoldOdd || ftype == kRsFoldOddAnti) {
    const int64_t sg = ftype == kRsFoldOddAnti ? -1 : 1;
    int mid = (N - 1) / 2;
    for (int i = 0; i < mid; i++) { e[i] += c[i]; e[N - 1 - i] += sg * c[i]; }
    e[mid] += c[mid];
  } else {
    for (int i = 0; i < N; i++) { e[i] = c[i]; }
  }
  return e;
}

// Kernel-side tap-order code of a (class, FTYPE) pair; -1 where the reference class has no branch for the FTYPE.
int internal_ftype(int kind, int ftype) {
  if (kind != ACDSP_FIR_REG_SHARE) { return (ftype >= ACDSP_SHIFT_REG && ftype <= ACDSP_TRANSPOSED) ? ftype : -1; }
  switch (ftype) {   // ac_fir_reg_share.h:288-306
    case ACDSP_SHIFT_REG: return kRsShiftReg;
    case ACDSP_FOLD_EVEN: return kRsFoldEven;
    case ACDSP_FOLD_EVEN_ANTI: return kRsFoldEvenAnti;
    case ACDSP_FOLD_ODD: return kRsFoldOdd;
    case ACDSP_FOLD_ODD_ANTI: return kRsFoldOddAnti;
    default: return -1;
  }
}
inline bool is_fold_odd(int ift) { return ift == ACDSP_FOLD_ODD || ift == kRsFoldOdd || ift == kRsFoldOddAnti; }

int fir_validate(const acdsp_fir_desc_t &d) {
  if (d.kind < ACDSP_FIR_CONST || d.kind > ACDSP_FIR_REG_SHARE) { return fail(ACDSP_EINVAL, "bad FIR class %d", d.kind); }
  if (d.ftype < 0 || d.ftype > ACDSP_FOLD_ODD_ANTI) { return fail(ACDSP_EINVAL, "bad ftype %d", d.ftype); }
  if (internal_ftype(d.kind, d.ftype) < 0) {
    return fail(ACDSP_EUNSUPPORTED, d.kind == ACDSP_FIR_REG_SHARE
                    ? "ac_fir_reg_share::run() has no branch for this FTYPE (output would be an unassigned value)"
                    : "FOLD_*_ANTI: the reference run() has no branch for these (output is an unassigned value)");
  }
  if (d.n_taps < 1 || d.n_taps > 2048) { return fail(ACDSP_EUNSUPPORTED, "n_taps=%d outside 1..2048", d.n_taps); }
  if (d.n_channels < 1) { return fail(ACDSP_EINVAL, "n_channels=%d must be positive", d.n_channels); }
  if (d.n_channels > 65535) { return fail(ACDSP_EUNSUPPORTED, "n_channels=%d outside 1..65535", d.n_channels); }
  int rc;
  if ((rc = check_fmt(d.in, "IN_TYPE")) || (rc = check_fmt(d.coeff, "COEFF_TYPE")) || (rc = check_fmt(d.acc, "ACC_TYPE", 128)) ||
      (rc = check_fmt(d.out, "OUT_TYPE", 128))) {
    return rc;
  }
  // exact intermediates must hold: product, aligned sum -- 128 bits on the 64-bit paths, 256 bits on the wide path (wide.hip)
  const bool wide = d.acc.W > 64 || d.out.W > 64;
  const int limit = wide ? 250 : 125;
  int fi = d.in.W - d.in.I, fc = d.coeff.W - d.coeff.I, fa = d.acc.W - d.acc.I, fo = d.out.W - d.out.I;
  int wp = d.in.W + d.coeff.W + 2, fp = fi + fc;
  if (is_fold_odd(internal_ftype(d.kind, d.ftype))) { wp = d.acc.W + d.coeff.W + 1; fp = fa + fc; }
  int f = fp > fa ? fp : fa;
  if (wp + (f - fp) > limit || d.acc.W + (f - fa) > limit || (wide && d.acc.W + (fo > fa ? fo - fa : 0) > limit)) {
    return fail(ACDSP_EUNSUPPORTED, "type combination needs more than %d-bit intermediates", wide ? 256 : 128);
  }
  // FOLD_ODD: the ACC_TYPE `fold` of the pre-add (ac_fir_const_coeffs.h:262-269) is formed from an (in.W + 1)-bit sum shifted to ACC's fraction
  if (wide && is_fold_odd(internal_ftype(d.kind, d.ftype)) && d.in.W + 1 + (fa > fi ? fa - fi : 0) > limit) {
    return fail(ACDSP_EUNSUPPORTED, "type combination needs more than 256-bit intermediates");
  }
  if (wide && d.kind == ACDSP_FIR_REG_SHARE) { return fail(ACDSP_EUNSUPPORTED, "ac_fir_reg_share: ACC / OUT wider than 64 bits not supported"); }
  return ACDSP_OK;
}

}  // namespace

extern "C" {

int32_t acdsp_fir_create(const acdsp_fir_desc_t *desc, acdsp_fir_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  int rc = fir_validate(*desc);
  if (rc) { return rc; }
  if ((rc = check_device(desc->device))) { return rc; }
  acdsp_fir *h = new acdsp_fir();
  h->d = *desc;
  h->in_eb = elem_bytes(desc->in.W);
  h->out_eb = elem_bytes(desc->out.W);
  h->wide = desc->acc.W > 64 || desc->out.W > 64;
  h->rt_eb = h->wide ? 16 : 8;
  h->hl = round_up(desc->n_taps + 15, 32);  // >= n_taps-1 for every kernel, >= n_taps+14 for the 16-aligned windows of fir_gen
  // a plan of NB K-blocks reaches 32 (NB - 1) samples back; a padded plan (fir_mfma_plan_blocks: even counts of 10 .. 32 blocks) one block
  // further than the tap count asks for.  Sized from the padded count whatever the ACDSP_NO_MID knob says, and from NB - 1, not NB:
  // round 3 grew the history of UNpadded plans too (240 taps: 288 instead of 256) and let the knob change the state geometry.
  if (h->hl < 32 * (fir_mfma_plan_blocks_padded(desc->n_taps) - 1)) { h->hl = 32 * (fir_mfma_plan_blocks_padded(desc->n_taps) - 1); }
  // reg_trans[] carries partial sums computed with the coefficients of their own time; only
  // the const-coefficient class may trade it for an input history.
  h->use_rt = desc->ftype == ACDSP_TRANSPOSED && desc->kind != ACDSP_FIR_CONST;
  const int fi = desc->in.W - desc->in.I, fc = desc->coeff.W - desc->coeff.I, fa = desc->acc.W - desc->acc.I;
  // exact-dot-product class: the kernels compute `sum << (fa - fi - fc)` in 64 bits, so the shift must be 0..63 (formats
  // with I outside [0, W] can ask for more: those stay on the per-tap path)
  static const bool no_hybrid = getenv("ACDSP_NO_RT_HYBRID") != nullptr;   // A/B knob: reg_trans on the exact-order kernel for every sample
  h->rt_hybrid = h->use_rt && !no_hybrid && !(desc->flags & ACDSP_FLAG_FORCE_GENERIC) && (desc->in.S || desc->in.W <= 15) && desc->acc.O == ACDSP_WRAP && fa >= fi + fc && fa - fi - fc < 64 && !h->wide;
  h->rt_since = desc->n_taps - 1;   // an all-zero state carries no coefficients
  bool lossless = desc->acc.O == ACDSP_WRAP && fa >= fi + fc && fa - fi - fc < 64 && (!h->use_rt || h->rt_hybrid) && !h->wide;
  const int ift = internal_ftype(desc->kind, desc->ftype);
  if (is_fold_odd(ift)) {
    // the ACC_TYPE `fold` must also keep every fraction bit of the pre-add (fc < 0 would let fa >= fi + fc pass with fa < fi)
    lossless = lossless && fa >= fi;
    // the ACC_TYPE `fold` variable must hold x[i] +/- x[N-1-i] without wrapping (a difference needs a signed type)
    int need_i = desc->in.I + 1 + ((desc->acc.S && !desc->in.S) ? 1 : 0);
    lossless = lossless && desc->acc.I >= need_i && (desc->acc.S || (!desc->in.S && ift != kRsFoldOddAnti));
  }
  h->lossless = lossless;
  h->coeffs_set = false;
  h->path = ACDSP_PATH_GENERIC;
  const size_t hist_bytes = (size_t)desc->n_channels * h->hl * h->in_eb;
  const size_t n_sets = desc->coeffs_per_channel ? (size_t)desc->n_channels : 1;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc(&h->d_hist[i], hist_bytes);
    if (e == hipSuccess) { e = hipMemset(h->d_hist[i], 0, hist_bytes); }
    if (e == hipSuccess && h->use_rt) {
      size_t rb = (size_t)desc->n_channels * desc->n_taps * h->rt_eb;
      e = hipMalloc((void **)&h->d_rt[i], rb);
      if (e == hipSuccess) { e = hipMemset(h->d_rt[i], 0, rb); }
    }
  }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_coeffs, n_sets * desc->n_taps * sizeof(int64_t)); }
  {
    const int nbk = fir_mfma_plan_blocks(desc->n_taps);
    if (e == hipSuccess) { e = hipMalloc((void **)&h->d_frag, n_sets * sizeof(uint32_t) * 2 * (size_t)(nbk > 0 ? nbk : 1) * 64 * 4); }
  }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_corr, n_sets * sizeof(int64_t)); }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_gfrag, 3 * 8 * 64 * 4 * sizeof(uint32_t)); }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_lzcl, kLossyTabWords * sizeof(uint32_t)); }
  if (e != hipSuccess || h->tm.init() != ACDSP_OK) {
    acdsp_fir_destroy(h);
    return fail(ACDSP_EHIP, "FIR state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_fir_destroy(acdsp_fir_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  for (int i = 0; i < 2; i++) {
    if (h->d_hist[i]) { (void)hipFree(h->d_hist[i]); }
    if (h->d_rt[i]) { (void)hipFree(h->d_rt[i]); }
  }
  if (h->d_coeffs) { (void)hipFree(h->d_coeffs); }
  if (h->d_frag) { (void)hipFree(h->d_frag); }
  if (h->d_corr) { (void)hipFree(h->d_corr); }
  if (h->d_gfrag) { (void)hipFree(h->d_gfrag); }
  if (h->d_lzcl) { (void)hipFree(h->d_lzcl); }
  h->tm.destroy();
  h->st.destroy();
  h->st_u.destroy();
  delete h;
  return ACDSP_OK;
}

// rt_hybrid: reg_trans[] of every channel from the input history and the coefficients in d_coeffs (the reference's recurrence unrolled
// in time: fir_rt_update_kernel with the history as a call of hl >= n_taps samples, so that no older partial sum enters).  Synchronous.
static int32_t fir_rt_from_hist(acdsp_fir *h) {
  const acdsp_fir_desc_t &d = h->d;
  FirParams k;
  memset(&k, 0, sizeof k);
  k.n_taps = d.n_taps; k.ftype = internal_ftype(d.kind, d.ftype); k.n_ch = d.n_channels; k.coeffs_per_channel = d.coeffs_per_channel;
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = make_dfmt(d.acc); k.out = make_dfmt(d.out);
  k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.hl = h->hl; k.use_rt = 1;
  k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
  k.x = h->d_hist[h->cur]; k.in_stride = h->hl; k.n = h->hl;
  k.coeffs = h->d_coeffs; k.rt = h->d_rt[h->cur_rt];
  const hipError_t e = launch_fir_rt_update(k, h->d_rt[h->cur_rt ^ 1], nullptr);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "reg_trans rebuild failed: %s", hipGetErrorString(e)); }
  HIP_TRY(hipDeviceSynchronize());
  h->cur_rt ^= 1;
  h->rt_valid = true;
  return ACDSP_OK;
}

int32_t acdsp_fir_set_coeffs(acdsp_fir_t h, const int64_t *coeffs) {
  if (!h || !coeffs) { return fail(ACDSP_EINVAL, "null argument"); }
  const acdsp_fir_desc_t &d = h->d;
  if (d.kind == ACDSP_FIR_CONST && h->coeffs_set && d.ftype == ACDSP_TRANSPOSED) {
    return fail(ACDSP_ESTATE, "const-coefficient TRANSPOSED filter: coefficients are bound once");
  }
  int rc = check_device(d.device);
  if (rc) { return rc; }
  const size_t n_sets = d.coeffs_per_channel ? (size_t)d.n_channels : 1;
  const acdsp::DFmt cf = make_dfmt(d.coeff);
  for (size_t i = 0; i < n_sets * d.n_taps; i++) {
    if (coeffs[i] < cf.lo || coeffs[i] > cf.hi) {
      return fail(ACDSP_EINVAL, "coefficient %zu = %lld is not a COEFF_TYPE raw word", i, (long long)coeffs[i]);
    }
  }
  // the same set again (ac_fir_prog_coeffs hands its coefficients to every one-sample call): nothing changes, no state event
  if (h->coeffs_set && h->h_coeffs.size() == n_sets * d.n_taps && memcmp(h->h_coeffs.data(), coeffs, n_sets * d.n_taps * sizeof(int64_t)) == 0) { return ACDSP_OK; }
  // Kernels of earlier run() calls may still be reading d_coeffs / d_frag.
  HIP_TRY(hipDeviceSynchronize());
  if (h->rt_hybrid && h->coeffs_set) {
    // a change mid-stream: the partial sums of the next n_taps - 1 outputs keep the OLD coefficients' products (ac_fir_load_coeffs.h:265-278)
    if (!h->rt_valid && (rc = fir_rt_from_hist(h))) { return rc; }
    h->rt_since = 0;
  }
  // from here on the device side changes: a failure below must not leave the OLD set looking current (the early return above compares
  // against h_coeffs), so the handle is without a set until the call succeeds
  h->coeffs_set = false;
  h->h_coeffs.clear();
  HIP_TRY(hipMemcpy(h->d_coeffs, coeffs, n_sets * d.n_taps * sizeof(int64_t), hipMemcpyHostToDevice));
  h->mfma_ok = false;
  h->in_flip = false;
  static const bool no_flip = getenv("ACDSP_NO_UNSIGNED16") != nullptr;   // A/B knob: unsigned 16-bit samples stay on the exact-sum VALU kernel
  const bool flip = !d.in.S && d.in.W == 16 && !no_flip && !h->use_rt;
  const bool i16_in = d.in.W <= 15 || (d.in.W == 16 && (d.in.S || flip));
  const bool i16_cf = d.coeff.S ? d.coeff.W <= 16 : d.coeff.W <= 15;
  if (h->lossless && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && i16_in && i16_cf && h->in_eb == 2 &&
      fir_mfma_plan_blocks(d.n_taps) <= fir_mfma_max_blocks()) {
    const int nb = fir_mfma_plan_blocks(d.n_taps);
    const size_t per_set = (size_t)2 * nb * 64 * 4;
    std::vector<uint32_t> frag(n_sets * per_set, 0u);
    std::vector<int64_t> corr(n_sets, 0);
    FirMfmaPlan worst;
    memset(&worst, 0, sizeof worst);
    bool ok = true;
    for (size_t st = 0; st < n_sets && ok; st++) {
      std::vector<int64_t> eff = effective_coeffs(coeffs + st * d.n_taps, d.n_taps, internal_ftype(d.kind, d.ftype));
      FirMfmaPlan pl;
      ok = fir_mfma_build_fragments(eff.data(), d.n_taps, &pl, frag.data() + st * per_set);
      if (!ok) { break; }
      if (flip) {   // + 32768 * sum(c): the samples go through the kernel as x - 32768
        int64_t sc = 0;
        for (int64_t v : eff) { sc += v; }
        pl.corr += 32768 * sc;
        // the kernel sees signed 16-bit samples (|x| <= 2^15) but the recombined sum is the UNSIGNED dot product, |y| <= 65535 * sum|c|:
        // the no-wrap proof of the fast epilogues (fir_mfma_epilogue_class: sum_abs * x_max against ACC's range) must use that bound
        pl.sum_abs *= 2;
      }
      corr[st] = pl.corr;
      worst.nb = pl.nb;
      worst.hi_mask |= pl.hi_mask; worst.lo_mask |= pl.lo_mask;
      if (pl.sum_abs > worst.sum_abs) { worst.sum_abs = pl.sum_abs; }
      if (pl.sum_abs_hi > worst.sum_abs_hi) { worst.sum_abs_hi = pl.sum_abs_hi; }
      if (pl.sum_abs_lo > worst.sum_abs_lo) { worst.sum_abs_lo = pl.sum_abs_lo; }
      const int64_t ca = pl.corr < 0 ? -pl.corr : pl.corr, wa = worst.corr < 0 ? -worst.corr : worst.corr;
      if (st == 0 || ca > wa) { worst.corr = pl.corr; }
    }
    if (ok && d.coeffs_per_channel && worst.nb > fir_mfma_max_reg_blocks()) {
      // a set per channel needs the register-resident kernels: beyond 9 K-blocks only band-limited sets with the fast int16 epilogue
      FirParams k;
      memset(&k, 0, sizeof k);
      k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = make_dfmt(d.acc); k.out = make_dfmt(d.out);
      k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
      ok = fir_mfma_register_resident(k, worst);
    }
    if (ok) {
      HIP_TRY(hipMemcpy(h->d_frag, frag.data(), frag.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(h->d_corr, corr.data(), corr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      h->plan = worst;
      h->mfma_ok = true;
      h->in_flip = flip;
    }
  }
  // wide inputs (more than 16 bits) / other misses of the int16 kernel: generalised multi-plane MFMA kernel
  h->gen_ok = false;
  static const bool no_gen = getenv("ACDSP_NO_GEN") != nullptr;
  if (!h->mfma_ok && h->lossless && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && !d.coeffs_per_channel && !no_gen &&
      (d.in.W + (d.in.S ? 0 : 1) + 7) / 8 <= h->in_eb) {
    std::vector<int64_t> eff = effective_coeffs(coeffs, d.n_taps, internal_ftype(d.kind, d.ftype));
    std::vector<uint32_t> gfrag;
    if (fir_gen_plan(eff.data(), d.n_taps, 1, 0, &h->gplan, &gfrag)) {
      HIP_TRY(hipMemcpy(h->d_gfrag, gfrag.data(), gfrag.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      h->gen_ok = true;
    }
  }
  // Class B (lossy accumulator, AC_TRN / AC_RND into AC_WRAP) on the matrix cores: sum_k Q(p_k) = (sum_k p_k + N h - sum_k ((p_k + h) mod 2^s)) >> s.
  // The exact sum is class A on the effective taps; the residues need the low s bits of every (folded) sample and coefficient (fir_gen.hip, LZ).
  // FOLD_ODD holds the pre-add in an ACC_TYPE variable (ac_fir_prog_coeffs.h:213-227): exact when ACC keeps the sample's fraction bits and
  // cannot wrap on the sum of two samples, and the product c * fold then drops the same s bits as an unfolded tap.
  h->lz_ok = false;
  FirParams kq;
  memset(&kq, 0, sizeof kq);
  kq.n_taps = d.n_taps; kq.ftype = internal_ftype(d.kind, d.ftype); kq.n_ch = d.n_channels; kq.coeffs_per_channel = d.coeffs_per_channel;
  kq.in = make_dfmt(d.in); kq.cf = make_dfmt(d.coeff);
  if (!h->wide) { kq.acc = make_dfmt(d.acc); kq.out = make_dfmt(d.out); }
  kq.in_eb = h->in_eb; kq.out_eb = h->out_eb; kq.hl = h->hl; kq.use_rt = (h->use_rt && !h->rt_hybrid) ? 1 : 0;
  kq.lossless_shift = kq.acc.F - kq.in.F - kq.cf.F;
  static const bool no_lz = getenv("ACDSP_NO_MFMA_LOSSY") != nullptr;        // A/B knob: class B stays on the VALU kernels
  static const bool lz_first = getenv("ACDSP_MFMA_LOSSY_FIRST") != nullptr;  // A/B knob: ... also takes the 16-bit types fir_lossy_kernel serves
  {
    const int ift = internal_ftype(d.kind, d.ftype);
    const int fi = kq.in.F, fc = kq.cf.F, fa = kq.acc.F, sbits = fi + fc - fa;
    const bool fold_odd = is_fold_odd(ift), fold_even = ift == ACDSP_FOLD_EVEN || ift == kRsFoldEven || ift == kRsFoldEvenAnti;
    const bool anti = ift == kRsFoldEvenAnti || ift == kRsFoldOddAnti;
    bool ok = !no_lz && !no_gen && !h->wide && !h->lossless && !h->use_rt && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && !d.coeffs_per_channel &&
              d.acc.O == ACDSP_WRAP && (d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && d.acc.S && d.acc.W <= 64 && sbits >= 1 && sbits <= 8 &&
              (h->in_eb == 2 || h->in_eb == 4) && (d.in.W + (d.in.S ? 0 : 1) + 7) / 8 <= h->in_eb &&
              (lz_first || !fir_lossy_fast_ok(kq));
    if (ok && fold_odd) {
      const int need_i = d.in.I + 1 + ((d.acc.S && !d.in.S) ? 1 : 0);
      ok = fa >= fi && d.acc.I >= need_i;
    }
    const int n_pair = fold_odd ? (d.n_taps - 1) / 2 : (fold_even ? d.n_taps / 2 : 0);
    const int n_single = fold_odd ? 1 : (fold_even ? 0 : d.n_taps);
    ok = ok && n_pair + n_single >= 1;
    if (ok) {
      std::vector<int64_t> eff = effective_coeffs(coeffs, d.n_taps, ift);
      std::vector<uint32_t> gfrag, tab;
      ok = fir_gen_plan(eff.data(), d.n_taps, 1, 0, &h->gplan, &gfrag);
      // the exact sum must not leave int64 (the shift by s follows it), unless ACC_TYPE only keeps bits that survive a wrap of 2^64
      const int xb = d.in.W - (d.in.S ? 1 : 0);
      const bool bounded = ok && h->gplan.sum_abs_h < (int64_t(1) << 61) && xb <= 61 && h->gplan.sum_abs_h <= ((int64_t(1) << 61) >> xb);
      ok = ok && (d.acc.W + sbits <= 64 || bounded);
      int acc_bits = d.acc.W;       // |acc| <= sum|c| 2^xb / 2^s + 1 when the sum is bounded: lets a 64-bit ACC_TYPE round into OUT_TYPE in int64
      if (bounded) {
        int sb = 0;
        while (sb < 62 && (int64_t(1) << sb) <= h->gplan.sum_abs_h) { sb++; }
        const int vb = sb + xb - sbits + 2;
        if (vb < acc_bits) { acc_bits = vb < 2 ? 2 : vb; }
      }
      ok = ok && fir_gen_lossy_shape_ok(kq, h->gplan, acc_bits);
      const int single0 = fold_odd ? (d.n_taps - 1) / 2 : 0;
      ok = ok && fir_gen_lossy_table(h->gplan, coeffs, d.n_taps, n_pair, n_single, single0, anti ? 1 : 0, sbits, d.acc.Q == ACDSP_RND, &h->lzp, &tab);
      if (ok) {
        h->lzp.d_tab = h->d_lzcl; h->lzp.acc_bits = acc_bits;
        HIP_TRY(hipMemcpy(h->d_gfrag, gfrag.data(), gfrag.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_lzcl, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        h->lz_ok = true;
      }
    }
  }
  h->path = h->wide ? ACDSP_PATH_WIDE
            : h->mfma_ok ? ACDSP_PATH_MFMA_I8
            : h->gen_ok ? ACDSP_PATH_MFMA_GEN
            : h->lz_ok ? ACDSP_PATH_MFMA_LOSSY
                        : ((h->lossless && !(d.flags & ACDSP_FLAG_FORCE_GENERIC)) ? ACDSP_PATH_LOSSLESS64 : ACDSP_PATH_GENERIC);
  {
    static const bool no_lossy = getenv("ACDSP_NO_LOSSY_FAST") != nullptr;
    h->kclass = h->path;
    if (h->path == ACDSP_PATH_GENERIC && !no_lossy && !(d.flags & ACDSP_FLAG_FORCE_GENERIC)) {
      h->kclass = fir_lossy_fast_ok(kq) ? ACDSP_KCLASS_LOSSY16 : (fir_satacc_fast_ok(kq) ? ACDSP_KCLASS_SATACC16 : ACDSP_PATH_GENERIC);
    }
  }
  h->coeffs_set = true;
  h->h_coeffs.assign(coeffs, coeffs + n_sets * d.n_taps);
  return ACDSP_OK;
}

int32_t acdsp_fir_clone(acdsp_fir_t h, acdsp_fir_t *out) {
  if (!h || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  acdsp_fir_t c = nullptr;
  int rc = acdsp_fir_create(&h->d, &c);
  if (rc) { return rc; }
  if (h->coeffs_set && (rc = acdsp_fir_set_coeffs(c, h->h_coeffs.data()))) { acdsp_fir_destroy(c); return rc; }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(c->d_hist[0], h->d_hist[h->cur], (size_t)h->d.n_channels * h->hl * h->in_eb, hipMemcpyDeviceToDevice));
  if (h->use_rt) {
    HIP_TRY(hipMemcpy(c->d_rt[0], h->d_rt[h->rt_hybrid ? h->cur_rt : h->cur], (size_t)h->d.n_channels * h->d.n_taps * h->rt_eb, hipMemcpyDeviceToDevice));
  }
  c->cur = 0; c->cur_rt = 0; c->rt_valid = h->rt_valid; c->rt_since = h->rt_since;
  *out = c;
  return ACDSP_OK;
}

int32_t acdsp_fir_path(acdsp_fir_t h) { return h ? h->path : -1; }
int32_t acdsp_fir_kernel_class(acdsp_fir_t h) { return (h && h->coeffs_set) ? h->kclass : -1; }

int32_t acdsp_fir_run(acdsp_fir_t h, const void *d_in, int64_t in_stride, int64_t n, void *d_out, int64_t out_stride,
                      void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n < 0 || (n > 0 && (!d_in || !d_out || in_stride < n || out_stride < n))) {
    return fail(ACDSP_EINVAL, "fir_run: bad buffer arguments");
  }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "fir_run before acdsp_fir_set_coeffs"); }
  if (n == 0) { return ACDSP_OK; }
  const acdsp_fir_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  FirParams k;
  k.hist_next = nullptr; k.t_begin = 0;
  k.n_taps = d.n_taps; k.ftype = internal_ftype(d.kind, d.ftype); k.n_ch = d.n_channels; k.coeffs_per_channel = d.coeffs_per_channel;
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff);
  if (h->wide) { memset(&k.acc, 0, sizeof k.acc); memset(&k.out, 0, sizeof k.out); k.acc.F = d.acc.W - d.acc.I; }
  else { k.acc = make_dfmt(d.acc); k.out = make_dfmt(d.out); }
  const bool hyb = h->rt_hybrid;
  k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.hl = h->hl; k.use_rt = (h->use_rt && !hyb) ? 1 : 0;
  k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
  k.in_stride = in_stride; k.out_stride = out_stride; k.n = n;
  k.x = d_in; k.y = d_out;
  k.hist = h->d_hist[h->cur]; k.coeffs = h->d_coeffs; k.rt = h->d_rt[hyb ? h->cur_rt : h->cur];
  // rt_hybrid: the first m outputs still carry partial sums of the previous coefficient set
  int64_t m_rt = 0;
  if (hyb) {
    m_rt = (int64_t)d.n_taps - 1 - h->rt_since;
    m_rt = m_rt < 0 ? 0 : (m_rt > n ? n : m_rt);
    if (m_rt > 0 && stream_is_capturing(s)) {
      return fail(ACDSP_ESTATE, "fir_run under graph capture: a TRANSPOSED filter within n_taps - 1 samples of a coefficient change keeps host-side state; run %lld more samples before capturing", (long long)m_rt);
    }
  }

  int path = h->path;
  if (h->wide) {
    FirWideParams kw;
    kw.p = k; kw.acc = make_wfmt(d.acc); kw.out = make_wfmt(d.out); kw.rt = h->d_rt[h->cur];
    HIP_TRY(hipEventRecord(h->tm.start(), s));
    hipError_t ew = launch_fir_wide(kw, s);
    if (ew != hipSuccess) { return fail(ACDSP_EHIP, "wide FIR kernel launch failed: %s", hipGetErrorString(ew)); }
    HIP_TRY(hipEventRecord(h->tm.stop(), s));
    h->tm.commit();
    const int nxw = hist_next_index(h->cur, !h->use_rt && k.n >= k.hl);
    ew = h->use_rt ? launch_fir_wide_rt_update(kw, h->d_rt[nxw], s) : launch_fir_hist_update(k, h->d_hist[nxw], s);
    if (ew != hipSuccess) { return fail(ACDSP_EHIP, "wide FIR state kernel launch failed: %s", hipGetErrorString(ew)); }
    h->cur = nxw;
    return ACDSP_OK;
  }
  FirParams kraw = k;   // the state kernels always see the caller's samples
  const bool flipped = h->in_flip && path == ACDSP_PATH_MFMA_I8;
  if (flipped) {
    // unsigned 16-bit samples: a sign-flipped image of the call's rows and of the history (see acdsp_fir::in_flip)
    const int64_t si = (n + 15) / 16 * 16;
    const size_t ub_in = (size_t)d.n_channels * si * 2, ub_h = (size_t)d.n_channels * h->hl * 2;
    if ((ub_in > h->st_u.cap_in || ub_h > h->st_u.cap_out) && stream_is_capturing(s)) {
      return fail(ACDSP_ESTATE, "fir_run under graph capture: the staging image of unsigned 16-bit samples must grow (run one call of this length before capturing)");
    }
    if ((rc = h->st_u.ensure(ub_in, ub_h))) { return rc; }
    if (!h->small_call) { HIP_TRY(hipEventRecord(h->tm.start(), s)); }   // the flip is part of this path's cost: inside the timed region
    const hipError_t ef = launch_flip16(d_in, in_stride, n, h->st_u.d_in, si, h->d_hist[h->cur], h->st_u.d_out, h->hl, d.n_channels, s);
    if (ef != hipSuccess) { return fail(ACDSP_EHIP, "FIR sample staging kernel launch failed: %s", hipGetErrorString(ef)); }
    k.x = h->st_u.d_in; k.in_stride = si; k.hist = h->st_u.d_out;
    k.in.S = 1; k.in.lo = -32768; k.in.hi = 32767;
  }
  if (!flipped && (path == ACDSP_PATH_MFMA_I8 || path == ACDSP_PATH_MFMA_GEN || path == ACDSP_PATH_MFMA_LOSSY)) {
    // The matrix-core kernels read rows with 16-byte vector loads (fir_gen: in whole 16-sample slots).  gfx950 serves a vector
    // access at any ELEMENT-aligned address, so the int8 kernel takes unaligned rows as they are (round 3: a row stride of 2^20 + 3
    // samples costs +12 %, profiles/r3_unaligned.txt; the staging copy below -- hipMemcpy2DAsync of misaligned rows -- cost 6.4 ms
    // per 2 GB, 7 x the filter itself) as long as a row is readable up to the next multiple of 8 samples.  fir_gen still wants
    // whole aligned slots: rows that are not laid out that way are first copied, on the device, into an aligned staging image.
    static const bool aligned_only = getenv("ACDSP_ALIGNED_ONLY") != nullptr;   // A/B knob: the round-2 behaviour
    bool aligned = ((uintptr_t)d_in % 16 == 0) && ((in_stride * h->in_eb) % 16 == 0) &&
                   (path == ACDSP_PATH_MFMA_I8 || in_stride >= (n + 15) / 16 * 16);
    if (!aligned && !aligned_only && path == ACDSP_PATH_MFMA_I8 && in_stride >= (n + 7) / 8 * 8) { aligned = true; }
    if (!aligned) {
      const int64_t si = (n + 15) / 16 * 16;
      if ((rc = h->st.ensure((size_t)d.n_channels * si * h->in_eb, 0))) { return rc; }
      HIP_TRY(hipMemcpy2DAsync(h->st.d_in, (size_t)si * h->in_eb, d_in, (size_t)in_stride * h->in_eb, (size_t)n * h->in_eb,
                               (size_t)d.n_channels, hipMemcpyDeviceToDevice, s));
      k.x = h->st.d_in; k.in_stride = si;
    }
  }
  // Small calls (the drop-in run() of one channel; ac_fir_prog_coeffs is ONE sample per call, reference ac_fir_prog_coeffs.h:281)
  // are launch-bound: no timing events, and the exact-order kernels write the next history themselves -- one launch per call.
  const bool small = h->small_call;
  const bool fuse_hist = small && !flipped && (!h->use_rt || hyb) && (path == ACDSP_PATH_LOSSLESS64 || path == ACDSP_PATH_GENERIC ||
                                                (path == ACDSP_PATH_MFMA_I8 && fir_mfma_register_resident(k, h->plan)));   // single-wave workgroups
  const int nxt_fused = hist_next_index(h->cur, false);
  if (fuse_hist) { k.hist_next = h->d_hist[nxt_fused]; }
  if (!small && !flipped) { HIP_TRY(hipEventRecord(h->tm.start(), s)); }
  hipError_t e;
  if (path == ACDSP_PATH_MFMA_I8) { e = launch_fir_mfma(k, h->plan, d.coeffs_per_channel, h->d_frag, h->d_corr, s); }
  else if (path == ACDSP_PATH_MFMA_GEN) { e = launch_fir_gen(k, h->gplan, h->d_gfrag, 0, 0, 0, n, s); }
  else if (path == ACDSP_PATH_MFMA_LOSSY) {
    // complete chunks on the matrix cores, the ragged rest (and calls shorter than a chunk) on the exact-order kernel
    int64_t cov = 0;
    e = launch_fir_gen(k, h->gplan, h->d_gfrag, 0, 0, 0, n, s, &h->lzp, &cov);
    if (e == hipSuccess && cov < n) { FirParams kt = k; kt.t_begin = cov; e = launch_fir_generic(kt, s); }
  }
  else if (path == ACDSP_PATH_LOSSLESS64) { e = launch_fir_lossless64(k, s); }
  else {
    static const bool no_lossy = getenv("ACDSP_NO_LOSSY_FAST") != nullptr;   // A/B knob: the exact-order kernel for every per-tap class
    e = (!no_lossy && fir_lossy_fast_ok(k)) ? launch_fir_lossy(k, s) : ((!no_lossy && fir_satacc_fast_ok(k)) ? launch_fir_satacc(k, s) : launch_fir_generic(k, s));
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "FIR kernel launch failed: %s", hipGetErrorString(e)); }
  if (!small) {
    HIP_TRY(hipEventRecord(h->tm.stop(), s));
    h->tm.commit();
  }
  if (hyb) {
    if (m_rt > 0) {
      // exact-order pass over the call's first m samples, on reg_trans, behind the main kernel (it overwrites those outputs)
      FirParams kt = k;
      kt.use_rt = 1; kt.n = m_rt; kt.hist_next = nullptr;
      e = launch_fir_generic(kt, s);
      if (e == hipSuccess && m_rt == n) { e = launch_fir_rt_update(kt, h->d_rt[h->cur_rt ^ 1], s); }
      if (e != hipSuccess) { return fail(ACDSP_EHIP, "FIR reg_trans kernel launch failed: %s", hipGetErrorString(e)); }
      if (m_rt == n) { h->cur_rt ^= 1; h->rt_valid = true; } else { h->rt_valid = false; }   // past the transition reg_trans is rebuilt on demand
    } else {
      h->rt_valid = false;
    }
    h->rt_since = h->rt_since + n >= (int64_t)d.n_taps - 1 ? (int64_t)d.n_taps - 1 : h->rt_since + n;
  }
  if (fuse_hist) { h->cur = nxt_fused; return ACDSP_OK; }
  // state carry.  A call of at least hl samples takes the new history from its input alone: written in place behind the
  // main kernel (same stream), no buffer flip -- the handle's host-side state is then the same after every call, which is what
  // lets any schedule of such calls be captured into a HIP graph.  Shorter calls (and reg_trans, which reads its old value)
  // go into the other buffer, then flip.
  const int nxt = hist_next_index(h->cur, (!h->use_rt || hyb) && k.n >= k.hl);
  if (h->use_rt && !hyb) {
    e = launch_fir_rt_update(k, h->d_rt[nxt], s);
  } else {
    e = launch_fir_hist_update(flipped ? kraw : k, h->d_hist[nxt], s);
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "FIR state kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur = nxt;
  return ACDSP_OK;
}

int32_t acdsp_fir_run_host(acdsp_fir_t h, const void *h_in, int64_t n, void *h_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n < 0 || (n > 0 && (!h_in || !h_out))) { return fail(ACDSP_EINVAL, "fir_run_host: bad arguments"); }
  if (n == 0) { return ACDSP_OK; }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  const int64_t stride = (n + 15) / 16 * 16;  // rows 16-byte aligned and readable in whole 16-sample slots
  const size_t bin = (size_t)h->d.n_channels * stride * h->in_eb, bout = (size_t)h->d.n_channels * stride * h->out_eb;
  static const bool no_pin = getenv("ACDSP_NO_PINNED") != nullptr;   // A/B knob: always go through the device staging buffers
  if (bin <= Staging::kPinBytes && bout <= Staging::kPinBytes && !no_pin) {
    if ((rc = h->st.ensure_pinned())) { return rc; }
    for (int c = 0; c < h->d.n_channels; c++) {
      memcpy((char *)h->st.pin_in + (size_t)c * stride * h->in_eb, (const char *)h_in + (size_t)c * n * h->in_eb, (size_t)n * h->in_eb);
    }
    h->small_call = true;
    rc = acdsp_fir_run(h, h->st.pin_in, stride, n, h->st.pin_out, stride, nullptr);
    h->small_call = false;
    if (rc) { return rc; }
    HIP_TRY(hipStreamSynchronize(nullptr));
    for (int c = 0; c < h->d.n_channels; c++) {
      memcpy((char *)h_out + (size_t)c * n * h->out_eb, (const char *)h->st.pin_out + (size_t)c * stride * h->out_eb, (size_t)n * h->out_eb);
    }
    return ACDSP_OK;
  }
  if ((rc = h->st.ensure(bin, bout))) { return rc; }
  HIP_TRY(hipMemcpy2D(h->st.d_in, (size_t)stride * h->in_eb, h_in, (size_t)n * h->in_eb, (size_t)n * h->in_eb,
                      (size_t)h->d.n_channels, hipMemcpyHostToDevice));
  if ((rc = acdsp_fir_run(h, h->st.d_in, stride, n, h->st.d_out, stride, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipMemcpy2D(h_out, (size_t)n * h->out_eb, h->st.d_out, (size_t)stride * h->out_eb, (size_t)n * h->out_eb,
                      (size_t)h->d.n_channels, hipMemcpyDeviceToHost));
  return ACDSP_OK;
}

int32_t acdsp_fir_reset(acdsp_fir_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipMemset(h->d_hist[i], 0, (size_t)h->d.n_channels * h->hl * h->in_eb));
    if (h->d_rt[i]) { HIP_TRY(hipMemset(h->d_rt[i], 0, (size_t)h->d.n_channels * h->d.n_taps * h->rt_eb)); }
  }
  h->rt_valid = true; h->rt_since = h->d.n_taps - 1;
  return ACDSP_OK;
}

int32_t acdsp_fir_last_kernel_ms(acdsp_fir_t h, float *ms) {
  if (!h || !ms) { return fail(ACDSP_EINVAL, "null argument"); }
  return h->tm.stats(1, ms, nullptr);
}

int32_t acdsp_fir_kernel_stats(acdsp_fir_t h, int32_t last_k, float *avg_ms, float *min_ms) {
  if (!h) { return fail(ACDSP_EINVAL, "null argument"); }
  return h->tm.stats(last_k, avg_ms, min_ms);
}

int32_t acdsp_fir_mfma_issued(acdsp_fir_t h, int32_t *per_1024_samples) {
  if (!h || !per_1024_samples) { return fail(ACDSP_EINVAL, "null argument"); }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "acdsp_fir_mfma_issued before acdsp_fir_set_coeffs"); }
  *per_1024_samples = 0;
  if (h->path == ACDSP_PATH_MFMA_I8) {
    const acdsp_fir_desc_t &d = h->d;
    FirParams k;
    memset(&k, 0, sizeof k);
    k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = make_dfmt(d.acc); k.out = make_dfmt(d.out);
    k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
    *per_1024_samples = fir_mfma_issued_per_step(k, h->plan);
  }
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// CIC
// ---------------------------------------------------------------------------------------------
namespace {

int log2_ceil_u64(uint64_t x) {
  int lf = 63;
  while (lf > 0 && !((x >> lf) & 1)) { lf--; }
  return (x == (1ull << lf)) ? lf : lf + 1;
}

// find_inter_type_cic_dec / _intr: reference ac_cic_dec_full.h:116-137, ac_cic_intr_full.h:107-127.
// power<> is an `int` enum there, so parameter sets whose product reaches 2^31 do not compile in
// the reference; they are rejected here.
int cic_int_type(const acdsp_cic_desc_t &d, acdsp_fmt_t *it) {
  if (d.R < 1 || d.M < 1 || d.N < 1) { return fail(ACDSP_EINVAL, "CIC: R, M, N must be >= 1"); }
  uint64_t pr = 1, pm = 1;
  const int er = d.interp ? d.N - 1 : d.N;
  for (int i = 0; i < er; i++) { pr *= (uint64_t)d.R; if (pr >= (1ull << 31)) { return fail(ACDSP_EUNSUPPORTED, "CIC: R^N overflows the reference's int power<>"); } }
  for (int i = 0; i < d.N; i++) { pm *= (uint64_t)d.M; if (pm >= (1ull << 31)) { return fail(ACDSP_EUNSUPPORTED, "CIC: M^N overflows the reference's int power<>"); } }
  if (pr * pm >= (1ull << 31)) { return fail(ACDSP_EUNSUPPORTED, "CIC: (R*M)^N overflows the reference's int power<>"); }
  const int outF = d.in.W - d.in.I;
  const int outW = log2_ceil_u64(pr * pm) + d.in.W + (d.in.S ? 0 : 1);
  it->W = outW; it->I = outW - outF; it->S = 1; it->Q = ACDSP_TRN; it->O = ACDSP_WRAP;
  return ACDSP_OK;
}

}  // namespace

extern "C" {

int32_t acdsp_cic_int_type(const acdsp_cic_desc_t *desc, acdsp_fmt_t *it) {
  if (!desc || !it) { return fail(ACDSP_EINVAL, "null argument"); }
  return cic_int_type(*desc, it);
}

int32_t acdsp_cic_create(const acdsp_cic_desc_t *desc, acdsp_cic_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  int rc;
  if ((rc = check_fmt(desc->in, "IN_TYPE")) || (rc = check_fmt(desc->out, "OUT_TYPE", 128))) { return rc; }
  acdsp_fmt_t it;
  if ((rc = cic_int_type(*desc, &it))) { return rc; }
  // INT_TYPE (reference ac_cic_dec_full.h:116-137, ac_cic_intr_full.h:107-127) of up to 128 bits; more than 64 -> wide.hip
  if (it.W > 128) { return fail(ACDSP_EUNSUPPORTED, "CIC: intermediate type needs %d bits (> 128)", it.W); }
  {
    const int fo = desc->out.W - desc->out.I, fi = desc->in.W - desc->in.I;
    if (it.W + (fo > fi ? fo - fi : 0) > 250) { return fail(ACDSP_EUNSUPPORTED, "CIC: OUT_TYPE conversion needs more than 256-bit intermediates"); }
  }
  if (desc->N > kCicMaxN) { return fail(ACDSP_EUNSUPPORTED, "CIC: N=%d > %d", desc->N, kCicMaxN); }
  // rate counters are ac_int<8,false> in the reference (ac_cic_full_core.h:72-73)
  if (desc->R > 256) { return fail(ACDSP_EUNSUPPORTED, "CIC: R=%d > 256 (8-bit rate counter in the reference)", desc->R); }
  if (desc->interp && desc->R < 2) { return fail(ACDSP_EUNSUPPORTED, "CIC interpolator: R=1 never re-arms in the reference (ac_cic_full_core.h:146-158)"); }
  if (desc->interp && desc->N > 255) { return fail(ACDSP_EUNSUPPORTED, "CIC: N too large"); }
  if (desc->n_channels < 1) { return fail(ACDSP_EINVAL, "CIC: n_channels=%d must be positive", desc->n_channels); }
  if (desc->n_channels > 65535) { return fail(ACDSP_EUNSUPPORTED, "CIC: n_channels=%d outside 1..65535", desc->n_channels); }
  if ((rc = check_device(desc->device))) { return rc; }
  acdsp_cic *h = new acdsp_cic();
  h->d = *desc;
  h->it = it;
  h->in_eb = elem_bytes(desc->in.W);
  h->out_eb = elem_bytes(desc->out.W);
  h->me = desc->M < 2 ? desc->M : 2;  // effective comb delay of the reference's delay line, see cic.hip
  h->wide = it.W > 64 || desc->out.W > 64;
  const int64_t mem = desc->interp ? (int64_t)desc->N * h->me + 1 : (int64_t)desc->N * desc->R * h->me - 1;
  h->hl = round_up((int)(mem > 1 ? mem : 1) + 16, kCicTile);   // + 16: the 16-aligned input windows of fir_gen
  const size_t hb = (size_t)desc->n_channels * h->hl * h->in_eb;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc(&h->d_hist[i], hb);
    if (e == hipSuccess) { e = hipMemset(h->d_hist[i], 0, hb); }
  }
  {
    // FIR identity of both directions: h = z^-(N-1) * boxcar(R*M')^N, all arithmetic mod 2^64 (then mod 2^outW)
    const int L = desc->R * h->me;
    std::vector<uint64_t> c(1, 1);
    for (int st = 0; st < desc->N; st++) {
      std::vector<uint64_t> nx(c.size() + L - 1, 0);
      for (size_t i = 0; i < c.size(); i++) { for (int j = 0; j < L; j++) { nx[i + j] += c[i]; } }
      c.swap(nx);
    }
    h->h_taps.assign((size_t)desc->N - 1, 0);
    for (uint64_t v : c) { h->h_taps.push_back((int64_t)v); }
    FirGenPlan probe;
    std::vector<uint32_t> fr;
    static const bool no_gen = getenv("ACDSP_NO_GEN") != nullptr;
    if (e == hipSuccess && ((desc->interp && !no_gen) || h->wide)) {   // interpolator (and wide.hip): polyphase FIR kernel reads the taps themselves
      e = hipMalloc((void **)&h->d_taps, h->h_taps.size() * sizeof(int64_t));
      if (e == hipSuccess) { e = hipMemcpy(h->d_taps, h->h_taps.data(), h->h_taps.size() * sizeof(int64_t), hipMemcpyHostToDevice); }
      // ... and, where the shape is compiled in, the same identity phase by phase on the matrix cores
      const int R = desc->R, n_taps = (int)h->h_taps.size(), kmax = (n_taps + R - 1) / R;
      const int px = (desc->in.W + (desc->in.S ? 0 : 1) + 7) / 8;
      if (e == hipSuccess && desc->interp && !h->wide && !(desc->flags & ACDSP_FLAG_FORCE_GENERIC) && ((px <= 2 && h->in_eb == 2) || (px <= 4 && h->in_eb == 4)) && R <= 32) {
        std::vector<int64_t> E((size_t)R * kmax, 0);
        for (int r = 0; r < R; r++) { for (int k = 0; k < kmax; k++) { if (r + R * k < n_taps) { E[(size_t)r * kmax + k] = h->h_taps[(size_t)(r + R * k)]; } } }
        std::vector<uint32_t> frag;
        std::vector<int64_t> ucorr;
        FirUpPlan pl;
        if (fir_up_plan(E.data(), R, kmax, h->in_eb, &pl, &frag, &ucorr) && pl.pc <= 3 && pl.nb == 1 && fir_up_shape_ok(h->in_eb, h->in_eb, pl.nb, R, h->out_eb)) {
          e = hipMalloc((void **)&h->d_upfrag, frag.size() * sizeof(uint32_t));
          if (e == hipSuccess) { e = hipMalloc((void **)&h->d_upcorr, ucorr.size() * sizeof(int64_t)); }
          if (e == hipSuccess) { e = hipMemcpy(h->d_upfrag, frag.data(), frag.size() * sizeof(uint32_t), hipMemcpyHostToDevice); }
          if (e == hipSuccess) { e = hipMemcpy(h->d_upcorr, ucorr.data(), ucorr.size() * sizeof(int64_t), hipMemcpyHostToDevice); }
          h->up_plan = pl; h->up_px = h->in_eb; h->up_ok = e == hipSuccess;
        }
      }
    }
    h->gen_ok = e == hipSuccess && !desc->interp && !h->wide && !no_gen && (desc->in.W + (desc->in.S ? 0 : 1) + 7) / 8 <= h->in_eb &&
                fir_gen_plan(h->h_taps.data(), (int)h->h_taps.size(), desc->R, 15, &probe, &fr) &&   // worst-case window offset
                fir_gen_plan(h->h_taps.data(), (int)h->h_taps.size(), desc->R, 0, &probe, &fr);
    if (h->gen_ok) { e = hipMalloc((void **)&h->d_gfrag, (size_t)16 * 3 * 8 * 64 * 4 * sizeof(uint32_t)); }
  }
  if (e != hipSuccess || h->tm.init() != ACDSP_OK) {
    acdsp_cic_destroy(h);
    return fail(ACDSP_EHIP, "CIC state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_cic_destroy(acdsp_cic_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  if (h->d_gfrag) { (void)hipFree(h->d_gfrag); }
  if (h->d_taps) { (void)hipFree(h->d_taps); }
  if (h->d_upfrag) { (void)hipFree(h->d_upfrag); }
  if (h->d_upcorr) { (void)hipFree(h->d_upcorr); }
  for (int i = 0; i < 2; i++) {
    if (h->d_hist[i]) { (void)hipFree(h->d_hist[i]); }
  }
  h->tm.destroy();
  h->st.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_cic_clone(acdsp_cic_t h, acdsp_cic_t *out) {
  if (!h || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  acdsp_cic_t c = nullptr;
  int rc = acdsp_cic_create(&h->d, &c);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(c->d_hist[0], h->d_hist[h->cur], (size_t)h->d.n_channels * h->hl * h->in_eb, hipMemcpyDeviceToDevice));
  c->cur = 0;
  c->t_total = h->t_total;
  *out = c;
  return ACDSP_OK;
}

int32_t acdsp_cic_path(acdsp_cic_t h) { return h ? h->last_path : -1; }

int32_t acdsp_cic_reset(acdsp_cic_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) { HIP_TRY(hipMemset(h->d_hist[i], 0, (size_t)h->d.n_channels * h->hl * h->in_eb)); }
  h->t_total = 0;
  return ACDSP_OK;
}

static void cic_window(const acdsp_cic *h, int64_t n_in, CicParams *p) {
  const int R = h->d.R;
  p->t_prev = h->t_total;
  if (!h->d.interp) {
    // decIntgCore emits when rate_cnt == 0, i.e. at global input indices 0, R, 2R, ... (ac_cic_full_core.h:116-133)
    p->phase0 = (int)(h->t_total % R);
    p->first = (R - p->phase0) % R;
    p->q_begin = p->q_end = p->q_skip = 0;
  } else {
    // intrIntg: the call that consumes inputs T..T+K-1 runs iterations [(T-1)R+1, (T+K-1)R+1)
    // (first call starts at 0); the first N-1 iterations ever are dropped (ac_cic_intr_full.h:200-213)
    p->phase0 = 0; p->first = 0;
    p->q_begin = h->t_total == 0 ? 0 : (h->t_total - 1) * R + 1;
    p->q_end = n_in > 0 ? (h->t_total + n_in - 1) * R + 1 : p->q_begin;
    p->q_skip = h->d.N - 1;
  }
}

int64_t acdsp_cic_out_count(acdsp_cic_t h, int64_t n_in) {
  if (!h || n_in < 0) { return -1; }
  if (n_in == 0) { return 0; }
  CicParams p;
  cic_window(h, n_in, &p);
  if (!h->d.interp) { return n_in > p.first ? (n_in - p.first + h->d.R - 1) / h->d.R : 0; }
  int64_t lo = p.q_begin > p.q_skip ? p.q_begin : p.q_skip;
  return p.q_end > lo ? p.q_end - lo : 0;
}

int32_t acdsp_cic_run(acdsp_cic_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                      int64_t *n_out, void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n_in < 0 || (n_in > 0 && (!d_in || in_stride < n_in))) { return fail(ACDSP_EINVAL, "cic_run: bad input arguments"); }
  const int64_t no = acdsp_cic_out_count(h, n_in);
  if (n_out) { *n_out = no; }
  if (n_in == 0) { return ACDSP_OK; }
  if (no > 0 && (!d_out || out_stride < no)) { return fail(ACDSP_EINVAL, "cic_run: output buffer too small for %lld outputs", (long long)no); }
  const acdsp_cic_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  if (stream_is_capturing(s)) {
    if (!d.interp && n_in % d.R != 0) {
      return fail(ACDSP_ESTATE, "cic_run under graph capture: n_in = %lld is not a multiple of R = %d (a replay would repeat the captured decimation phase)",
                  (long long)n_in, d.R);
    }
    if (d.interp && h->t_total == 0) {
      return fail(ACDSP_ESTATE, "cic_run under graph capture: the interpolator's first call drops its start-up outputs and cannot be replayed; run it before capturing");
    }
  }
  CicParams p;
  cic_window(h, n_in, &p);
  p.q_from = p.q_to = 0;
  p.interp = d.interp; p.R = d.R; p.me = h->me; p.N = d.N; p.n_ch = d.n_channels;
  p.w_int = h->it.W;
  p.in = make_dfmt(d.in);
  if (h->wide) { memset(&p.out, 0, sizeof p.out); } else { p.out = make_dfmt(d.out); }
  p.in_eb = h->in_eb; p.out_eb = h->out_eb;
  p.hl = h->hl; p.warm_tiles = h->hl / kCicTile;
  p.vec_ok = ((uintptr_t)d_in % 16 == 0) && ((in_stride * h->in_eb) % 16 == 0);
  p.out_simple = (p.out.F == p.in.F && p.out.O == ACDSP_WRAP) ? ((p.out.S && p.out.W >= h->it.W) ? 2 : 1) : 0;
  p.in_stride = in_stride; p.out_stride = out_stride; p.n_in = n_in;
  p.x = d_in; p.y = d_out; p.hist = h->d_hist[h->cur];
  // chunking: aim at >= 4096 waves, keep the warm-up below ~6 % of a chunk
  const int64_t groups = (d.n_channels + 63) / 64;
  int64_t chunk = (n_in * groups + 4095) / 4096;
  const int64_t floor_chunk = (int64_t)16 * h->hl > 1024 ? (int64_t)16 * h->hl : 1024;
  if (chunk < floor_chunk) { chunk = floor_chunk; }
  p.chunk = (chunk + kCicTile - 1) / kCicTile * kCicTile;
  // decimator on the matrix cores when the FIR identity fits and the rows are slot-aligned
  bool use_gen = h->gen_ok && !d.interp && p.vec_ok && in_stride >= (n_in + 15) / 16 * 16;
  const uint32_t *gfrag = nullptr;
  FirGenPlan gpl;
  if (use_gen) {
    const int fm = (int)(p.first % 16);
    if (!h->gen_have[fm]) {
      std::vector<uint32_t> fr;
      if (!fir_gen_plan(h->h_taps.data(), (int)h->h_taps.size(), d.R, fm, &h->gen_plan[fm], &fr)) { use_gen = false; }
      else {
        HIP_TRY(hipMemcpyAsync(h->d_gfrag + (size_t)fm * 3 * 8 * 64 * 4, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));   // fr is a stack vector
        h->gen_have[fm] = true;
      }
    }
    if (use_gen) { gfrag = h->d_gfrag + (size_t)fm * 3 * 8 * 64 * 4; gpl = h->gen_plan[fm]; }
  }
  const bool use_intr_fir = d.interp && h->d_taps != nullptr && !h->wide;
  h->last_path = h->wide ? ACDSP_PATH_WIDE : (use_gen ? ACDSP_PATH_MFMA_GEN : (use_intr_fir ? ACDSP_PATH_LOSSLESS64 : 0));
  HIP_TRY(hipEventRecord(h->tm.start(), s));
  hipError_t e;
  if (h->wide) {
    CicWideParams pw;
    pw.p = p; pw.out = make_wfmt(d.out);
    e = launch_cic_wide(pw, h->d_taps, (int)h->h_taps.size(), no, s);
  } else if (use_gen) {
    FirParams k;
    memset(&k, 0, sizeof k);
    k.n_ch = d.n_channels;
    k.in = p.in; k.out = p.out; k.acc = p.out; k.cf = p.in;
    k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.hl = h->hl;
    k.in_stride = in_stride; k.out_stride = out_stride; k.n = n_in;
    k.x = d_in; k.y = d_out; k.hist = h->d_hist[h->cur];
    e = launch_fir_gen(k, gpl, gfrag, 1, h->it.W, p.first, no, s);
  } else if (use_intr_fir) {
    // whole steps of 32 input slots on the matrix cores; the head (history, earlier-call phase) and the tail (the call's
    // last input emits only its first iteration, ac_cic_intr_full.h:200-205) on the polyphase VALU kernel
    int64_t q_a = 0, q_b = 0;
    const int64_t lo = p.q_begin > p.q_skip ? p.q_begin : p.q_skip;
    p.q_from = p.q_to = 0;
    static const bool no_up = getenv("ACDSP_NO_CIC_UP") != nullptr;   // A/B knob: polyphase VALU kernel only
    if (h->up_ok && p.vec_ok && !no_up) {
      const int64_t slot_a = h->up_plan.hs, n_steps = ((n_in - 1) / 16 - slot_a) / 32;
      const int64_t out_off = (int64_t)d.R * p.t_prev - lo;
      // 8-byte stores: 4-byte containers may start on odd elements (a continuing call starts R - 1 outputs into a phase group, a first
      // call N - 1): gfx950 serves dword-aligned multi-dword stores
      const int64_t oal = h->out_eb == 4 ? 4 : 8;
      const bool out_ok = ((uintptr_t)d_out % oal == 0) && ((out_stride * h->out_eb) % oal == 0) && ((out_off * h->out_eb) % oal == 0);
      if (n_steps > 0 && out_ok && (int64_t)d.R * (p.t_prev + 16 * slot_a) >= lo) {
        FirParams k;
        memset(&k, 0, sizeof k);
        k.n_ch = d.n_channels; k.in = p.in; k.out = p.out; k.acc = p.out; k.cf = p.in;
        k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.in_stride = in_stride; k.out_stride = out_stride; k.n = n_in; k.x = d_in; k.y = d_out;
        e = launch_fir_up(k, h->up_plan, h->up_px, h->d_upfrag, h->d_upcorr, 1, h->it.W, p.out_simple, 0, -1, slot_a, n_steps, out_off, s);
        if (e == hipSuccess) {
          q_a = (int64_t)d.R * (p.t_prev + 16 * slot_a); q_b = (int64_t)d.R * (p.t_prev + 16 * (slot_a + 32 * n_steps));
          h->last_path = ACDSP_PATH_MFMA_GEN;
        } else if (e != hipErrorNotSupported) {
          return fail(ACDSP_EHIP, "CIC interpolator matrix-core kernel launch failed: %s", hipGetErrorString(e));
        }
      }
    }
    if (q_b > q_a) {
      e = hipSuccess;
      if (q_a > lo) { p.q_from = lo; p.q_to = q_a; e = launch_cic_intr_fir(p, h->d_taps, (int)h->h_taps.size(), s); }
      if (e == hipSuccess && p.q_end > q_b) { p.q_from = q_b; p.q_to = p.q_end; e = launch_cic_intr_fir(p, h->d_taps, (int)h->h_taps.size(), s); }
      p.q_from = p.q_to = 0;
    } else {
      e = launch_cic_intr_fir(p, h->d_taps, (int)h->h_taps.size(), s);
    }
  } else {
    e = launch_cic(p, s);
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "CIC kernel launch failed: %s", hipGetErrorString(e)); }
  HIP_TRY(hipEventRecord(h->tm.stop(), s));
  h->tm.commit();
  const int nxt = hist_next_index(h->cur, p.n_in >= p.hl);
  e = launch_cic_hist_update(p, h->d_hist[nxt], s);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "CIC state kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur = nxt;
  h->t_total += n_in;
  return ACDSP_OK;
}

int32_t acdsp_cic_run_host(acdsp_cic_t h, const void *h_in, int64_t n_in, void *h_out, int64_t out_cap, int64_t *n_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n_in < 0 || (n_in > 0 && !h_in)) { return fail(ACDSP_EINVAL, "cic_run_host: bad arguments"); }
  const int64_t no = acdsp_cic_out_count(h, n_in);
  if (n_out) { *n_out = no; }
  if (n_in == 0) { return ACDSP_OK; }
  if (no > out_cap || (no > 0 && !h_out)) { return fail(ACDSP_EINVAL, "cic_run_host: output capacity %lld < %lld", (long long)out_cap, (long long)no); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  const int64_t si = (n_in + 15) / 16 * 16, so = (no + 7) / 8 * 8 + 8;
  static const bool no_pin = getenv("ACDSP_NO_PINNED") != nullptr;
  if ((size_t)h->d.n_channels * si * h->in_eb <= Staging::kPinBytes && (size_t)h->d.n_channels * so * h->out_eb <= Staging::kPinBytes && !no_pin) {
    if ((rc = h->st.ensure_pinned())) { return rc; }
    for (int c = 0; c < h->d.n_channels; c++) {
      memcpy((char *)h->st.pin_in + (size_t)c * si * h->in_eb, (const char *)h_in + (size_t)c * n_in * h->in_eb, (size_t)n_in * h->in_eb);
    }
    int64_t got = 0;
    if ((rc = acdsp_cic_run(h, h->st.pin_in, si, n_in, h->st.pin_out, so, &got, nullptr))) { return rc; }
    HIP_TRY(hipStreamSynchronize(nullptr));
    for (int c = 0; c < h->d.n_channels && no > 0; c++) {
      memcpy((char *)h_out + (size_t)c * no * h->out_eb, (const char *)h->st.pin_out + (size_t)c * so * h->out_eb, (size_t)no * h->out_eb);
    }
    return ACDSP_OK;
  }
  if ((rc = h->st.ensure((size_t)h->d.n_channels * si * h->in_eb, (size_t)h->d.n_channels * so * h->out_eb))) { return rc; }
  HIP_TRY(hipMemcpy2D(h->st.d_in, (size_t)si * h->in_eb, h_in, (size_t)n_in * h->in_eb, (size_t)n_in * h->in_eb,
                      (size_t)h->d.n_channels, hipMemcpyHostToDevice));
  int64_t no2 = 0;
  if ((rc = acdsp_cic_run(h, h->st.d_in, si, n_in, h->st.d_out, so, &no2, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (no > 0) {
    HIP_TRY(hipMemcpy2D(h_out, (size_t)no * h->out_eb, h->st.d_out, (size_t)so * h->out_eb, (size_t)no * h->out_eb,
                        (size_t)h->d.n_channels, hipMemcpyDeviceToHost));
  }
  return ACDSP_OK;
}

int32_t acdsp_cic_last_kernel_ms(acdsp_cic_t h, float *ms) {
  if (!h || !ms) { return fail(ACDSP_EINVAL, "null argument"); }
  return h->tm.stats(1, ms, nullptr);
}

int32_t acdsp_cic_kernel_stats(acdsp_cic_t h, int32_t last_k, float *avg_ms, float *min_ms) {
  if (!h) { return fail(ACDSP_EINVAL, "null argument"); }
  return h->tm.stats(last_k, avg_ms, min_ms);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// polyphase decimator
// ---------------------------------------------------------------------------------------------
struct acdsp_polydec {
  acdsp_polydec_desc_t d;
  int in_eb, out_eb, hl;
  bool lossless = false, coeffs_set = false, gen_ok = false;
  void *d_hist[2] = {nullptr, nullptr};
  int cur = 0;
  int64_t *d_coeffs = nullptr;
  uint32_t *d_gfrag = nullptr;
  FirGenPlan gplan;
  int last_path = ACDSP_PATH_GENERIC;
  Staging st;
};

extern "C" {

int32_t acdsp_polydec_create(const acdsp_polydec_desc_t *desc, acdsp_polydec_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  if (desc->n_taps < 1 || desc->df < 1 || (int64_t)desc->n_taps * desc->df > 2048) {
    return fail(ACDSP_EUNSUPPORTED, "poly_dec: NTAPS*DF = %lld outside 1..2048", (long long)desc->n_taps * desc->df);
  }
  if (desc->n_channels < 1) { return fail(ACDSP_EINVAL, "n_channels=%d must be positive", desc->n_channels); }
  if (desc->n_channels > 65535) { return fail(ACDSP_EUNSUPPORTED, "n_channels=%d outside 1..65535", desc->n_channels); }
  int rc;
  if ((rc = check_fmt(desc->in, "IN_TYPE")) || (rc = check_fmt(desc->coeff, "COEFF_TYPE")) || (rc = check_fmt(desc->acc, "ACC_TYPE")) ||
      (rc = check_fmt(desc->out, "OUT_TYPE"))) {
    return rc;
  }
  const int fi = desc->in.W - desc->in.I, fc = desc->coeff.W - desc->coeff.I, fa = desc->acc.W - desc->acc.I;
  const int f = fi + fc > fa ? fi + fc : fa;
  if (desc->in.W + desc->coeff.W + 2 + (f - fi - fc) > 125 || desc->acc.W + (f - fa) > 125) {
    return fail(ACDSP_EUNSUPPORTED, "type combination needs more than 128-bit intermediates");
  }
  if ((rc = check_device(desc->device))) { return rc; }
  acdsp_polydec *h = new acdsp_polydec();
  h->d = *desc;
  h->in_eb = elem_bytes(desc->in.W);
  h->out_eb = elem_bytes(desc->out.W);
  h->hl = round_up(desc->n_taps * desc->df + 15, 32);
  h->lossless = desc->acc.O == ACDSP_WRAP && fa >= fi + fc && fa - fi - fc < 64;
  const size_t hb = (size_t)desc->n_channels * h->hl * h->in_eb;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc(&h->d_hist[i], hb);
    if (e == hipSuccess) { e = hipMemset(h->d_hist[i], 0, hb); }
  }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_coeffs, (size_t)desc->n_taps * desc->df * sizeof(int64_t)); }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_gfrag, 3 * 8 * 64 * 4 * sizeof(uint32_t)); }
  if (e != hipSuccess) {
    acdsp_polydec_destroy(h);
    return fail(ACDSP_EHIP, "poly_dec state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_polydec_destroy(acdsp_polydec_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  for (int i = 0; i < 2; i++) { if (h->d_hist[i]) { (void)hipFree(h->d_hist[i]); } }
  if (h->d_coeffs) { (void)hipFree(h->d_coeffs); }
  if (h->d_gfrag) { (void)hipFree(h->d_gfrag); }
  h->st.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_polydec_set_coeffs(acdsp_polydec_t h, const int64_t *coeffs) {
  if (!h || !coeffs) { return fail(ACDSP_EINVAL, "null argument"); }
  const acdsp_polydec_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  const int n = d.n_taps * d.df;
  const acdsp::DFmt cf = make_dfmt(d.coeff);
  for (int i = 0; i < n; i++) {
    if (coeffs[i] < cf.lo || coeffs[i] > cf.hi) { return fail(ACDSP_EINVAL, "coefficient %d = %lld is not a COEFF_TYPE raw word", i, (long long)coeffs[i]); }
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->d_coeffs, coeffs, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
  h->gen_ok = false;
  static const bool no_gen = getenv("ACDSP_NO_GEN") != nullptr;
  if (h->lossless && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && !no_gen && (d.in.W + (d.in.S ? 0 : 1) + 7) / 8 <= h->in_eb) {
    // decimating FIR  y[g] = sum_k hh[k] x[g*DF + DF-1 - k],  hh[df + tp*DF] = c[tp + NTAPS*df]
    std::vector<int64_t> hh((size_t)n, 0);
    for (int df = 0; df < d.df; df++) { for (int tp = 0; tp < d.n_taps; tp++) { hh[df + tp * d.df] = coeffs[tp + d.n_taps * df]; } }
    std::vector<uint32_t> fr;
    if (fir_gen_plan(hh.data(), n, d.df, (d.df - 1) % 16, &h->gplan, &fr)) {
      HIP_TRY(hipMemcpy(h->d_gfrag, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      h->gen_ok = true;
    }
  }
  h->coeffs_set = true;
  return ACDSP_OK;
}

int32_t acdsp_polydec_path(acdsp_polydec_t h) { return h ? h->last_path : -1; }

int32_t acdsp_polydec_run(acdsp_polydec_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                          void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  const acdsp_polydec_desc_t &d = h->d;
  if (n_in < 0 || n_in % d.df != 0) { return fail(ACDSP_EINVAL, "poly_dec: n_in = %lld is not a multiple of DF = %d", (long long)n_in, d.df); }
  const int64_t n_out = n_in / d.df;
  if (n_in > 0 && (!d_in || !d_out || in_stride < n_in || out_stride < n_out)) { return fail(ACDSP_EINVAL, "poly_dec: bad buffer arguments"); }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "poly_dec run before set_coeffs"); }
  if (n_in == 0) { return ACDSP_OK; }
  int rc = check_device(d.device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  FirParams k;
  memset(&k, 0, sizeof k);
  k.n_taps = d.n_taps * d.df; k.ftype = ACDSP_SHIFT_REG; k.n_ch = d.n_channels;
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = make_dfmt(d.acc); k.out = make_dfmt(d.out);
  k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.hl = h->hl;
  k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
  k.in_stride = in_stride; k.out_stride = out_stride; k.n = n_in;
  k.x = d_in; k.y = d_out; k.hist = h->d_hist[h->cur]; k.coeffs = h->d_coeffs;
  const bool aligned = ((uintptr_t)d_in % 16 == 0) && ((in_stride * h->in_eb) % 16 == 0) && in_stride >= (n_in + 15) / 16 * 16;
  hipError_t e;
  if (h->gen_ok && aligned) {
    h->last_path = ACDSP_PATH_MFMA_GEN;
    e = launch_fir_gen(k, h->gplan, h->d_gfrag, 0, 0, d.df - 1, n_out, s);
  } else {
    h->last_path = ACDSP_PATH_GENERIC;
    e = launch_polydec_generic(k, d.n_taps, d.df, n_out, s);
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "poly_dec kernel launch failed: %s", hipGetErrorString(e)); }
  const int nxt = hist_next_index(h->cur, k.n >= k.hl);
  e = launch_fir_hist_update(k, h->d_hist[nxt], s);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "poly_dec state kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur = nxt;
  return ACDSP_OK;
}

int32_t acdsp_polydec_run_host(acdsp_polydec_t h, const void *h_in, int64_t n_in, void *h_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n_in < 0 || n_in % h->d.df != 0 || (n_in > 0 && (!h_in || !h_out))) { return fail(ACDSP_EINVAL, "poly_dec run_host: bad arguments"); }
  if (n_in == 0) { return ACDSP_OK; }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  const int64_t n_out = n_in / h->d.df;
  const int64_t si = (n_in + 15) / 16 * 16, so = (n_out + 7) / 8 * 8;
  if ((rc = h->st.ensure((size_t)h->d.n_channels * si * h->in_eb, (size_t)h->d.n_channels * so * h->out_eb))) { return rc; }
  HIP_TRY(hipMemcpy2D(h->st.d_in, (size_t)si * h->in_eb, h_in, (size_t)n_in * h->in_eb, (size_t)n_in * h->in_eb,
                      (size_t)h->d.n_channels, hipMemcpyHostToDevice));
  if ((rc = acdsp_polydec_run(h, h->st.d_in, si, n_in, h->st.d_out, so, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipMemcpy2D(h_out, (size_t)n_out * h->out_eb, h->st.d_out, (size_t)so * h->out_eb, (size_t)n_out * h->out_eb,
                      (size_t)h->d.n_channels, hipMemcpyDeviceToHost));
  return ACDSP_OK;
}

int32_t acdsp_polydec_reset(acdsp_polydec_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) { HIP_TRY(hipMemset(h->d_hist[i], 0, (size_t)h->d.n_channels * h->hl * h->in_eb)); }
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// DDC cascade: ac_cic_dec_full -> ac_fir_* on the decimator's lossless INT_TYPE words
// ---------------------------------------------------------------------------------------------
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

extern "C" {

int32_t acdsp_ddc_destroy(acdsp_ddc_t h) {
  if (!h) { return ACDSP_OK; }
  if (h->cic) { (void)hipSetDevice(h->cic->d.device); }
  for (int i = 0; i < 2; i++) { if (h->d_hist[i]) { (void)hipFree(h->d_hist[i]); } }
  if (h->d_fragA) { (void)hipFree(h->d_fragA); }
  if (h->d_fragB) { (void)hipFree(h->d_fragB); }
  if (h->d_mid) { (void)hipFree(h->d_mid); }
  if (h->cic) { acdsp_cic_destroy(h->cic); }
  if (h->fir) { acdsp_fir_destroy(h->fir); }
  h->tm.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_ddc_create(const acdsp_cic_desc_t *cic, const acdsp_fir_desc_t *fir, acdsp_ddc_t *out) {
  if (!cic || !fir || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  if (cic->interp) { return fail(ACDSP_EINVAL, "ddc: stage A must be the decimator"); }
  if (cic->n_channels != fir->n_channels || cic->device != fir->device) { return fail(ACDSP_EINVAL, "ddc: both stages must cover the same channels on one device"); }
  if (fir->coeffs_per_channel) { return fail(ACDSP_EUNSUPPORTED, "ddc: one shared coefficient set"); }
  acdsp_ddc *h = new acdsp_ddc();
  int rc = acdsp_cic_create(cic, &h->cic);
  if (!rc) { rc = acdsp_fir_create(fir, &h->fir); }
  if (rc) { acdsp_ddc_destroy(h); return rc; }
  // the cascade is lossless in the middle: the decimator writes its INT_TYPE and the FIR reads exactly that type
  const acdsp_fmt_t &it = h->cic->it;
  const bool same = [&](const acdsp_fmt_t &f) { return f.W == it.W && f.I == it.I && f.S == it.S; }(cic->out) && fir->in.W == it.W &&
                    fir->in.I == it.I && fir->in.S == it.S;
  if (!same) {
    acdsp_ddc_destroy(h);
    return fail(ACDSP_EINVAL, "ddc: the decimator's OUT_TYPE and the FIR's IN_TYPE must both be the INT_TYPE <%d,%d>", it.W, it.I);
  }
  static const bool no_fuse = getenv("ACDSP_NO_FUSE") != nullptr;
  h->fused = !no_fuse && h->cic->gen_ok && h->fir->lossless && !(fir->flags & ACDSP_FLAG_FORCE_GENERIC) &&
             !(cic->flags & ACDSP_FLAG_FORCE_GENERIC) && h->cic->in_eb == 2 && h->fir->out_eb == 4;
  hipError_t e = hipSuccess;
  if (h->fused) {
    h->hl = round_up(256 * cic->R + (int)h->cic->h_taps.size() + 48, 64);
    const size_t hb = (size_t)cic->n_channels * h->hl * h->cic->in_eb;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
      e = hipMalloc(&h->d_hist[i], hb);
      if (e == hipSuccess) { e = hipMemset(h->d_hist[i], 0, hb); }
    }
    if (e == hipSuccess) { e = hipMalloc((void **)&h->d_fragA, (size_t)16 * 3 * 8 * 64 * 4 * sizeof(uint32_t)); }
    if (e == hipSuccess) { e = hipMalloc((void **)&h->d_fragB, (size_t)3 * 8 * 64 * 4 * sizeof(uint32_t)); }
  }
  if (e != hipSuccess || h->tm.init() != ACDSP_OK) {
    acdsp_ddc_destroy(h);
    return fail(ACDSP_EHIP, "ddc state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_ddc_set_coeffs(acdsp_ddc_t h, const int64_t *coeffs) {
  if (!h || !coeffs) { return fail(ACDSP_EINVAL, "null argument"); }
  int rc = acdsp_fir_set_coeffs(h->fir, coeffs);   // validation + the two-kernel path's own fragments
  if (rc) { return rc; }
  if (h->fused) {
    const acdsp_fir_desc_t &d = h->fir->d;
    std::vector<int64_t> eff = effective_coeffs(coeffs, d.n_taps, internal_ftype(d.kind, d.ftype));
    std::vector<uint32_t> fr;
    if (!fir_gen_plan(eff.data(), d.n_taps, 1, 0, &h->planB, &fr) || fr.size() > (size_t)3 * 8 * 64 * 4 || h->planB.nb > 3 ||
        h->planB.pc > 2 || h->planB.off != 128) {
      if (h->t_total != 0) { return fail(ACDSP_EUNSUPPORTED, "ddc: this coefficient set does not fit the fused kernel and the stream has started"); }
      h->fused = false;   // before the first sample: the two kernels for the handle's lifetime
    } else {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(h->d_fragB, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
  }
  h->coeffs_set = true;
  return ACDSP_OK;
}

int64_t acdsp_ddc_out_count(acdsp_ddc_t h, int64_t n_in) {
  if (!h || n_in < 0) { return -1; }
  if (!h->fused) { return acdsp_cic_out_count(h->cic, n_in); }
  const int R = h->cic->d.R;
  const int64_t first = (R - h->t_total % R) % R;
  return n_in > first ? (n_in - first + R - 1) / R : 0;
}

int32_t acdsp_ddc_path(acdsp_ddc_t h) { return h ? (h->fused ? 1 : 0) : -1; }

int32_t acdsp_ddc_reset(acdsp_ddc_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = acdsp_cic_reset(h->cic);
  if (!rc) { rc = acdsp_fir_reset(h->fir); }
  if (rc) { return rc; }
  for (int i = 0; i < 2; i++) {
    if (h->d_hist[i]) { HIP_TRY(hipMemset(h->d_hist[i], 0, (size_t)h->cic->d.n_channels * h->hl * h->cic->in_eb)); }
  }
  h->t_total = 0;
  return ACDSP_OK;
}

int32_t acdsp_ddc_run(acdsp_ddc_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                      int64_t *n_out, void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "ddc_run before acdsp_ddc_set_coeffs"); }
  if (n_in < 0 || (n_in > 0 && (!d_in || in_stride < n_in))) { return fail(ACDSP_EINVAL, "ddc_run: bad input arguments"); }
  const int64_t no = acdsp_ddc_out_count(h, n_in);
  if (n_out) { *n_out = no; }
  if (n_in == 0) { return ACDSP_OK; }
  if (no > 0 && (!d_out || out_stride < no)) { return fail(ACDSP_EINVAL, "ddc_run: output buffer too small for %lld outputs", (long long)no); }
  const acdsp_cic_desc_t &cd = h->cic->d;
  const acdsp_fir_desc_t &fd = h->fir->d;
  int rc = check_device(cd.device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  if (!h->fused) {
    // two kernels with the INT_TYPE stream in HBM between them
    const int64_t cap = (no + 15) / 16 * 16 + 16;
    if (cap > h->mid_cap) {
      HIP_TRY(hipStreamSynchronize(s));
      if (h->d_mid) { HIP_TRY(hipFree(h->d_mid)); h->d_mid = nullptr; }
      HIP_TRY(hipMalloc(&h->d_mid, (size_t)cd.n_channels * cap * h->cic->out_eb));
      h->mid_cap = cap;
    }
    int64_t nn = 0;
    HIP_TRY(hipEventRecord(h->tm.start(), s));
    rc = acdsp_cic_run(h->cic, d_in, in_stride, n_in, h->d_mid, h->mid_cap, &nn, stream);
    if (!rc && nn > 0) { rc = acdsp_fir_run(h->fir, h->d_mid, h->mid_cap, nn, d_out, out_stride, stream); }
    HIP_TRY(hipEventRecord(h->tm.stop(), s));
    h->tm.commit();
    return rc;
  }
  // fused: alignment is the only per-call requirement
  if ((uintptr_t)d_in % 16 || (in_stride * h->cic->in_eb) % 16 || in_stride < (n_in + 15) / 16 * 16 || (uintptr_t)d_out % 16 ||
      (out_stride * h->fir->out_eb) % 16) {
    return fail(ACDSP_EUNSUPPORTED, "ddc_run (fused): rows must be 16-byte aligned and readable up to a multiple of 16 samples");
  }
  const int R = cd.R;
  if (stream_is_capturing((hipStream_t)stream) && n_in % R != 0) {
    return fail(ACDSP_ESTATE, "ddc_run under graph capture: n_in = %lld is not a multiple of R = %d (a replay would repeat the captured decimation phase)",
                (long long)n_in, R);
  }
  const int64_t first = (R - h->t_total % R) % R;
  const int fm = (int)(first % 16);
  if (!h->haveA[fm]) {
    std::vector<uint32_t> fr;
    if (!fir_gen_plan(h->cic->h_taps.data(), (int)h->cic->h_taps.size(), R, fm, &h->planA[fm], &fr) || fr.size() > (size_t)3 * 8 * 64 * 4) {
      return fail(ACDSP_EUNSUPPORTED, "ddc_run (fused): decimator shape outside the fused kernel");
    }
    HIP_TRY(hipMemcpyAsync(h->d_fragA + (size_t)fm * 3 * 8 * 64 * 4, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->haveA[fm] = true;
  }
  FirParams pa, pb;
  memset(&pa, 0, sizeof pa); memset(&pb, 0, sizeof pb);
  pa.n_ch = cd.n_channels; pa.in = make_dfmt(cd.in); pa.out = make_dfmt(cd.out); pa.acc = pa.out; pa.cf = pa.in;
  pa.in_eb = h->cic->in_eb; pa.out_eb = h->cic->out_eb; pa.hl = h->hl;
  pa.in_stride = in_stride; pa.n = n_in; pa.x = d_in; pa.hist = h->d_hist[h->cur];
  pb.n_ch = fd.n_channels; pb.in = make_dfmt(fd.in); pb.cf = make_dfmt(fd.coeff); pb.acc = make_dfmt(fd.acc); pb.out = make_dfmt(fd.out);
  pb.in_eb = h->fir->in_eb; pb.out_eb = h->fir->out_eb;
  pb.lossless_shift = pb.acc.F - pb.in.F - pb.cf.F;
  pb.y = d_out; pb.out_stride = out_stride; pb.n = no;
  HIP_TRY(hipEventRecord(h->tm.start(), s));
  hipError_t e = launch_cascade(pa, h->planA[fm], h->d_fragA + (size_t)fm * 3 * 8 * 64 * 4, h->cic->it.W, first, pb, h->planB, h->d_fragB, no, s);
  if (e == hipErrorNotSupported) { return fail(ACDSP_EUNSUPPORTED, "ddc_run (fused): shape outside the compiled cascade kernel"); }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "cascade kernel launch failed: %s", hipGetErrorString(e)); }
  HIP_TRY(hipEventRecord(h->tm.stop(), s));
  h->tm.commit();
  const int nxt = hist_next_index(h->cur, pa.n >= pa.hl);
  e = launch_fir_hist_update(pa, h->d_hist[nxt], s);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "ddc state kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur = nxt;
  h->t_total += n_in;
  return ACDSP_OK;
}

int32_t acdsp_ddc_kernel_stats(acdsp_ddc_t h, int32_t last_k, float *avg_ms, float *min_ms) {
  if (!h) { return fail(ACDSP_EINVAL, "null argument"); }
  return h->tm.stats(last_k, avg_ms, min_ms);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// polyphase interpolator (second half): ac_poly_intr
// ---------------------------------------------------------------------------------------------
struct acdsp_polyintr {
  acdsp_polyintr_desc_t d;
  int in_eb, out_eb, hl;
  bool ctrl_set = false;
  void *d_hist[2] = {nullptr, nullptr};
  int64_t *d_saved[2] = {nullptr, nullptr};   // sums of the last sample, emitted by the next call (folded cores)
  int cur = 0;
  int64_t t_total = 0;
  int64_t *d_coeffs = nullptr;
  uint8_t *d_sign = nullptr, *d_corr = nullptr;
  // exact-accumulation class on the matrix cores (fir_up.hip): folded per-phase taps of the current control words
  bool up_ok = false;
  bool acc64_ok = false;        // a 64-bit ACC_TYPE whose sums the current control words keep inside 62 bits: the exact-accumulation class applies
  int up_px = 2;                // input byte planes of the matrix-core kernel (= container bytes)
  FirUpPlan up_plan;
  uint32_t up_shmask = 0;
  int64_t up_max_abs = -1;      // bound on |z| of the folded taps (enables the 32-bit epilogue)
  uint32_t *d_upfrag = nullptr;
  int64_t *d_upcorr = nullptr;
  int last_path = ACDSP_PATH_GENERIC;
  Staging st;
};

namespace {

// Per-phase taps of ac_poly_intr as one linear filter of the input (host side; see fir_up.hip).  E[j][k] multiplies
// x[n - k] in the output group of input sample n.  Returns false when the control words make the cores non-linear in the
// input (a phase with sign[j] = 0 negates samples in IN_TYPE: -min(IN_TYPE) is not representable).
bool polyintr_linear_taps(const acdsp_polyintr_desc_t &d, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr,
                          std::vector<int64_t> *E, int *nt, uint32_t *sh_mask, __int128 *max_abs_sum) {
  const int N = d.n_taps, L = d.ifac;
  std::vector<std::vector<__int128>> sub((size_t)L, std::vector<__int128>((size_t)N, 0));   // sub-filter sums acc_n[j] = sum_k sub[j][k] x[n-k]
  for (int j = 0; j < L; j++) {
    if (d.ftype == ACDSP_POLY_FOLD_ANTI) {
      for (int i = 0; i < N; i++) { sub[j][i] = coeffs[i + N * j]; }                           // ac_poly_intr.h:246-256
      continue;
    }
    if (!sign[j]) { return false; }
    if (d.ftype == ACDSP_POLY_FOLD_EVEN) {                                                      // :141-151
      for (int i = 0; i < N / 2; i++) { const int64_t c = coeffs[i + j * N / 2]; sub[j][i] += c; sub[j][N - 1 - i] += c; }
    } else {                                                                                    // :194-209
      const int mid = (N - 1) / 2;
      for (int i = 0; i <= mid; i++) {
        const int64_t c = coeffs[i + (N / 2 + 1) * j];
        sub[j][i] += c;
        if (i != mid) { sub[j][N - 1 - i] += c; }
      }
    }
  }
  *max_abs_sum = 0;
  for (int j = 0; j < L; j++) {
    __int128 sa = 0;
    for (int k = 0; k < N; k++) { sa += sub[j][k] < 0 ? -sub[j][k] : sub[j][k]; }
    if (sa > *max_abs_sum) { *max_abs_sum = sa; }
  }
  const int lag = d.ftype == ACDSP_POLY_FOLD_ANTI ? 0 : 1;   // banks: the sums of sample n-1 leave with sample n (:153-175)
  *nt = N + lag;
  *sh_mask = 0;
  E->assign((size_t)L * (size_t)*nt, 0);
  for (int j = 0; j < L; j++) {
    const int cj = d.ftype == ACDSP_POLY_FOLD_ANTI ? j : corr[j];
    for (int k = 0; k < N; k++) {
      __int128 g = sub[j][k];
      if (cj != j) { g -= sub[cj][k]; }                      // (t1 + ACC(-t2)) >> 1 with sign[j] set (:164-172)
      if (g < INT64_MIN / 4 || g > INT64_MAX / 4) { return false; }
      (*E)[(size_t)j * *nt + k + lag] = (int64_t)g;
    }
    if (cj != j) { *sh_mask |= 1u << j; }
  }
  return true;
}

}  // namespace

extern "C" {

int32_t acdsp_polyintr_destroy(acdsp_polyintr_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  for (int i = 0; i < 2; i++) {
    if (h->d_hist[i]) { (void)hipFree(h->d_hist[i]); }
    if (h->d_saved[i]) { (void)hipFree(h->d_saved[i]); }
  }
  if (h->d_coeffs) { (void)hipFree(h->d_coeffs); }
  if (h->d_sign) { (void)hipFree(h->d_sign); }
  if (h->d_corr) { (void)hipFree(h->d_corr); }
  if (h->d_upfrag) { (void)hipFree(h->d_upfrag); }
  if (h->d_upcorr) { (void)hipFree(h->d_upcorr); }
  h->st.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_polyintr_path(acdsp_polyintr_t h) { return h ? h->last_path : -1; }

int32_t acdsp_polyintr_create(const acdsp_polyintr_desc_t *desc, acdsp_polyintr_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  const acdsp_polyintr_desc_t &d = *desc;
  if (d.ftype < ACDSP_POLY_FOLD_EVEN || d.ftype > ACDSP_POLY_FOLD_ANTI) { return fail(ACDSP_EINVAL, "bad poly_intr ftype %d", d.ftype); }
  if (d.n_taps < 1 || d.n_taps > 2048) { return fail(ACDSP_EUNSUPPORTED, "NTAPS=%d outside 1..2048", d.n_taps); }
  if (d.ifac < 1 || d.ifac > 255) { return fail(ACDSP_EUNSUPPORTED, "IF=%d outside 1..255 (corr[] is ac_int<8,false>)", d.ifac); }
  if (d.coeff_sz < 1 || d.coeff_sz > (1 << 20)) { return fail(ACDSP_EUNSUPPORTED, "COEFFSZ=%d outside 1..2^20", d.coeff_sz); }
  if (d.n_channels < 1) { return fail(ACDSP_EINVAL, "n_channels=%d must be positive", d.n_channels); }
  if (d.n_channels > 65535) { return fail(ACDSP_EUNSUPPORTED, "n_channels=%d outside 1..65535", d.n_channels); }
  int rc;
  if ((rc = check_fmt(d.in, "IN_TYPE")) || (rc = check_fmt(d.coeff, "COEFF_TYPE")) || (rc = check_fmt(d.acc, "ACC_TYPE")) ||
      (rc = check_fmt(d.out, "OUT_TYPE"))) {
    return rc;
  }
  {  // 128-bit exact intermediates: product coeff * fold (ACC_TYPE) or taps * coeff, aligned with the accumulator
    const int fi = d.in.W - d.in.I, fc = d.coeff.W - d.coeff.I, fa = d.acc.W - d.acc.I;
    const int wp = d.ftype == ACDSP_POLY_FOLD_ANTI ? d.in.W + d.coeff.W + 2 : d.acc.W + d.coeff.W + 1;
    const int fp = d.ftype == ACDSP_POLY_FOLD_ANTI ? fi + fc : fa + fc;
    const int f = fp > fa ? fp : fa;
    if (wp + (f - fp) > 125 || d.acc.W + (f - fa) > 125) { return fail(ACDSP_EUNSUPPORTED, "type combination needs more than 128-bit intermediates"); }
  }
  if ((rc = check_device(d.device))) { return rc; }
  acdsp_polyintr *h = new acdsp_polyintr();
  h->d = d;
  h->in_eb = elem_bytes(d.in.W); h->out_eb = elem_bytes(d.out.W);
  h->hl = round_up(d.n_taps + 15, 32);
  hipError_t e = hipSuccess;
  const size_t hb = (size_t)d.n_channels * h->hl * h->in_eb, sb = (size_t)d.n_channels * d.ifac * sizeof(int64_t);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc(&h->d_hist[i], hb);
    if (e == hipSuccess) { e = hipMemset(h->d_hist[i], 0, hb); }
    if (e == hipSuccess) { e = hipMalloc((void **)&h->d_saved[i], sb); }
    if (e == hipSuccess) { e = hipMemset(h->d_saved[i], 0, sb); }   // acc_a / acc_b start at 0 (ac_poly_intr.h:117-118)
  }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_coeffs, (size_t)d.coeff_sz * sizeof(int64_t)); }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_sign, (size_t)d.ifac); }
  if (e == hipSuccess) { e = hipMalloc((void **)&h->d_corr, (size_t)d.ifac); }
  if (e != hipSuccess) {
    acdsp_polyintr_destroy(h);
    return fail(ACDSP_EHIP, "poly_intr state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_polyintr_set_ctrl(acdsp_polyintr_t h, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr) {
  if (!h || !coeffs || !sign || !corr) { return fail(ACDSP_EINVAL, "null argument"); }
  const acdsp_polyintr_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  const acdsp::DFmt cf = make_dfmt(d.coeff);
  for (int i = 0; i < d.coeff_sz; i++) {
    if (coeffs[i] < cf.lo || coeffs[i] > cf.hi) { return fail(ACDSP_EINVAL, "coefficient %d = %lld is not a COEFF_TYPE raw word", i, (long long)coeffs[i]); }
  }
  const int N = d.n_taps, J = d.ifac - 1;
  const int max_ci = d.ftype == ACDSP_POLY_FOLD_EVEN ? (N / 2 - 1) + J * N / 2
                     : d.ftype == ACDSP_POLY_FOLD_ODD ? (N - 1) / 2 + (N / 2 + 1) * J : (N - 1) + N * J;
  if (max_ci >= d.coeff_sz) { return fail(ACDSP_EINVAL, "the reference would read coeffs[%d] of coeffs[COEFFSZ = %d]", max_ci, d.coeff_sz); }
  for (int j = 0; j < d.ifac; j++) {
    if (d.ftype != ACDSP_POLY_FOLD_ANTI && corr[j] >= d.ifac) { return fail(ACDSP_EINVAL, "corr[%d] = %d indexes outside the IF = %d accumulator banks", j, corr[j], d.ifac); }
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->d_coeffs, coeffs, (size_t)d.coeff_sz * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_sign, sign, (size_t)d.ifac, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_corr, corr, (size_t)d.ifac, hipMemcpyHostToDevice));
  h->ctrl_set = true;
  // matrix-core path: exact-accumulation class, int16 samples, control words that keep the cores linear, and an
  // accumulator that cannot wrap (the symmetric-pair halving (t1 -/+ t2) >> 1 does not commute with a wrap)
  h->up_ok = false;
  static const bool no_gen = getenv("ACDSP_NO_GEN") != nullptr;
  const int fi = d.in.W - d.in.I, fc = d.coeff.W - d.coeff.I, fa = d.acc.W - d.acc.I, ls = fa - fi - fc;
  // (a 64-bit ACC_TYPE -- the header's own usage example, ac_poly_intr.h:45-48: <32,16> samples and coefficients into <64,32> -- belongs to
  // the class when the control words bound every sub-filter sum to 62 bits: nothing can wrap and the pair sum t1 -/+ t2 stays inside int64)
  const bool lossless = d.in.S && d.acc.S && d.acc.O == ACDSP_WRAP && ls >= 0 && ls < 64 && fa >= fi && d.acc.I >= d.in.I + 1 && d.acc.W <= 64;
  h->acc64_ok = false;
  if (lossless && d.acc.W == 64) {
    std::vector<int64_t> E;
    int nt = 0;
    uint32_t shm = 0;
    __int128 sa = 0;
    h->acc64_ok = polyintr_linear_taps(d, coeffs, sign, corr, &E, &nt, &shm, &sa) && d.in.W - 1 + ls < 100 && (sa << (d.in.W - 1 + ls)) < ((__int128)1 << 62);
  }
  const int upx = h->in_eb;
  if (lossless && (d.acc.W <= 63 || h->acc64_ok) && !no_gen && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && (h->in_eb == 2 || h->in_eb == 4) && d.ifac <= 32) {
    std::vector<int64_t> E;
    int nt = 0;
    uint32_t shm = 0;
    __int128 sa = 0;
    std::vector<uint32_t> frag;
    std::vector<int64_t> ucorr;
    FirUpPlan pl;
    if (polyintr_linear_taps(d, coeffs, sign, corr, &E, &nt, &shm, &sa) &&
        // |acc| <= sum|taps| * 2^(W_in - 1) << ls must stay inside ACC_TYPE; a pair sum then fits one more bit
        (sa << (d.in.W - 1 + ls)) < ((__int128)1 << ((d.acc.W < 64 ? d.acc.W : 63) - 1)) &&
        fir_up_plan(E.data(), d.ifac, nt, upx, &pl, &frag, &ucorr) && fir_up_shape_ok(h->in_eb, upx, pl.nb, d.ifac, h->out_eb)) {
      if (h->d_upfrag) { (void)hipFree(h->d_upfrag); h->d_upfrag = nullptr; }
      if (h->d_upcorr) { (void)hipFree(h->d_upcorr); h->d_upcorr = nullptr; }
      HIP_TRY(hipMalloc((void **)&h->d_upfrag, frag.size() * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void **)&h->d_upcorr, ucorr.size() * sizeof(int64_t)));
      HIP_TRY(hipMemcpy(h->d_upfrag, frag.data(), frag.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(h->d_upcorr, ucorr.data(), ucorr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      h->up_plan = pl; h->up_shmask = shm; h->up_ok = true; h->up_px = upx;
      {  // |z| <= max_j sum_k |E_j[k]| * 2^(W_in - 1)
        __int128 worst = 0;
        for (int j = 0; j < d.ifac; j++) {
          __int128 sj = 0;
          for (int k = 0; k < nt; k++) { const int64_t v = E[(size_t)j * nt + k]; sj += v < 0 ? -(__int128)v : (__int128)v; }
          if (sj > worst) { worst = sj; }
        }
        worst <<= (d.in.W - 1);
        h->up_max_abs = worst < ((__int128)1 << 62) ? (int64_t)worst : -1;
      }
    }
  }
  return ACDSP_OK;
}

int64_t acdsp_polyintr_out_count(acdsp_polyintr_t h, int64_t n_in) {
  if (!h || n_in < 0) { return -1; }
  if (n_in == 0) { return 0; }
  const int64_t groups = (h->d.ftype != ACDSP_POLY_FOLD_ANTI && h->t_total == 0) ? n_in - 1 : n_in;   // `init` (:175)
  return groups * h->d.ifac;
}

int32_t acdsp_polyintr_run(acdsp_polyintr_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                           int64_t *n_out, void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (!h->ctrl_set) { return fail(ACDSP_ESTATE, "poly_intr run before acdsp_polyintr_set_ctrl (the reference reads uninitialised structs)"); }
  if (n_in < 0 || (n_in > 0 && (!d_in || in_stride < n_in))) { return fail(ACDSP_EINVAL, "poly_intr run: bad input arguments"); }
  const int64_t no = acdsp_polyintr_out_count(h, n_in);
  if (n_out) { *n_out = no; }
  if (n_in == 0) { return ACDSP_OK; }
  if (no > 0 && (!d_out || out_stride < no)) { return fail(ACDSP_EINVAL, "poly_intr run: output buffer too small for %lld outputs", (long long)no); }
  if (stream_is_capturing((hipStream_t)stream) && h->d.ftype != ACDSP_POLY_FOLD_ANTI && h->t_total == 0) {
    return fail(ACDSP_ESTATE, "poly_intr run under graph capture: the stream's first call emits one group less and cannot be replayed; run it before capturing");
  }
  const acdsp_polyintr_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  PolyIntrParams p;
  memset(&p, 0, sizeof p);
  p.n_taps = d.n_taps; p.coeff_sz = d.coeff_sz; p.ifac = d.ifac; p.ftype = d.ftype; p.n_ch = d.n_channels;
  p.in = make_dfmt(d.in); p.cf = make_dfmt(d.coeff); p.acc = make_dfmt(d.acc); p.out = make_dfmt(d.out);
  p.in_eb = h->in_eb; p.out_eb = h->out_eb; p.hl = h->hl;
  p.skip = (d.ftype != ACDSP_POLY_FOLD_ANTI && h->t_total == 0) ? 1 : 0;
  {  // exact-accumulation class (see polyintr_acc_fast): the `fold` needs one more integer bit than IN_TYPE
    const int fi = p.in.F, fc = p.cf.F, fa = p.acc.F;
    p.lossless_shift = fa - fi - fc;
    p.lossless = !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && d.in.S && d.acc.S && d.acc.O == ACDSP_WRAP && p.lossless_shift >= 0 && p.lossless_shift < 64 && fa >= fi &&
                 d.acc.I >= d.in.I + 1 && (d.acc.W <= 63 || h->acc64_ok) && (d.in.O == ACDSP_WRAP || d.in.O == ACDSP_SAT || d.in.O == ACDSP_SAT_SYM || d.in.O == ACDSP_SAT_ZERO);
  }
  p.in_stride = in_stride; p.out_stride = out_stride; p.n = n_in; p.n_out = no;
  p.x = d_in; p.y = d_out; p.hist = h->d_hist[h->cur];
  p.coeffs = h->d_coeffs; p.sign = h->d_sign; p.corr = h->d_corr; p.saved = h->d_saved[h->cur];
  p.o_begin = 0; p.o_end = no;
  hipError_t e = hipSuccess;
  // Complete steps of 32 input slots go to the matrix-core kernel; the head (history, the saved sums of the previous call)
  // and the ragged tail stay on the VALU kernels.
  int64_t o_a = 0, o_b = 0;   // outputs [o_a, o_b) are produced by fir_up
  h->last_path = p.lossless ? ACDSP_PATH_LOSSLESS64 : ACDSP_PATH_GENERIC;
  if (h->up_ok && p.lossless) {
    const int L = d.ifac;
    const int64_t out_off = -(int64_t)p.skip * L, slot_a = h->up_plan.hs;
    const int64_t n_steps = (n_in / 16 - slot_a) / 32;
    // the tile stores need dword alignment only (gfx950 serves dword-aligned multi-dword stores): IF = 2 into 2-byte containers starts its
    // first call one input's outputs = 4 bytes into the 8-byte grid
    const int64_t oal = h->out_eb >= 8 ? 8 : 4;
    const bool aligned = ((uintptr_t)d_in % 16 == 0) && ((in_stride * h->in_eb) % 16 == 0) && ((uintptr_t)d_out % oal == 0) &&
                         ((out_stride * h->out_eb) % oal == 0) && ((out_off * h->out_eb) % oal == 0);
    if (aligned && n_steps > 0) {
      FirParams k;
      memset(&k, 0, sizeof k);
      k.n_ch = d.n_channels; k.in = p.in; k.cf = p.cf; k.acc = p.acc; k.out = p.out; k.in_eb = h->in_eb; k.out_eb = h->out_eb;
      k.lossless_shift = p.lossless_shift; k.in_stride = in_stride; k.out_stride = out_stride; k.n = n_in; k.x = d_in; k.y = d_out;
      e = launch_fir_up(k, h->up_plan, h->up_px, h->d_upfrag, h->d_upcorr, 0, 0, 0, h->up_shmask, h->up_max_abs, slot_a, n_steps, out_off, s);
      if (e == hipSuccess) {
        o_a = 16 * slot_a * L + out_off; o_b = 16 * (slot_a + 32 * n_steps) * L + out_off;
        h->last_path = ACDSP_PATH_MFMA_GEN;
      } else if (e != hipErrorNotSupported) {
        return fail(ACDSP_EHIP, "poly_intr matrix-core kernel launch failed: %s", hipGetErrorString(e));
      }
    }
  }
  if (o_b > o_a) {
    p.o_begin = 0; p.o_end = o_a;
    e = launch_polyintr(p, nullptr, s);
    if (e == hipSuccess) { p.o_begin = o_b; p.o_end = no; e = launch_polyintr(p, h->d_saved[h->cur ^ 1], s); }
  } else {
    e = launch_polyintr(p, h->d_saved[h->cur ^ 1], s);
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "poly_intr kernel launch failed: %s", hipGetErrorString(e)); }
  FirParams k;
  memset(&k, 0, sizeof k);
  k.n_ch = d.n_channels; k.in = p.in; k.in_eb = h->in_eb; k.hl = h->hl; k.in_stride = in_stride; k.n = n_in; k.x = d_in; k.hist = p.hist;
  e = launch_fir_hist_update(k, h->d_hist[h->cur ^ 1], s);   // (always the other buffer: the saved sums flip with it)
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "poly_intr state kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur ^= 1;
  h->t_total += n_in;
  return ACDSP_OK;
}

int32_t acdsp_polyintr_run_host(acdsp_polyintr_t h, const void *h_in, int64_t n_in, void *h_out, int64_t out_cap, int64_t *n_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (n_in < 0 || (n_in > 0 && !h_in)) { return fail(ACDSP_EINVAL, "poly_intr run_host: bad arguments"); }
  const int64_t no = acdsp_polyintr_out_count(h, n_in);
  if (n_out) { *n_out = no; }
  if (n_in == 0) { return ACDSP_OK; }
  if (no > 0 && (!h_out || out_cap < no)) { return fail(ACDSP_EINVAL, "poly_intr run_host: output buffer too small"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  const int64_t si = (n_in + 15) / 16 * 16, so = (no + 15) / 16 * 16 + 16;
  if ((rc = h->st.ensure((size_t)h->d.n_channels * si * h->in_eb, (size_t)h->d.n_channels * so * h->out_eb))) { return rc; }
  HIP_TRY(hipMemcpy2D(h->st.d_in, (size_t)si * h->in_eb, h_in, (size_t)n_in * h->in_eb, (size_t)n_in * h->in_eb,
                      (size_t)h->d.n_channels, hipMemcpyHostToDevice));
  int64_t got = 0;
  if ((rc = acdsp_polyintr_run(h, h->st.d_in, si, n_in, h->st.d_out, so, &got, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (got > 0) {
    HIP_TRY(hipMemcpy2D(h_out, (size_t)out_cap * h->out_eb, h->st.d_out, (size_t)so * h->out_eb, (size_t)got * h->out_eb,
                        (size_t)h->d.n_channels, hipMemcpyDeviceToHost));
  }
  return ACDSP_OK;
}

int32_t acdsp_polyintr_reset(acdsp_polyintr_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipMemset(h->d_hist[i], 0, (size_t)h->d.n_channels * h->hl * h->in_eb));
    HIP_TRY(hipMemset(h->d_saved[i], 0, (size_t)h->d.n_channels * h->d.ifac * sizeof(int64_t)));
  }
  h->t_total = 0;
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// integrate-and-dump: ac_intg_dump
// ---------------------------------------------------------------------------------------------
struct acdsp_intgdump {
  acdsp_intgdump_desc_t d;
  int in_eb, out_eb;
  int64_t *d_temp[2] = {nullptr, nullptr};
  int cur = 0;
  int64_t *d_blk = nullptr;     // [3][cap] off / rounds / out
  int32_t *d_chain = nullptr;   // [cap]
  int64_t blk_cap = 0;
  bool pending = false;         // the last call ended on a block that did not dump: temp[] is non-zero
  // block table of the last call: a stream that dumps on a fixed schedule passes the same n_sample[] every call, and then
  // neither the table is rebuilt nor uploaded and run() stays asynchronous (no stream synchronisation)
  std::vector<int64_t> last_ns;
  void *last_stream = nullptr;
  int64_t tbl_grp = 0, tbl_uni_rounds = 0;
  int32_t tbl_start = 0;
  Staging st;
};

namespace {
// per block: rounds consumed and whether it dumps (ac_intg_dump.h:138-146)
inline int64_t intg_rounds(int64_t n_sample, int ns, bool *dumps) {
  *dumps = n_sample >= 1 && n_sample <= ns;
  return *dumps ? n_sample : ns;
}
}  // namespace

extern "C" {

int32_t acdsp_intgdump_destroy(acdsp_intgdump_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  for (int i = 0; i < 2; i++) { if (h->d_temp[i]) { (void)hipFree(h->d_temp[i]); } }
  if (h->d_blk) { (void)hipFree(h->d_blk); }
  if (h->d_chain) { (void)hipFree(h->d_chain); }
  h->st.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_intgdump_create(const acdsp_intgdump_desc_t *desc, acdsp_intgdump_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  const acdsp_intgdump_desc_t &d = *desc;
  if (d.ns < 1 || d.ns > (1 << 24)) { return fail(ACDSP_EUNSUPPORTED, "NS=%d outside 1..2^24", d.ns); }
  if (d.chn < 1 || d.chn > 4096) { return fail(ACDSP_EUNSUPPORTED, "CHN=%d outside 1..4096", d.chn); }
  if (d.n_objects < 1) { return fail(ACDSP_EINVAL, "n_objects=%d must be positive", d.n_objects); }
  if (d.n_objects > 65535) { return fail(ACDSP_EUNSUPPORTED, "n_objects=%d outside 1..65535", d.n_objects); }
  int rc;
  if ((rc = check_fmt(d.in, "IN_TYPE")) || (rc = check_fmt(d.acc, "ACC_TYPE")) || (rc = check_fmt(d.out, "OUT_TYPE"))) { return rc; }
  if ((rc = check_device(d.device))) { return rc; }
  acdsp_intgdump *h = new acdsp_intgdump();
  h->d = d;
  h->in_eb = elem_bytes(d.in.W); h->out_eb = elem_bytes(d.out.W);
  hipError_t e = hipSuccess;
  const size_t tb = (size_t)d.n_objects * d.chn * sizeof(int64_t);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void **)&h->d_temp[i], tb);
    if (e == hipSuccess) { e = hipMemset(h->d_temp[i], 0, tb); }   // temp[i] = 0.0 (ac_intg_dump.h:86-89)
  }
  if (e != hipSuccess) { acdsp_intgdump_destroy(h); return fail(ACDSP_EHIP, "intg_dump state allocation failed: %s", hipGetErrorString(e)); }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_intgdump_counts(acdsp_intgdump_t h, const int64_t *n_sample, int64_t n_blocks, int64_t *n_in, int64_t *n_out) {
  if (!h || (n_blocks > 0 && !n_sample) || n_blocks < 0) { return fail(ACDSP_EINVAL, "intg_dump counts: bad arguments"); }
  int64_t rounds = 0, groups = 0;
  for (int64_t b = 0; b < n_blocks; b++) {
    bool dumps;
    rounds += intg_rounds(n_sample[b], h->d.ns, &dumps);
    groups += dumps ? 1 : 0;
  }
  if (n_in) { *n_in = rounds * h->d.chn; }
  if (n_out) { *n_out = groups * h->d.chn; }
  return ACDSP_OK;
}

int32_t acdsp_intgdump_run(acdsp_intgdump_t h, const void *d_in, int64_t in_stride, const int64_t *n_sample, int64_t n_blocks,
                           void *d_out, int64_t out_stride, int64_t *n_out, void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int64_t ni = 0, no = 0;
  int rc = acdsp_intgdump_counts(h, n_sample, n_blocks, &ni, &no);
  if (rc) { return rc; }
  if (n_out) { *n_out = no; }
  if (n_blocks == 0) { return ACDSP_OK; }
  if (n_blocks > (1 << 24)) { return fail(ACDSP_EUNSUPPORTED, "intg_dump run: more than 2^24 blocks in one call"); }
  if ((ni > 0 && (!d_in || in_stride < ni)) || (no > 0 && (!d_out || out_stride < no))) { return fail(ACDSP_EINVAL, "intg_dump run: buffers too small"); }
  const acdsp_intgdump_desc_t &d = h->d;
  if ((rc = check_device(d.device))) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  if (n_blocks > h->blk_cap) {
    HIP_TRY(hipStreamSynchronize(s));
    if (h->d_blk) { HIP_TRY(hipFree(h->d_blk)); h->d_blk = nullptr; }
    if (h->d_chain) { HIP_TRY(hipFree(h->d_chain)); h->d_chain = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->d_blk, (size_t)3 * n_blocks * sizeof(int64_t)));
    HIP_TRY(hipMalloc((void **)&h->d_chain, (size_t)n_blocks * sizeof(int32_t)));
    h->blk_cap = n_blocks;
    h->last_ns.clear();   // new device arrays: the table has to be uploaded again
  }
  const bool same_table = h->last_stream == stream && (int64_t)h->last_ns.size() == n_blocks &&
                          memcmp(h->last_ns.data(), n_sample, (size_t)n_blocks * sizeof(int64_t)) == 0;
  if (!same_table) {
    std::vector<int64_t> blk((size_t)3 * n_blocks);
    std::vector<int32_t> chain((size_t)n_blocks);
    int64_t off = 0, grp = 0;
    int32_t start = 0;
    for (int64_t b = 0; b < n_blocks; b++) {
      bool dumps;
      const int64_t r = intg_rounds(n_sample[b], d.ns, &dumps);
      blk[(size_t)b] = off; blk[(size_t)(n_blocks + b)] = r; blk[(size_t)(2 * n_blocks + b)] = dumps ? grp : -1;
      chain[(size_t)b] = start;
      off += r;
      if (dumps) { grp++; start = (int32_t)(b + 1); }
    }
    h->last_ns.clear();                 // (stays empty if the upload fails)
    // the previous table may still be read by a kernel on another stream: drain the device before overwriting it
    if (h->last_stream != stream) { HIP_TRY(hipDeviceSynchronize()); }
    HIP_TRY(hipMemcpyAsync(h->d_blk, blk.data(), blk.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_chain, chain.data(), chain.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // blk / chain are stack vectors
    h->tbl_grp = grp; h->tbl_start = start;
    h->tbl_uni_rounds = blk[(size_t)n_blocks];
    for (int64_t b = 1; b < n_blocks && h->tbl_uni_rounds > 0; b++) { if (blk[(size_t)(n_blocks + b)] != h->tbl_uni_rounds) { h->tbl_uni_rounds = 0; } }
    h->last_ns.assign(n_sample, n_sample + n_blocks);
    h->last_stream = stream;
  }
  const int64_t grp = h->tbl_grp;
  const int32_t start = h->tbl_start;
  IntgDumpParams p;
  memset(&p, 0, sizeof p);
  p.chn = d.chn; p.n_obj = d.n_objects; p.n_blocks = (int32_t)n_blocks;
  p.in = make_dfmt(d.in); p.acc = make_dfmt(d.acc); p.out = make_dfmt(d.out);
  p.in_eb = h->in_eb; p.out_eb = h->out_eb; p.in_stride = in_stride; p.out_stride = out_stride;
  p.lossless = d.acc.O == ACDSP_WRAP && p.acc.F >= p.in.F && p.acc.F - p.in.F < 64 - d.in.W;
  p.tile_ok = p.lossless && !h->pending && grp == n_blocks;
  if (p.tile_ok) { p.uni_rounds = h->tbl_uni_rounds; }
  p.x = d_in; p.y = d_out; p.temp = h->d_temp[h->cur];
  p.blk_off = h->d_blk; p.blk_rounds = h->d_blk + n_blocks; p.blk_out = h->d_blk + 2 * n_blocks; p.blk_chain = h->d_chain;
  hipError_t e = launch_intg_dump(p, h->d_temp[h->cur ^ 1], s);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "intg_dump kernel launch failed: %s", hipGetErrorString(e)); }
  h->cur ^= 1;
  h->pending = start != (int32_t)n_blocks;   // the call ended on blocks that did not dump: their sums sit in temp[]
  return ACDSP_OK;
}

int32_t acdsp_intgdump_run_host(acdsp_intgdump_t h, const void *h_in, const int64_t *n_sample, int64_t n_blocks, void *h_out,
                                int64_t out_cap, int64_t *n_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int64_t ni = 0, no = 0;
  int rc = acdsp_intgdump_counts(h, n_sample, n_blocks, &ni, &no);
  if (rc) { return rc; }
  if (n_out) { *n_out = no; }
  if (n_blocks == 0) { return ACDSP_OK; }
  if ((ni > 0 && !h_in) || (no > 0 && (!h_out || out_cap < no))) { return fail(ACDSP_EINVAL, "intg_dump run_host: bad buffers"); }
  if ((rc = check_device(h->d.device))) { return rc; }
  const int64_t si = ni > 0 ? ni : 1, so = no > 0 ? no : 1;
  if ((rc = h->st.ensure((size_t)h->d.n_objects * si * h->in_eb, (size_t)h->d.n_objects * so * h->out_eb))) { return rc; }
  if (ni > 0) {
    HIP_TRY(hipMemcpy2D(h->st.d_in, (size_t)si * h->in_eb, h_in, (size_t)ni * h->in_eb, (size_t)ni * h->in_eb, (size_t)h->d.n_objects,
                        hipMemcpyHostToDevice));
  }
  if ((rc = acdsp_intgdump_run(h, h->st.d_in, si, n_sample, n_blocks, h->st.d_out, so, nullptr, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (no > 0) {
    HIP_TRY(hipMemcpy2D(h_out, (size_t)out_cap * h->out_eb, h->st.d_out, (size_t)so * h->out_eb, (size_t)no * h->out_eb,
                        (size_t)h->d.n_objects, hipMemcpyDeviceToHost));
  }
  return ACDSP_OK;
}

int32_t acdsp_intgdump_reset(acdsp_intgdump_t h) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < 2; i++) { HIP_TRY(hipMemset(h->d_temp[i], 0, (size_t)h->d.n_objects * h->d.chn * sizeof(int64_t))); }
  h->pending = false;
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// moving average (second half)
// ---------------------------------------------------------------------------------------------
struct acdsp_mvavg {
  acdsp_mvavg_desc_t d;
  int in_eb, out_eb;
  bool coeffs_set = false;
  int64_t *d_coeffs = nullptr;
  std::vector<int64_t> h_coeffs;
  int last_path = 0;
  Staging st;
};

extern "C" {

int32_t acdsp_mvavg_create(const acdsp_mvavg_desc_t *desc, acdsp_mvavg_t *out) {
  if (!desc || !out) { return fail(ACDSP_EINVAL, "null argument"); }
  int rc;
  if ((rc = check_fmt(desc->in, "IN_TYPE")) || (rc = check_fmt(desc->coeff, "COEFF_TYPE")) || (rc = check_fmt(desc->acc, "ACC_TYPE")) ||
      (rc = check_fmt(desc->out, "OUT_TYPE"))) {
    return rc;
  }
  if (desc->taps < 1 || desc->taps > 1025) { return fail(ACDSP_EUNSUPPORTED, "mv_avg: TAPS=%d outside 1..1025", desc->taps); }
  if (!(desc->taps & 1)) { return fail(ACDSP_EUNSUPPORTED, "mv_avg: even TAPS: the reference's MAC loop reads coeffs[TAPS] (ac_mv_avg.h:117-119)"); }
  if (desc->win_mode < ACDSP_WIN_PLAIN || desc->win_mode > ACDSP_WIN_CLIP) { return fail(ACDSP_EINVAL, "mv_avg: bad window mode %d", desc->win_mode); }
  if (desc->max_sample < 1) { return fail(ACDSP_EINVAL, "mv_avg: MAX_SAMPLE must be >= 1"); }
  if (desc->n_objects < 1) { return fail(ACDSP_EINVAL, "mv_avg: n_objects must be >= 1"); }
  // 128-bit exact intermediates: ACC x COEFF product aligned with the accumulator
  const int fc = desc->coeff.W - desc->coeff.I;
  if (desc->acc.W + desc->coeff.W + 2 + (fc < 0 ? -fc : 0) > 125) { return fail(ACDSP_EUNSUPPORTED, "mv_avg: type combination needs more than 128-bit intermediates"); }
  if ((rc = check_device(desc->device))) { return rc; }
  acdsp_mvavg *h = new acdsp_mvavg();
  h->d = *desc;
  h->in_eb = elem_bytes(desc->in.W);
  h->out_eb = elem_bytes(desc->out.W);
  if (hipMalloc((void **)&h->d_coeffs, (size_t)desc->taps * sizeof(int64_t)) != hipSuccess) {
    delete h;
    return fail(ACDSP_EHIP, "mv_avg: coefficient allocation failed");
  }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_mvavg_destroy(acdsp_mvavg_t h) {
  if (!h) { return ACDSP_OK; }
  (void)hipSetDevice(h->d.device);
  if (h->d_coeffs) { (void)hipFree(h->d_coeffs); }
  h->st.destroy();
  delete h;
  return ACDSP_OK;
}

int32_t acdsp_mvavg_set_coeffs(acdsp_mvavg_t h, const int64_t *coeffs) {
  if (!h || !coeffs) { return fail(ACDSP_EINVAL, "null argument"); }
  const DFmt cf = make_dfmt(h->d.coeff);
  for (int i = 0; i < h->d.taps; i++) {
    if (coeffs[i] < cf.lo || coeffs[i] > cf.hi) { return fail(ACDSP_EINVAL, "coefficient %d = %lld is not a COEFF_TYPE raw word", i, (long long)coeffs[i]); }
  }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->d_coeffs, coeffs, (size_t)h->d.taps * sizeof(int64_t), hipMemcpyHostToDevice));
  h->h_coeffs.assign(coeffs, coeffs + h->d.taps);
  h->coeffs_set = true;
  return ACDSP_OK;
}

int32_t acdsp_mvavg_path(acdsp_mvavg_t h) { return h ? h->last_path : -1; }

int64_t acdsp_mvavg_out_per_frame(acdsp_mvavg_t h, int64_t n_sample) {
  if (!h || n_sample < 1 || n_sample > h->d.max_sample) { return -1; }
  if (h->d.win_mode == ACDSP_WIN_PLAIN) { return n_sample >= h->d.taps ? n_sample - h->d.taps + 1 : 0; }
  return n_sample;
}

int32_t acdsp_mvavg_run(acdsp_mvavg_t h, const void *d_in, int64_t in_stride, int64_t n_sample, int64_t n_frames, void *d_out,
                        int64_t out_stride, int64_t *n_out, void *stream) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "mvavg_run before acdsp_mvavg_set_coeffs"); }
  const int64_t opf = acdsp_mvavg_out_per_frame(h, n_sample);
  if (opf < 0) { return fail(ACDSP_EINVAL, "mv_avg: n_sample=%lld outside 1..MAX_SAMPLE=%d (the reference's frame loop would lose alignment)", (long long)n_sample, h->d.max_sample); }
  if (n_frames < 0 || n_frames > (int64_t(1) << 40) / n_sample) { return fail(ACDSP_EINVAL, "mv_avg: bad frame count"); }
  const int64_t no = opf * n_frames;
  if (n_out) { *n_out = no; }
  if (n_frames == 0) { return ACDSP_OK; }
  if (!d_in || in_stride < n_sample * n_frames) { return fail(ACDSP_EINVAL, "mv_avg run: bad input arguments"); }
  if (no > 0 && (!d_out || out_stride < no)) { return fail(ACDSP_EINVAL, "mv_avg run: output buffer too small for %lld outputs", (long long)no); }
  const acdsp_mvavg_desc_t &d = h->d;
  int rc = check_device(d.device);
  if (rc) { return rc; }
  MvAvgParams p;
  memset(&p, 0, sizeof p);
  p.taps = d.taps; p.win_mode = d.win_mode; p.n_obj = d.n_objects;
  p.in = make_dfmt(d.in); p.cf = make_dfmt(d.coeff); p.acc = make_dfmt(d.acc); p.out = make_dfmt(d.out);
  p.in_eb = h->in_eb; p.out_eb = h->out_eb;
  p.force_generic = (d.flags & ACDSP_FLAG_FORCE_GENERIC) != 0;
  p.fast = !p.force_generic && d.acc.O == ACDSP_WRAP && (d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && p.cf.F >= 0 &&
           p.cf.F < 62 && d.acc.W + d.coeff.W <= 62;
  p.n_sample = n_sample; p.n_frames = n_frames; p.out_per_frame = opf; p.in_stride = in_stride; p.out_stride = out_stride;
  p.x = d_in; p.y = d_out; p.coeffs = h->d_coeffs; p.h_coeffs = h->h_coeffs.data();
  hipError_t e = launch_mv_avg(p, (hipStream_t)stream, &h->last_path);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "mv_avg kernel launch failed: %s", hipGetErrorString(e)); }
  return ACDSP_OK;
}

int32_t acdsp_mvavg_run_host(acdsp_mvavg_t h, const void *h_in, int64_t n_sample, int64_t n_frames, void *h_out, int64_t out_cap,
                             int64_t *n_out) {
  if (!h) { return fail(ACDSP_EINVAL, "null handle"); }
  const int64_t opf = acdsp_mvavg_out_per_frame(h, n_sample);
  if (opf < 0 || n_frames < 0) { return fail(ACDSP_EINVAL, "mv_avg run_host: bad n_sample / n_frames"); }
  const int64_t ni = n_sample * n_frames, no = opf * n_frames;
  if (n_out) { *n_out = no; }
  if (n_frames == 0) { return ACDSP_OK; }
  if (!h_in || (no > 0 && (!h_out || out_cap < no))) { return fail(ACDSP_EINVAL, "mv_avg run_host: bad buffers"); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  const size_t nobj = (size_t)h->d.n_objects;
  if ((rc = h->st.ensure(nobj * ni * h->in_eb, nobj * (no > 0 ? no : 1) * h->out_eb))) { return rc; }
  HIP_TRY(hipMemcpy(h->st.d_in, h_in, nobj * ni * h->in_eb, hipMemcpyHostToDevice));
  if ((rc = acdsp_mvavg_run(h, h->st.d_in, ni, n_sample, n_frames, h->st.d_out, no > 0 ? no : 1, nullptr, nullptr))) { return rc; }
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (no > 0) {
    HIP_TRY(hipMemcpy2D(h_out, (size_t)out_cap * h->out_eb, h->st.d_out, (size_t)no * h->out_eb, (size_t)no * h->out_eb, nobj, hipMemcpyDeviceToHost));
  }
  return ACDSP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// state save / restore (SURVEY 8b export list, section 5 checkpoint / resume hook)
// ---------------------------------------------------------------------------------------------
namespace {

// 64-byte little-endian header in front of the raw state words.
struct StateHdr {
  char magic[8];            // "ACDSPST1"
  uint32_t version;         // 1
  uint32_t kind;            // 1: FIR input history, 2: FIR reg_trans partial sums, 3: CIC input history + input count,
                            // 4: fused DDC input history + input count, 5: two-kernel DDC (CIC blob + FIR blob follow)
  uint32_t n_channels;
  uint32_t elem_bytes;      // bytes per state word (IN container; reg_trans: 8, or 16 for an ACC_TYPE wider than 64 bits)
  uint64_t per_channel;     // state words per channel
  int64_t t_total;          // CIC: inputs consumed so far (decimation / interpolation phase); 0 for FIR
  uint32_t p0, p1, p2, p3;  // FIR: n_taps, ftype, class, 0;  CIC: R, M, N, interp
  uint64_t reserved;
};
static_assert(sizeof(StateHdr) == 64, "state header is 64 bytes");
const char kStateMagic[8] = {'A', 'C', 'D', 'S', 'P', 'S', 'T', '1'};

StateHdr fir_state_hdr(const acdsp_fir *h) {
  StateHdr s;
  memset(&s, 0, sizeof s);
  memcpy(s.magic, kStateMagic, 8);
  s.version = 1;
  s.kind = h->use_rt ? 2 : 1;
  s.n_channels = (uint32_t)h->d.n_channels;
  s.elem_bytes = h->use_rt ? (uint32_t)h->rt_eb : (uint32_t)h->in_eb;
  s.per_channel = h->use_rt ? (uint64_t)h->d.n_taps : (uint64_t)h->hl;
  s.p0 = (uint32_t)h->d.n_taps; s.p1 = (uint32_t)h->d.ftype; s.p2 = (uint32_t)h->d.kind;
  return s;
}
StateHdr cic_state_hdr(const acdsp_cic *h) {
  StateHdr s;
  memset(&s, 0, sizeof s);
  memcpy(s.magic, kStateMagic, 8);
  s.version = 1;
  s.kind = 3;
  s.n_channels = (uint32_t)h->d.n_channels;
  s.elem_bytes = (uint32_t)h->in_eb;
  s.per_channel = (uint64_t)h->hl;
  s.t_total = h->t_total;
  s.p0 = (uint32_t)h->d.R; s.p1 = (uint32_t)h->d.M; s.p2 = (uint32_t)h->d.N; s.p3 = (uint32_t)h->d.interp;
  return s;
}
uint64_t state_payload(const StateHdr &s) { return (uint64_t)s.n_channels * s.per_channel * s.elem_bytes; }

// everything but t_total must agree between the blob and the handle it is loaded into
bool state_compatible(const StateHdr &blob, const StateHdr &mine) {
  return memcmp(blob.magic, kStateMagic, 8) == 0 && blob.version == 1 && blob.kind == mine.kind && blob.n_channels == mine.n_channels &&
         blob.elem_bytes == mine.elem_bytes && blob.per_channel == mine.per_channel && blob.p0 == mine.p0 && blob.p1 == mine.p1 &&
         blob.p2 == mine.p2 && blob.p3 == mine.p3;
}

}  // namespace

extern "C" {

int64_t acdsp_fir_state_size(acdsp_fir_t h) { return h ? (int64_t)(sizeof(StateHdr) + state_payload(fir_state_hdr(h))) : -1; }

int32_t acdsp_fir_state_get(acdsp_fir_t h, void *buf, uint64_t cap_bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr s = fir_state_hdr(h);
  const uint64_t pay = state_payload(s);
  if (cap_bytes < sizeof s + pay) { return fail(ACDSP_EINVAL, "fir_state_get: buffer of %llu bytes, state needs %llu", (unsigned long long)cap_bytes, (unsigned long long)(sizeof s + pay)); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());   // run() is asynchronous: the state of the last call must have landed
  if (h->rt_hybrid && !h->rt_valid && (rc = fir_rt_from_hist(h))) { return rc; }
  memcpy(buf, &s, sizeof s);
  HIP_TRY(hipMemcpy((char *)buf + sizeof s, h->use_rt ? (const void *)h->d_rt[h->rt_hybrid ? h->cur_rt : h->cur] : h->d_hist[h->cur], pay, hipMemcpyDeviceToHost));
  return ACDSP_OK;
}

int32_t acdsp_fir_state_set(acdsp_fir_t h, const void *buf, uint64_t bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr mine = fir_state_hdr(h);
  StateHdr s;
  if (bytes < sizeof s) { return fail(ACDSP_EINVAL, "fir_state_set: blob shorter than its header"); }
  memcpy(&s, buf, sizeof s);
  // An input-history blob (kind 1) of another history LENGTH is still this filter's state as long as it covers the taps: the handle's
  // length is n_taps - 1 rounded up for the kernels' windows (and changed between builds: a padded MFMA plan reaches a block further
  // back), the samples beyond n_taps - 1 only ever meet zero coefficients.  The newest min(blob, mine) samples are kept, older ones zero.
  StateHdr same_len = s;
  same_len.per_channel = mine.per_channel;
  const bool relen = s.kind == 1 && mine.kind == 1 && s.per_channel != mine.per_channel && s.per_channel + 1 >= (uint64_t)h->d.n_taps &&
                     s.per_channel <= (uint64_t(1) << 20) && state_compatible(same_len, mine) && bytes == sizeof s + state_payload(s);
  if (!relen && (!state_compatible(s, mine) || bytes != sizeof s + state_payload(mine))) {
    return fail(ACDSP_EINVAL, "fir_state_set: blob does not belong to a filter of this class / tap count / channel count");
  }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  if (relen) {
    const size_t eb = mine.elem_bytes, pm = (size_t)mine.per_channel, pb = (size_t)s.per_channel, keep = pm < pb ? pm : pb;
    std::vector<unsigned char> img((size_t)mine.n_channels * pm * eb, 0);
    const unsigned char *src = (const unsigned char *)buf + sizeof s;
    for (size_t c = 0; c < mine.n_channels; c++) { memcpy(&img[(c * pm + (pm - keep)) * eb], src + (c * pb + (pb - keep)) * eb, keep * eb); }
    HIP_TRY(hipMemcpy(h->d_hist[h->cur], img.data(), img.size(), hipMemcpyHostToDevice));
    return ACDSP_OK;
  }
  HIP_TRY(hipMemcpy(h->use_rt ? (void *)h->d_rt[h->rt_hybrid ? h->cur_rt : h->cur] : h->d_hist[h->cur], (const char *)buf + sizeof s, state_payload(mine), hipMemcpyHostToDevice));
  if (h->rt_hybrid) {   // partial sums of unknown coefficients and samples: the next n_taps - 1 outputs come from them, the history starts empty
    HIP_TRY(hipMemset(h->d_hist[h->cur], 0, (size_t)h->d.n_channels * h->hl * h->in_eb));
    h->rt_valid = true; h->rt_since = 0;
  }
  return ACDSP_OK;
}

int64_t acdsp_cic_state_size(acdsp_cic_t h) { return h ? (int64_t)(sizeof(StateHdr) + state_payload(cic_state_hdr(h))) : -1; }

int32_t acdsp_cic_state_get(acdsp_cic_t h, void *buf, uint64_t cap_bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr s = cic_state_hdr(h);
  const uint64_t pay = state_payload(s);
  if (cap_bytes < sizeof s + pay) { return fail(ACDSP_EINVAL, "cic_state_get: buffer of %llu bytes, state needs %llu", (unsigned long long)cap_bytes, (unsigned long long)(sizeof s + pay)); }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  memcpy(buf, &s, sizeof s);
  HIP_TRY(hipMemcpy((char *)buf + sizeof s, h->d_hist[h->cur], pay, hipMemcpyDeviceToHost));
  return ACDSP_OK;
}

int32_t acdsp_cic_state_set(acdsp_cic_t h, const void *buf, uint64_t bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr mine = cic_state_hdr(h);
  StateHdr s;
  if (bytes < sizeof s) { return fail(ACDSP_EINVAL, "cic_state_set: blob shorter than its header"); }
  memcpy(&s, buf, sizeof s);
  if (!state_compatible(s, mine) || bytes != sizeof s + state_payload(mine) || s.t_total < 0) {
    return fail(ACDSP_EINVAL, "cic_state_set: blob does not belong to a CIC filter of these parameters / channel count");
  }
  int rc = check_device(h->d.device);
  if (rc) { return rc; }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->d_hist[h->cur], (const char *)buf + sizeof s, state_payload(mine), hipMemcpyHostToDevice));
  h->t_total = s.t_total;
  return ACDSP_OK;
}

// DDC cascade.  Fused mode: the input history + input count are the whole state (stage B's window is recomputed from it):
// one kind-4 blob.  Two-kernel mode: a kind-5 header followed by the stage blobs (CIC, then FIR).
static StateHdr ddc_state_hdr(const acdsp_ddc *h) {
  StateHdr s;
  memset(&s, 0, sizeof s);
  memcpy(s.magic, kStateMagic, 8);
  s.version = 1;
  s.kind = h->fused ? 4 : 5;
  s.n_channels = (uint32_t)h->cic->d.n_channels;
  s.p0 = (uint32_t)h->cic->d.R; s.p1 = (uint32_t)h->cic->d.M; s.p2 = (uint32_t)h->cic->d.N; s.p3 = (uint32_t)h->fir->d.n_taps;
  if (h->fused) { s.elem_bytes = (uint32_t)h->cic->in_eb; s.per_channel = (uint64_t)h->hl; s.t_total = h->t_total; }
  else {   // payload = the two stage blobs; per_channel x elem_bytes x n_channels must give its size
    s.elem_bytes = 1; s.n_channels = 1;
    s.per_channel = (uint64_t)(acdsp_cic_state_size(h->cic) + acdsp_fir_state_size(h->fir));
    s.reserved = (uint64_t)h->cic->d.n_channels;
  }
  return s;
}

int64_t acdsp_ddc_state_size(acdsp_ddc_t h) { return h ? (int64_t)(sizeof(StateHdr) + state_payload(ddc_state_hdr(h))) : -1; }

int32_t acdsp_ddc_state_get(acdsp_ddc_t h, void *buf, uint64_t cap_bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr s = ddc_state_hdr(h);
  const uint64_t pay = state_payload(s);
  if (cap_bytes < sizeof s + pay) { return fail(ACDSP_EINVAL, "ddc_state_get: buffer of %llu bytes, state needs %llu", (unsigned long long)cap_bytes, (unsigned long long)(sizeof s + pay)); }
  int rc = check_device(h->cic->d.device);
  if (rc) { return rc; }
  memcpy(buf, &s, sizeof s);
  if (h->fused) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy((char *)buf + sizeof s, h->d_hist[h->cur], pay, hipMemcpyDeviceToHost));
    return ACDSP_OK;
  }
  const uint64_t nc = (uint64_t)acdsp_cic_state_size(h->cic);
  if ((rc = acdsp_cic_state_get(h->cic, (char *)buf + sizeof s, nc))) { return rc; }
  return acdsp_fir_state_get(h->fir, (char *)buf + sizeof s + nc, pay - nc);
}

int32_t acdsp_ddc_state_set(acdsp_ddc_t h, const void *buf, uint64_t bytes) {
  if (!h || !buf) { return fail(ACDSP_EINVAL, "null argument"); }
  const StateHdr mine = ddc_state_hdr(h);
  StateHdr s;
  if (bytes < sizeof s) { return fail(ACDSP_EINVAL, "ddc_state_set: blob shorter than its header"); }
  memcpy(&s, buf, sizeof s);
  if (!state_compatible(s, mine) || s.reserved != mine.reserved || bytes != sizeof s + state_payload(mine) || s.t_total < 0) {
    return fail(ACDSP_EINVAL, "ddc_state_set: blob does not belong to a cascade of these parameters / channel count / kernel mode");
  }
  int rc = check_device(h->cic->d.device);
  if (rc) { return rc; }
  if (h->fused) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->d_hist[h->cur], (const char *)buf + sizeof s, state_payload(mine), hipMemcpyHostToDevice));
    h->t_total = s.t_total;
    return ACDSP_OK;
  }
  const uint64_t nc = (uint64_t)acdsp_cic_state_size(h->cic);
  if ((rc = acdsp_cic_state_set(h->cic, (const char *)buf + sizeof s, nc))) { return rc; }
  return acdsp_fir_state_set(h->fir, (const char *)buf + sizeof s + nc, state_payload(mine) - nc);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// raw-integer stream files (second half): the [channel][time] layout of section 3 of DESIGN.md on disk /
// on the wire -- a 64-byte little-endian header followed by n_channels rows of `stride` containers.  Host-side only.
// ---------------------------------------------------------------------------------------------
extern "C" {

static const char kStreamMagic[8] = {'A', 'C', 'D', 'S', 'P', 'R', 'A', 'W'};
static_assert(sizeof(acdsp_stream_hdr_t) == 64, "stream header is 64 bytes on disk");

// n_channels * stride * elem_bytes of a header, or false when the product leaves 64 bits (a header from an untrusted file
// could otherwise wrap to a small size that passes the capacity check)
static bool stream_payload_bytes(const acdsp_stream_hdr_t &h, uint64_t *bytes) {
  uint64_t rows = 0;
  return !__builtin_mul_overflow(h.n_channels, h.stride, &rows) && !__builtin_mul_overflow(rows, (uint64_t)h.elem_bytes, bytes);
}

int32_t acdsp_stream_write(const char *path, const acdsp_stream_hdr_t *hdr, const void *data) {
  if (!path || !hdr || (!data && hdr->n_channels * hdr->stride > 0)) { return fail(ACDSP_EINVAL, "stream_write: null argument"); }
  if (hdr->elem_bytes != (uint32_t)elem_bytes(hdr->fmt.W) || hdr->stride < hdr->n_samples) {
    return fail(ACDSP_EINVAL, "stream_write: elem_bytes must be acdsp_elem_bytes(W) and stride >= n_samples");
  }
  int rc = check_fmt(hdr->fmt, "stream format");
  if (rc) { return rc; }
  uint64_t bytes = 0;
  if (!stream_payload_bytes(*hdr, &bytes)) { return fail(ACDSP_EINVAL, "stream_write: n_channels * stride * elem_bytes overflows"); }
  FILE *f = fopen(path, "wb");
  if (!f) { return fail(ACDSP_EINVAL, "stream_write: cannot open %s", path); }
  acdsp_stream_hdr_t h = *hdr;
  memcpy(h.magic, kStreamMagic, 8);
  h.version = 1; h.reserved = 0;
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (bytes == 0 || fwrite(data, 1, bytes, f) == bytes);
  fclose(f);
  return ok ? ACDSP_OK : fail(ACDSP_EINVAL, "stream_write: short write to %s", path);
}

int32_t acdsp_stream_read_header(const char *path, acdsp_stream_hdr_t *hdr) {
  if (!path || !hdr) { return fail(ACDSP_EINVAL, "stream_read_header: null argument"); }
  FILE *f = fopen(path, "rb");
  if (!f) { return fail(ACDSP_EINVAL, "stream_read_header: cannot open %s", path); }
  const bool ok = fread(hdr, sizeof *hdr, 1, f) == 1;
  long fsize = -1;
  if (ok && fseek(f, 0, SEEK_END) == 0) { fsize = ftell(f); }
  fclose(f);
  if (!ok || memcmp(hdr->magic, kStreamMagic, 8) != 0 || hdr->version != 1) { return fail(ACDSP_EINVAL, "%s is not an ACDSPRAW v1 stream", path); }
  uint64_t bytes = 0;
  if (check_fmt(hdr->fmt, "stream format") != ACDSP_OK || hdr->elem_bytes != (uint32_t)elem_bytes(hdr->fmt.W) || hdr->stride < hdr->n_samples ||
      !stream_payload_bytes(*hdr, &bytes)) {
    return fail(ACDSP_EINVAL, "%s: inconsistent stream header", path);
  }
  if (fsize < 0 || (uint64_t)fsize < sizeof *hdr || (uint64_t)fsize - sizeof *hdr < bytes) {
    return fail(ACDSP_EINVAL, "%s: header announces %llu payload bytes, the file holds fewer", path, (unsigned long long)bytes);
  }
  return ACDSP_OK;
}

int32_t acdsp_stream_read(const char *path, void *data, uint64_t cap_bytes) {
  acdsp_stream_hdr_t h;
  int rc = acdsp_stream_read_header(path, &h);
  if (rc) { return rc; }
  uint64_t bytes = 0;
  if (!stream_payload_bytes(h, &bytes)) { return fail(ACDSP_EINVAL, "stream_read: inconsistent stream header"); }
  if (bytes > cap_bytes || (bytes > 0 && !data)) { return fail(ACDSP_EINVAL, "stream_read: buffer of %llu bytes for %llu", (unsigned long long)cap_bytes, (unsigned long long)bytes); }
  FILE *f = fopen(path, "rb");
  if (!f) { return fail(ACDSP_EINVAL, "stream_read: cannot open %s", path); }
  const bool ok = fseek(f, (long)sizeof h, SEEK_SET) == 0 && (bytes == 0 || fread(data, 1, (size_t)bytes, f) == bytes);
  fclose(f);
  return ok ? ACDSP_OK : fail(ACDSP_EINVAL, "stream_read: %s is truncated", path);
}

}  // extern "C"
