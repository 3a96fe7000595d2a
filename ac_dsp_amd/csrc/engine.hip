// engine.hip -- the C ABI of libacdsp.so (declared in include/acdsp.h): what every operator family shares -- error state, device check,
// device memory helpers, the stimulus generator, the diagnostics of bench.py -- plus the state blobs and the stream files.
// The families live in engine_fir.hip, engine_cic.hip, engine_ddc.hip, engine_poly.hip (poly_dec, poly_intr) and engine_misc.hip
// (intg_dump, mv_avg); engine_common.hpp holds the handle structs and helpers they share.
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

namespace {
thread_local std::string g_err;
}  // namespace

namespace acdsp {
namespace eng {

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

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

}  // namespace eng
}  // namespace acdsp

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

extern "C" int32_t acdsp_diag_shader_clock_mhz(int32_t device, void *stream, float *mhz) {
  if (!mhz) { return fail(ACDSP_EINVAL, "diag_shader_clock: null output"); }
  int rc = check_device(device);
  if (rc) { return rc; }
  // one 32-byte result buffer per device and calling thread (a buffer of another device would fault without peer access: advisor, round 5);
  // they live as long as the process -- a measurement helper, 16 devices at most
  static thread_local float *d_bufs[16] = {nullptr};
  if (device < 0 || device >= 16) { return fail(ACDSP_EINVAL, "diag_shader_clock: device %d outside 0..15", device); }
  if (!d_bufs[device]) { HIP_TRY(hipMalloc((void **)&d_bufs[device], 8 * sizeof(float))); }
  float *d_buf = d_bufs[device];
  hipStream_t s = (hipStream_t)stream;
  const hipError_t e = launch_diag_clock(d_buf, 8, s);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "diag clock launch failed: %s", hipGetErrorString(e)); }
  float h[8];
  HIP_TRY(hipMemcpyAsync(h, d_buf, sizeof h, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  float sum = 0;
  for (int i = 0; i < 8; i++) { sum += h[i]; }
  *mhz = sum / 8;
  return ACDSP_OK;
}

extern "C" int32_t acdsp_diag_mix_ms(int32_t device, const void *d_src, uint64_t src_bytes, void *d_dst, uint64_t dst_bytes, int32_t warmup, int32_t reps, void *stream,
                                     float *ms_avg) {
  if (!d_src || !d_dst || src_bytes < 1024 || dst_bytes < 1024 || ((uintptr_t)d_src | (uintptr_t)d_dst) % 16) { return fail(ACDSP_EINVAL, "diag_mix: 16-byte aligned blocks of at least 1 KB"); }
  int rc = check_device(device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  return time_launches([&] { return launch_diag_mix(d_src, (int64_t)src_bytes, d_dst, (int64_t)dst_bytes, s); }, warmup, reps, s, ms_avg);
}

// Allocation with a placement probe.  The HBM-bound operators run 3 - 8 % apart (config 3: 14.4 / 15.3 ms) on different (input, output)
// allocation pairs -- same bytes, same TLB and L2 behaviour, more DRAM credit stalls on the slow pair (profiles/r3_placement_modes.txt) -- and
// the level belongs to the pair of allocations, not to an offset inside one.  A caller that allocates its output AFTER its input can shop:
// n_candidates blocks are allocated, each is timed with the bare mixed stream of acdsp_diag_mix_ms against the partner block (partner_reads != 0:
// the partner is the stream that is read, the new block the one written; 0: the other way round), the fastest is kept and the others freed.
extern "C" int32_t acdsp_dev_alloc_paired(int32_t device, uint64_t bytes, const void *d_partner, uint64_t partner_bytes, int32_t partner_reads,
                                          int32_t n_candidates, void **d_ptr, float *probe_ms) {
  if (!d_ptr || bytes == 0) { return fail(ACDSP_EINVAL, "dev_alloc_paired: null output or zero size"); }
  if (n_candidates < 1) { n_candidates = 1; }
  if (n_candidates > 16) { n_candidates = 16; }
  int rc = check_device(device);
  if (rc) { return rc; }
  const bool probe = d_partner && partner_bytes >= (1u << 20) && bytes >= (1u << 20) && n_candidates > 1 && (uintptr_t)d_partner % 16 == 0;
  if (!probe) {
    HIP_TRY(hipMalloc(d_ptr, bytes));
    if (probe_ms) { probe_ms[0] = 0.f; }
    return ACDSP_OK;
  }
  void *cand[16] = {nullptr};
  float ms[16];
  int n = 0, best = 0;
  for (; n < n_candidates; n++) {
    if (hipMalloc(&cand[n], bytes) != hipSuccess) { (void)hipGetLastError(); cand[n] = nullptr; break; }   // out of memory: shop among what fitted
  }
  if (n == 0) { return fail(ACDSP_EHIP, "dev_alloc_paired: hipMalloc of %llu bytes failed", (unsigned long long)bytes); }
  // the probe walks BOTH blocks end to end (a 4 GB prefix does not show the level: profiles/r6_placement.txt)
  const uint64_t pb = partner_bytes / 1024 * 1024, nb = bytes / 1024 * 1024;
  for (int pass = 0; pass < 2; pass++) {          // two passes: a candidate's time does not depend on its turn
    for (int i = 0; i < n; i++) {
      float t = 0;
      rc = partner_reads ? acdsp_diag_mix_ms(device, d_partner, pb, cand[i], nb, 2, 5, nullptr, &t) : acdsp_diag_mix_ms(device, cand[i], nb, const_cast<void *>(d_partner), pb, 2, 5, nullptr, &t);
      if (rc) { for (int k = 0; k < n; k++) { (void)hipFree(cand[k]); } return rc; }
      ms[i] = pass == 0 ? t : (t < ms[i] ? t : ms[i]);
    }
  }
  for (int i = 1; i < n; i++) { if (ms[i] < ms[best]) { best = i; } }
  for (int i = 0; i < n; i++) { if (i != best) { (void)hipFree(cand[i]); } }
  if (probe_ms) { for (int i = 0; i < n_candidates; i++) { probe_ms[i] = i < n ? ms[i] : 0.f; } }
  *d_ptr = cand[best];
  return ACDSP_OK;
}

// ... and with the caller's OWN call as the probe: the bare stream ranks candidate blocks like the operator only where the thin stream is
// thin enough (16 : 1 and beyond: a 12 % level on a <16,1> R = 64 decimator found by both; config 3's 4 : 1 pairs, 6 % apart, are invisible to
// it: profiles/r6_placement.txt).  trial(ctx, candidate) enqueues one call of the operator with the candidate as its output (or input) on the
// NULL stream and returns 0; every candidate is timed over `reps` calls behind one untimed call, twice round; the fastest is kept.
extern "C" int32_t acdsp_dev_alloc_shop(int32_t device, uint64_t bytes, int32_t n_candidates, int32_t (*trial)(void *ctx, void *d_candidate), void *ctx, int32_t reps,
                                        void **d_ptr, float *trial_ms) {
  if (!d_ptr || bytes == 0) { return fail(ACDSP_EINVAL, "dev_alloc_shop: null output or zero size"); }
  if (n_candidates < 1) { n_candidates = 1; }
  if (n_candidates > 16) { n_candidates = 16; }
  if (reps < 1) { reps = 1; }
  int rc = check_device(device);
  if (rc) { return rc; }
  if (!trial || n_candidates == 1) {
    HIP_TRY(hipMalloc(d_ptr, bytes));
    if (trial_ms) { trial_ms[0] = 0.f; }
    return ACDSP_OK;
  }
  void *cand[16] = {nullptr};
  float ms[16];
  int n = 0, best = 0;
  for (; n < n_candidates; n++) {
    if (hipMalloc(&cand[n], bytes) != hipSuccess) { (void)hipGetLastError(); cand[n] = nullptr; break; }
  }
  if (n == 0) { return fail(ACDSP_EHIP, "dev_alloc_shop: hipMalloc of %llu bytes failed", (unsigned long long)bytes); }
  auto free_all = [&](int keep) { for (int k = 0; k < n; k++) { if (k != keep) { (void)hipFree(cand[k]); } } };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { free_all(-1); return fail(ACDSP_EHIP, "dev_alloc_shop: event creation failed"); }
  for (int pass = 0; pass < 2 && rc == 0; pass++) {
    for (int i = 0; i < n && rc == 0; i++) {
      rc = trial(ctx, cand[i]);
      if (rc == 0) { (void)hipEventRecord(e0, nullptr); }
      for (int r = 0; r < reps && rc == 0; r++) { rc = trial(ctx, cand[i]); }
      float t = 0;
      if (rc == 0 && (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess)) { rc = ACDSP_EHIP; }
      t /= (float)reps;
      ms[i] = pass == 0 ? t : (t < ms[i] ? t : ms[i]);
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) { free_all(-1); return rc == ACDSP_EHIP ? fail(ACDSP_EHIP, "dev_alloc_shop: timing a trial failed") : rc; }
  for (int i = 1; i < n; i++) { if (ms[i] < ms[best]) { best = i; } }
  free_all(best);
  if (trial_ms) { for (int i = 0; i < n_candidates; i++) { trial_ms[i] = i < n ? ms[i] : 0.f; } }
  *d_ptr = cand[best];
  return ACDSP_OK;
}

// A operands with the statistics of the product's: Toeplitz fragments of the caller's set (fir_mfma_build_fragments: [2 planes][nb][64][4]
// dwords), four blocks of the low-byte plane spread over the taps and two non-zero blocks of the high-byte plane (the centre of the band)
static int envelope_fragments(const int64_t *coeffs, int32_t n_taps, int32_t mfma_per_step, std::vector<uint32_t> *six_out) {
  std::vector<uint32_t> &six = *six_out;
  six.assign((size_t)6 * 64 * 4, 0u);
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
  return ACDSP_OK;
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
  std::vector<uint32_t> six;
  if ((rc = envelope_fragments(coeffs, n_taps, mfma_per_step, &six))) { return rc; }
  uint32_t *d_frag = nullptr;
  HIP_TRY(hipMalloc((void **)&d_frag, six.size() * sizeof(uint32_t)));
  hipError_t ce = hipMemcpy(d_frag, six.data(), six.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (ce != hipSuccess) { (void)hipFree(d_frag); return fail(ACDSP_EHIP, "diag_fir_envelope: upload failed: %s", hipGetErrorString(ce)); }
  int out = time_launches([&] { return launch_diag_envelope(d_frag, d_x, d_y, (int64_t)bytes, mfma_per_step, mfma_hi_per_step, s); }, warmup, reps, s, ms_avg);
  (void)hipFree(d_frag);
  return out;
}

extern "C" int32_t acdsp_diag_fir_envelope_copygeom_ms(int32_t device, const int64_t *coeffs, int32_t n_taps, int32_t mfma_per_step, int32_t mfma_hi_per_step,
                                                       const void *d_x, void *d_y, uint64_t bytes, int32_t warmup, int32_t reps, void *stream, float *ms_avg) {
  if (!d_x || !d_y || bytes < 16 || bytes % 16 || ((uintptr_t)d_x | (uintptr_t)d_y) % 16) { return fail(ACDSP_EINVAL, "diag_fir_envelope_copygeom: 16-byte aligned buffers of a multiple of 16 bytes"); }
  if (mfma_per_step < 0 || mfma_hi_per_step < 0 || mfma_hi_per_step > mfma_per_step || (coeffs && (n_taps < 1 || n_taps > 1025))) {
    return fail(ACDSP_EINVAL, "diag_fir_envelope_copygeom: bad coefficient set or MFMA counts");
  }
  if (!diag_envelope_compiled(mfma_per_step, mfma_hi_per_step)) {
    return fail(ACDSP_EUNSUPPORTED, "diag_fir_envelope_copygeom: (%d, %d) MFMAs per step is not a compiled count", mfma_per_step, mfma_hi_per_step);
  }
  int rc = check_device(device);
  if (rc) { return rc; }
  hipStream_t s = (hipStream_t)stream;
  uint32_t *d_frag = nullptr;
  if (coeffs) {     // the real fragments, four elements per thread
    std::vector<uint32_t> six;
    if ((rc = envelope_fragments(coeffs, n_taps, mfma_per_step, &six))) { return rc; }
    HIP_TRY(hipMalloc((void **)&d_frag, six.size() * sizeof(uint32_t)));
    hipError_t ce = hipMemcpy(d_frag, six.data(), six.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (ce != hipSuccess) { (void)hipFree(d_frag); return fail(ACDSP_EHIP, "diag_fir_envelope_copygeom: upload failed: %s", hipGetErrorString(ce)); }
  }
  int out = time_launches([&] { return launch_diag_envelope_copygeom(d_frag, d_x, d_y, (int64_t)bytes, mfma_per_step, mfma_hi_per_step, s); }, warmup, reps, s, ms_avg);
  if (d_frag) { (void)hipFree(d_frag); }
  return out;
}

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
  // A blob of another history LENGTH is still this filter's state as long as it covers the filter memory (N R M' - 1 inputs for the
  // decimator): a handle that can take the two-stage kernel (cic2.hip) keeps a longer history than one that cannot (ACDSP_NO_CIC2, an
  // older build), and the samples beyond the memory only ever feed warm-up values the combs cancel.  The newest min(blob, mine) samples
  // are kept, older ones zero.
  StateHdr same_len = s;
  same_len.per_channel = mine.per_channel;
  const uint64_t mem = h->d.interp ? (uint64_t)h->d.N * h->me + 1 : (uint64_t)h->d.N * h->d.R * h->me - 1;
  const bool relen = s.per_channel != mine.per_channel && s.per_channel >= mem && s.per_channel <= (uint64_t(1) << 24) && state_compatible(same_len, mine) &&
                     bytes == sizeof s + state_payload(s);
  if ((!relen && (!state_compatible(s, mine) || bytes != sizeof s + state_payload(mine))) || s.t_total < 0) {
    return fail(ACDSP_EINVAL, "cic_state_set: blob does not belong to a CIC filter of these parameters / channel count");
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
  } else {
    HIP_TRY(hipMemcpy(h->d_hist[h->cur], (const char *)buf + sizeof s, state_payload(mine), hipMemcpyHostToDevice));
  }
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

