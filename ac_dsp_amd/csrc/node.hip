// node.hip -- channel-slice sharding over the GPUs of one node, as a PRODUCT entry point (acdsp_node_* in include/acdsp.h).
//
// Every channel is an independent filter object with private state -- no cross-channel term anywhere in the reference
// (include/ac_dsp/ac_fir_const_coeffs.h:124-127, ac_cic_full_core.h:71-74,219) -- so a bank of n_channels filters splits into
// contiguous channel slices, one per device, with replicated coefficients and NO collective ("one host thread + stream per
// device, per-device timing, aggregate = sum of samples / max(time)").  Round 3 had this only inside bench.py (one process per GPU);
// a caller with BASELINE config 4's 8192 channels wrote the thread-per-device loop themself.  Here:
//
//   node handle = n_shards x { device, channel slice [lo, hi), one ordinary engine handle (acdsp_fir_t / acdsp_cic_t / acdsp_ddc_t),
//                              one non-blocking HIP stream, one persistent host thread bound to that device }
//   run()       = every shard's thread launches its slice on its stream and waits for it; the call returns when all have
//                 (per-shard kernel time from the handle's own HIP events; acdsp_node_last_ms reports them and their maximum).
//
// The same device may appear more than once in the device list (shards are then concurrent streams of one GPU): that is how the
// one-GPU test box runs the N-shard path.  Nothing here touches sample data on the host except the *_run_host conveniences, which
// hand each thread its rows of the caller's dense [n_channels][n] host block.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fir_kernels.hpp"

using namespace acdsp;

namespace {

struct Worker {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // kinds whose engine handle keeps no kernel timer: the shard's call on its stream
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false, ready = false;
  int rc = 0, init_rc = 0;
  std::string err;

  void loop() {
    int rc0 = ACDSP_OK;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) { rc0 = ACDSP_EHIP; }
    {
      std::lock_guard<std::mutex> lk(m);
      init_rc = rc0; ready = true;
    }
    cv.notify_all();
    for (;;) {
      std::function<int()> j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return has_job || quit; });
        if (quit) { break; }
        j = job;
      }
      const int r = j();
      const std::string e = r != ACDSP_OK ? std::string(acdsp_last_error()) : std::string();   // the engine's message is thread-local
      {
        std::lock_guard<std::mutex> lk(m);
        rc = r; err = e; has_job = false; done = true;
      }
      cv.notify_all();
    }
    if (ev0) { (void)hipEventDestroy(ev0); }
    if (ev1) { (void)hipEventDestroy(ev1); }
    if (stream) { (void)hipStreamDestroy(stream); }
  }
  int start(int dev) {
    device = dev;
    th = std::thread([this] { loop(); });
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return ready; });
    return init_rc;
  }
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = std::move(j); has_job = true; done = false;
    }
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
    return rc;
  }
  void stop() {
    if (!th.joinable()) { return; }
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv.notify_all();
    th.join();
  }
};

enum { kNodeFir = 1, kNodeCic = 2, kNodeDdc = 3, kNodePolyDec = 4, kNodePolyIntr = 5, kNodeIntgDump = 6, kNodeMvAvg = 7 };

struct Shard {
  int device = 0;
  int64_t lo = 0, hi = 0;
  void *handle = nullptr;
  float last_ms = 0;
  std::unique_ptr<Worker> w;
};

}  // namespace

struct acdsp_node {
  int kind = 0;
  int64_t n_channels = 0;
  int32_t n_taps = 0, coeffs_per_channel = 0, in_eb = 0, out_eb = 0;
  std::vector<Shard> shards;
};

namespace {

void slice(int64_t n_total, int n_shards, int shard, int64_t *lo, int64_t *hi) {
  const int64_t base = n_total / n_shards, rem = n_total % n_shards;
  *lo = shard * base + (shard < rem ? shard : rem);
  *hi = *lo + base + (shard < rem ? 1 : 0);
}

// run job(shard index) on every shard's thread, wait for all; first failure wins (its message becomes the caller's last error)
int run_all(acdsp_node *h, const std::function<int(int)> &job) {
  for (size_t i = 0; i < h->shards.size(); i++) {
    const int idx = (int)i;
    h->shards[i].w->post([&job, idx] { return job(idx); });
  }
  int rc = ACDSP_OK;
  std::string err;
  for (size_t i = 0; i < h->shards.size(); i++) {
    const int r = h->shards[i].w->wait();
    if (r != ACDSP_OK && rc == ACDSP_OK) { rc = r; err = "shard " + std::to_string(i) + " (device " + std::to_string(h->shards[i].device) + "): " + h->shards[i].w->err; }
  }
  return rc == ACDSP_OK ? ACDSP_OK : set_error(rc, err.c_str());
}

int make_node(int kind, int64_t n_channels, int32_t n_devices, const int32_t *devices, acdsp_node **out) {
  if (!out) { return set_error(ACDSP_EINVAL, "node create: null output pointer"); }
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) { return set_error(ACDSP_EINVAL, "node create: n_devices must be 1..64"); }
  if (n_channels < n_devices) { return set_error(ACDSP_EINVAL, "node create: fewer channels than shards"); }
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { return set_error(ACDSP_ENODEVICE, "no HIP device visible"); }
  std::unique_ptr<acdsp_node> h(new acdsp_node);
  h->kind = kind; h->n_channels = n_channels;
  h->shards.resize((size_t)n_devices);
  for (int i = 0; i < n_devices; i++) {
    Shard &s = h->shards[(size_t)i];
    s.device = devices ? devices[i] : i;
    if (s.device < 0 || s.device >= visible) {
      for (int k = 0; k < i; k++) { h->shards[(size_t)k].w->stop(); }
      return set_error(ACDSP_EINVAL, "node create: device out of range");
    }
    slice(n_channels, n_devices, i, &s.lo, &s.hi);
    s.w.reset(new Worker);
    if (s.w->start(s.device) != ACDSP_OK) {
      for (int k = 0; k <= i; k++) { h->shards[(size_t)k].w->stop(); }
      return set_error(ACDSP_EHIP, "node create: could not bind a host thread and stream to the device");
    }
  }
  *out = h.release();
  return ACDSP_OK;
}

void destroy_handle(int kind, void *p) {
  if (!p) { return; }
  if (kind == kNodeFir) { (void)acdsp_fir_destroy((acdsp_fir_t)p); }
  else if (kind == kNodeCic) { (void)acdsp_cic_destroy((acdsp_cic_t)p); }
  else if (kind == kNodeDdc) { (void)acdsp_ddc_destroy((acdsp_ddc_t)p); }
  else if (kind == kNodePolyDec) { (void)acdsp_polydec_destroy((acdsp_polydec_t)p); }
  else if (kind == kNodePolyIntr) { (void)acdsp_polyintr_destroy((acdsp_polyintr_t)p); }
  else if (kind == kNodeIntgDump) { (void)acdsp_intgdump_destroy((acdsp_intgdump_t)p); }
  else { (void)acdsp_mvavg_destroy((acdsp_mvavg_t)p); }
}

int free_node(acdsp_node *h) {
  if (!h) { return ACDSP_OK; }
  const int kind = h->kind;
  // handles are destroyed on their own threads (device already current there), then the threads stop
  (void)run_all(h, [h, kind](int i) { destroy_handle(kind, h->shards[(size_t)i].handle); h->shards[(size_t)i].handle = nullptr; return (int)ACDSP_OK; });
  for (auto &s : h->shards) { s.w->stop(); }
  delete h;
  return ACDSP_OK;
}

int check_node(acdsp_node *h, int kind, const char *what) {
  if (!h) { return set_error(ACDSP_EINVAL, "null node handle"); }
  if (h->kind != kind) { return set_error(ACDSP_EINVAL, what); }
  return ACDSP_OK;
}

int finish_timed(Shard &s, int kind) {
  const hipError_t e = hipStreamSynchronize(s.w->stream);
  if (e != hipSuccess) { return set_error(ACDSP_EHIP, hipGetErrorString(e)); }
  float ms = 0, mn = 0;
  int rc = kind == kNodeFir ? acdsp_fir_last_kernel_ms((acdsp_fir_t)s.handle, &ms)
                            : (kind == kNodeCic ? acdsp_cic_last_kernel_ms((acdsp_cic_t)s.handle, &ms) : acdsp_ddc_kernel_stats((acdsp_ddc_t)s.handle, 1, &ms, &mn));
  s.last_ms = rc == ACDSP_OK ? ms : 0.f;   // calls too small to be timed leave 0
  return ACDSP_OK;
}

// the f-row classes (poly_dec, poly_intr, intg_dump, mv_avg): run `call` on the shard's stream between two events of that stream
template <typename F>
int run_timed(Shard &s, F call) {
  if (hipEventRecord(s.w->ev0, s.w->stream) != hipSuccess) { return set_error(ACDSP_EHIP, "node: event record failed"); }
  const int r = call();
  if (r) { return r; }
  hipError_t e = hipEventRecord(s.w->ev1, s.w->stream);
  if (e == hipSuccess) { e = hipStreamSynchronize(s.w->stream); }
  float ms = 0;
  if (e == hipSuccess) { e = hipEventElapsedTime(&ms, s.w->ev0, s.w->ev1); }
  if (e != hipSuccess) { return set_error(ACDSP_EHIP, hipGetErrorString(e)); }
  s.last_ms = ms;
  return ACDSP_OK;
}

// create one engine handle per shard with `make(desc with this slice's row count and device)`
template <typename D, typename H, typename MK>
int create_shards(acdsp_node *h, const D &d0, int32_t D::*rows, MK make) {
  const int rc = run_all(h, [h, d0, rows, make](int i) {
    Shard &s = h->shards[(size_t)i];
    D d = d0;
    d.*rows = (int32_t)(s.hi - s.lo); d.device = s.device;
    H e = nullptr;
    const int r = make(&d, &e);
    s.handle = e;
    return r;
  });
  if (rc) { const std::string keep = acdsp_last_error(); free_node(h); return set_error(rc, keep.c_str()); }
  return ACDSP_OK;
}

}  // namespace

extern "C" {

int32_t acdsp_node_shard(int64_t n_total, int32_t n_shards, int32_t shard, int64_t *lo, int64_t *hi) {
  if (n_shards < 1 || shard < 0 || shard >= n_shards || n_total < 0 || !lo || !hi) { return set_error(ACDSP_EINVAL, "node_shard: bad arguments"); }
  slice(n_total, n_shards, shard, lo, hi);
  return ACDSP_OK;
}

int32_t acdsp_node_n_shards(acdsp_node_t h) { return h ? (int32_t)h->shards.size() : -1; }

int32_t acdsp_node_shard_info(acdsp_node_t h, int32_t shard, int32_t *device, int64_t *ch_lo, int64_t *ch_hi, void **handle, void **stream) {
  if (!h || shard < 0 || shard >= (int32_t)h->shards.size()) { return set_error(ACDSP_EINVAL, "node_shard_info: bad handle or shard index"); }
  const Shard &s = h->shards[(size_t)shard];
  if (device) { *device = s.device; }
  if (ch_lo) { *ch_lo = s.lo; }
  if (ch_hi) { *ch_hi = s.hi; }
  if (handle) { *handle = s.handle; }
  if (stream) { *stream = (void *)s.w->stream; }
  return ACDSP_OK;
}

int32_t acdsp_node_last_ms(acdsp_node_t h, float *per_shard_ms, float *max_ms) {
  if (!h) { return set_error(ACDSP_EINVAL, "null node handle"); }
  float mx = 0;
  for (size_t i = 0; i < h->shards.size(); i++) {
    if (per_shard_ms) { per_shard_ms[i] = h->shards[i].last_ms; }
    mx = h->shards[i].last_ms > mx ? h->shards[i].last_ms : mx;
  }
  if (max_ms) { *max_ms = mx; }
  return ACDSP_OK;
}

int32_t acdsp_node_destroy(acdsp_node_t h) { return free_node(h); }

// ---- FIR ----
int32_t acdsp_node_fir_create(const acdsp_fir_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_fir_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodeFir, desc->n_channels, n_devices, devices, &h);
  if (rc) { return rc; }
  h->n_taps = desc->n_taps; h->coeffs_per_channel = desc->coeffs_per_channel;
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  const acdsp_fir_desc_t d0 = *desc;
  rc = run_all(h, [h, d0](int i) {
    Shard &s = h->shards[(size_t)i];
    acdsp_fir_desc_t d = d0;
    d.n_channels = (int32_t)(s.hi - s.lo); d.device = s.device;
    acdsp_fir_t f = nullptr;
    const int r = acdsp_fir_create(&d, &f);
    s.handle = f;
    return r;
  });
  if (rc) { const std::string keep = acdsp_last_error(); free_node(h); return set_error(rc, keep.c_str()); }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_node_fir_set_coeffs(acdsp_node_t h, const int64_t *coeffs) {
  int rc = check_node(h, kNodeFir, "not a FIR node handle");
  if (rc) { return rc; }
  if (!coeffs) { return set_error(ACDSP_EINVAL, "node_fir_set_coeffs: null coefficients"); }
  return run_all(h, [h, coeffs](int i) {
    Shard &s = h->shards[(size_t)i];
    return (int)acdsp_fir_set_coeffs((acdsp_fir_t)s.handle, coeffs + (h->coeffs_per_channel ? s.lo * h->n_taps : 0));   // replicated, or this slice's sets
  });
}

int32_t acdsp_node_fir_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n, void *const *d_out, int64_t out_stride) {
  int rc = check_node(h, kNodeFir, "not a FIR node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_fir_run: null pointer arrays"); }
  return run_all(h, [=](int i) {
    Shard &s = h->shards[(size_t)i];
    const int r = acdsp_fir_run((acdsp_fir_t)s.handle, d_in[i], in_stride, n, d_out[i], out_stride, (void *)s.w->stream);
    return r ? r : finish_timed(s, kNodeFir);
  });
}

int32_t acdsp_node_fir_run_host(acdsp_node_t h, const void *h_in, int64_t n, void *h_out) {
  int rc = check_node(h, kNodeFir, "not a FIR node handle");
  if (rc) { return rc; }
  if (n > 0 && (!h_in || !h_out)) { return set_error(ACDSP_EINVAL, "node_fir_run_host: null buffers"); }
  return run_all(h, [=](int i) {
    Shard &s = h->shards[(size_t)i];
    return (int)acdsp_fir_run_host((acdsp_fir_t)s.handle, (const char *)h_in + s.lo * n * h->in_eb, n, (char *)h_out + s.lo * n * h->out_eb);
  });
}

// ---- CIC ----
int32_t acdsp_node_cic_create(const acdsp_cic_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_cic_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodeCic, desc->n_channels, n_devices, devices, &h);
  if (rc) { return rc; }
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  const acdsp_cic_desc_t d0 = *desc;
  rc = run_all(h, [h, d0](int i) {
    Shard &s = h->shards[(size_t)i];
    acdsp_cic_desc_t d = d0;
    d.n_channels = (int32_t)(s.hi - s.lo); d.device = s.device;
    acdsp_cic_t c = nullptr;
    const int r = acdsp_cic_create(&d, &c);
    s.handle = c;
    return r;
  });
  if (rc) { const std::string keep = acdsp_last_error(); free_node(h); return set_error(rc, keep.c_str()); }
  *out = h;
  return ACDSP_OK;
}

int64_t acdsp_node_cic_out_count(acdsp_node_t h, int64_t n_in) {
  if (!h || h->kind != kNodeCic || h->shards.empty()) { return -1; }
  return acdsp_cic_out_count((acdsp_cic_t)h->shards[0].handle, n_in);   // every shard is in the same phase
}

int32_t acdsp_node_cic_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride,
                           int64_t *n_out) {
  int rc = check_node(h, kNodeCic, "not a CIC node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_cic_run: null pointer arrays"); }
  std::vector<int64_t> no(h->shards.size(), 0);
  rc = run_all(h, [&, h](int i) {
    Shard &s = h->shards[(size_t)i];
    const int r = acdsp_cic_run((acdsp_cic_t)s.handle, d_in[i], in_stride, n_in, d_out[i], out_stride, &no[(size_t)i], (void *)s.w->stream);
    return r ? r : finish_timed(s, kNodeCic);
  });
  if (rc == ACDSP_OK && n_out) { *n_out = no[0]; }
  return rc;
}

// ---- fused DDC ----
int32_t acdsp_node_ddc_create(const acdsp_cic_desc_t *cic, const acdsp_fir_desc_t *fir, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!cic || !fir) { return set_error(ACDSP_EINVAL, "node_ddc_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodeDdc, cic->n_channels, n_devices, devices, &h);
  if (rc) { return rc; }
  h->n_taps = fir->n_taps;
  h->in_eb = acdsp_elem_bytes(cic->in.W); h->out_eb = acdsp_elem_bytes(fir->out.W);
  const acdsp_cic_desc_t c0 = *cic;
  const acdsp_fir_desc_t f0 = *fir;
  rc = run_all(h, [h, c0, f0](int i) {
    Shard &s = h->shards[(size_t)i];
    acdsp_cic_desc_t c = c0;
    acdsp_fir_desc_t f = f0;
    c.n_channels = f.n_channels = (int32_t)(s.hi - s.lo); c.device = f.device = s.device;
    acdsp_ddc_t d = nullptr;
    const int r = acdsp_ddc_create(&c, &f, &d);
    s.handle = d;
    return r;
  });
  if (rc) { const std::string keep = acdsp_last_error(); free_node(h); return set_error(rc, keep.c_str()); }
  *out = h;
  return ACDSP_OK;
}

int32_t acdsp_node_ddc_set_coeffs(acdsp_node_t h, const int64_t *coeffs) {
  int rc = check_node(h, kNodeDdc, "not a DDC node handle");
  if (rc) { return rc; }
  if (!coeffs) { return set_error(ACDSP_EINVAL, "node_ddc_set_coeffs: null coefficients"); }
  return run_all(h, [h, coeffs](int i) { return (int)acdsp_ddc_set_coeffs((acdsp_ddc_t)h->shards[(size_t)i].handle, coeffs); });
}

int64_t acdsp_node_ddc_out_count(acdsp_node_t h, int64_t n_in) {
  if (!h || h->kind != kNodeDdc || h->shards.empty()) { return -1; }
  return acdsp_ddc_out_count((acdsp_ddc_t)h->shards[0].handle, n_in);
}

int32_t acdsp_node_ddc_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride,
                           int64_t *n_out) {
  int rc = check_node(h, kNodeDdc, "not a DDC node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_ddc_run: null pointer arrays"); }
  std::vector<int64_t> no(h->shards.size(), 0);
  rc = run_all(h, [&, h](int i) {
    Shard &s = h->shards[(size_t)i];
    const int r = acdsp_ddc_run((acdsp_ddc_t)s.handle, d_in[i], in_stride, n_in, d_out[i], out_stride, &no[(size_t)i], (void *)s.w->stream);
    return r ? r : finish_timed(s, kNodeDdc);
  });
  if (rc == ACDSP_OK && n_out) { *n_out = no[0]; }
  return rc;
}

// ---- the f-row classes (round 5): same slicing, one engine handle of the class per shard ----
int32_t acdsp_node_polydec_create(const acdsp_polydec_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_polydec_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodePolyDec, desc->n_channels, n_devices, devices, &h);
  if (rc) { return rc; }
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  if ((rc = create_shards<acdsp_polydec_desc_t, acdsp_polydec_t>(h, *desc, &acdsp_polydec_desc_t::n_channels, acdsp_polydec_create))) { return rc; }
  *out = h;
  return ACDSP_OK;
}
int32_t acdsp_node_polydec_set_coeffs(acdsp_node_t h, const int64_t *coeffs) {
  int rc = check_node(h, kNodePolyDec, "not a poly_dec node handle");
  if (rc) { return rc; }
  if (!coeffs) { return set_error(ACDSP_EINVAL, "node_polydec_set_coeffs: null coefficients"); }
  return run_all(h, [h, coeffs](int i) { return (int)acdsp_polydec_set_coeffs((acdsp_polydec_t)h->shards[(size_t)i].handle, coeffs); });
}
int32_t acdsp_node_polydec_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride) {
  int rc = check_node(h, kNodePolyDec, "not a poly_dec node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_polydec_run: null pointer arrays"); }
  return run_all(h, [=](int i) {
    Shard &s = h->shards[(size_t)i];
    return run_timed(s, [&] { return (int)acdsp_polydec_run((acdsp_polydec_t)s.handle, d_in[i], in_stride, n_in, d_out[i], out_stride, (void *)s.w->stream); });
  });
}

int32_t acdsp_node_polyintr_create(const acdsp_polyintr_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_polyintr_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodePolyIntr, desc->n_channels, n_devices, devices, &h);
  if (rc) { return rc; }
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  if ((rc = create_shards<acdsp_polyintr_desc_t, acdsp_polyintr_t>(h, *desc, &acdsp_polyintr_desc_t::n_channels, acdsp_polyintr_create))) { return rc; }
  *out = h;
  return ACDSP_OK;
}
int32_t acdsp_node_polyintr_set_ctrl(acdsp_node_t h, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr) {
  int rc = check_node(h, kNodePolyIntr, "not a poly_intr node handle");
  if (rc) { return rc; }
  return run_all(h, [=](int i) { return (int)acdsp_polyintr_set_ctrl((acdsp_polyintr_t)h->shards[(size_t)i].handle, coeffs, sign, corr); });
}
int64_t acdsp_node_polyintr_out_count(acdsp_node_t h, int64_t n_in) {
  if (!h || h->kind != kNodePolyIntr || h->shards.empty()) { return -1; }
  return acdsp_polyintr_out_count((acdsp_polyintr_t)h->shards[0].handle, n_in);
}
int32_t acdsp_node_polyintr_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride, int64_t *n_out) {
  int rc = check_node(h, kNodePolyIntr, "not a poly_intr node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_polyintr_run: null pointer arrays"); }
  std::vector<int64_t> no(h->shards.size(), 0);
  rc = run_all(h, [&, h](int i) {
    Shard &s = h->shards[(size_t)i];
    return run_timed(s, [&] { return (int)acdsp_polyintr_run((acdsp_polyintr_t)s.handle, d_in[i], in_stride, n_in, d_out[i], out_stride, &no[(size_t)i], (void *)s.w->stream); });
  });
  if (rc == ACDSP_OK && n_out) { *n_out = no[0]; }
  return rc;
}

int32_t acdsp_node_intgdump_create(const acdsp_intgdump_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_intgdump_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodeIntgDump, desc->n_objects, n_devices, devices, &h);
  if (rc) { return rc; }
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  if ((rc = create_shards<acdsp_intgdump_desc_t, acdsp_intgdump_t>(h, *desc, &acdsp_intgdump_desc_t::n_objects, acdsp_intgdump_create))) { return rc; }
  *out = h;
  return ACDSP_OK;
}
int32_t acdsp_node_intgdump_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, const int64_t *n_sample, int64_t n_blocks, void *const *d_out,
                                int64_t out_stride, int64_t *n_out) {
  int rc = check_node(h, kNodeIntgDump, "not an intg_dump node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_intgdump_run: null pointer arrays"); }
  std::vector<int64_t> no(h->shards.size(), 0);
  rc = run_all(h, [&, h](int i) {
    Shard &s = h->shards[(size_t)i];
    return run_timed(s, [&] { return (int)acdsp_intgdump_run((acdsp_intgdump_t)s.handle, d_in[i], in_stride, n_sample, n_blocks, d_out[i], out_stride, &no[(size_t)i], (void *)s.w->stream); });
  });
  if (rc == ACDSP_OK && n_out) { *n_out = no[0]; }
  return rc;
}

int32_t acdsp_node_mvavg_create(const acdsp_mvavg_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out) {
  if (!desc) { return set_error(ACDSP_EINVAL, "node_mvavg_create: null descriptor"); }
  acdsp_node *h = nullptr;
  int rc = make_node(kNodeMvAvg, desc->n_objects, n_devices, devices, &h);
  if (rc) { return rc; }
  h->in_eb = acdsp_elem_bytes(desc->in.W); h->out_eb = acdsp_elem_bytes(desc->out.W);
  if ((rc = create_shards<acdsp_mvavg_desc_t, acdsp_mvavg_t>(h, *desc, &acdsp_mvavg_desc_t::n_objects, acdsp_mvavg_create))) { return rc; }
  *out = h;
  return ACDSP_OK;
}
int32_t acdsp_node_mvavg_set_coeffs(acdsp_node_t h, const int64_t *coeffs) {
  int rc = check_node(h, kNodeMvAvg, "not a mv_avg node handle");
  if (rc) { return rc; }
  if (!coeffs) { return set_error(ACDSP_EINVAL, "node_mvavg_set_coeffs: null coefficients"); }
  return run_all(h, [h, coeffs](int i) { return (int)acdsp_mvavg_set_coeffs((acdsp_mvavg_t)h->shards[(size_t)i].handle, coeffs); });
}
int32_t acdsp_node_mvavg_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_sample, int64_t n_frames, void *const *d_out,
                             int64_t out_stride, int64_t *n_out) {
  int rc = check_node(h, kNodeMvAvg, "not a mv_avg node handle");
  if (rc) { return rc; }
  if (!d_in || !d_out) { return set_error(ACDSP_EINVAL, "node_mvavg_run: null pointer arrays"); }
  std::vector<int64_t> no(h->shards.size(), 0);
  rc = run_all(h, [&, h](int i) {
    Shard &s = h->shards[(size_t)i];
    return run_timed(s, [&] { return (int)acdsp_mvavg_run((acdsp_mvavg_t)s.handle, d_in[i], in_stride, n_sample, n_frames, d_out[i], out_stride, &no[(size_t)i], (void *)s.w->stream); });
  });
  if (rc == ACDSP_OK && n_out) { *n_out = no[0]; }
  return rc;
}

}  // extern "C"
