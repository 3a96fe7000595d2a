// acdsp_engine.h -- C++ (header-only) front end of the MI355X filter engine.
//
// Maps ac_fixed template parameters onto the runtime descriptors of the C ABI
// (include/acdsp.h) and owns the engine handles.  Two layers use it:
//   * the drop-in class templates ac_fir_{const,load,prog}_coeffs and
//     ac_cic_{dec,intr}_full (one channel, ac_channel I/O, same signatures as
//     the reference's include/ac_dsp/*.h), and
//   * the batched many-channel API acdsp::fir_engine / acdsp::cic_engine, which
//     works on [channel][time] arrays of raw words (host or device memory).
//
// Only public AC Datatypes API is used on the ac_fixed side (width, i_width,
// sign, q_mode, o_mode, slc, set_slc), so the headers work both with the
// subset in include/ac_types/ and with a genuine hlslibs/ac_types install.
//
// Errors: the reference signals none (void run()).  Here every non-zero engine
// status aborts with the engine's message -- never a silent wrong answer and
// never a CPU fallback.
#ifndef AC_DSP_AMD_ACDSP_ENGINE_H
#define AC_DSP_AMD_ACDSP_ENGINE_H

#include <ac_channel.h>
#include <ac_fixed.h>
#include <ac_int.h>
#include <acdsp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace acdsp {

inline void check(int32_t rc, const char *what) {
  if (rc != ACDSP_OK) {
    fprintf(stderr, "ac_dsp_amd: %s failed (%d): %s\n", what, (int)rc, acdsp_last_error());
    abort();
  }
}

// Device used by objects that are constructed without an explicit one.
inline int default_device() {
  const char *e = getenv("ACDSP_DEVICE");
  return e ? atoi(e) : 0;
}

inline int32_t q_code(ac_q_mode q) {
  switch (q) {
    case AC_TRN: return ACDSP_TRN;
    case AC_RND: return ACDSP_RND;
    case AC_TRN_ZERO: return ACDSP_TRN_ZERO;
    case AC_RND_ZERO: return ACDSP_RND_ZERO;
    case AC_RND_INF: return ACDSP_RND_INF;
    case AC_RND_MIN_INF: return ACDSP_RND_MIN_INF;
    case AC_RND_CONV: return ACDSP_RND_CONV;
    case AC_RND_CONV_ODD: return ACDSP_RND_CONV_ODD;
  }
  return -1;
}
inline int32_t o_code(ac_o_mode o) {
  switch (o) {
    case AC_WRAP: return ACDSP_WRAP;
    case AC_SAT: return ACDSP_SAT;
    case AC_SAT_ZERO: return ACDSP_SAT_ZERO;
    case AC_SAT_SYM: return ACDSP_SAT_SYM;
  }
  return -1;
}

// ac_fixed<W,I,S,Q,O>  ->  acdsp_fmt_t   (cf. reference ac_cic_dec_full.h:119-121 reading T::width etc.)
template <class T> inline acdsp_fmt_t fmt_of() {
  static_assert(T::width >= 1 && T::width <= 128, "ac_dsp_amd engine: ac_fixed widths up to 128 bits (IN / COEFF: 64; checked by the engine)");
  acdsp_fmt_t f;
  f.W = T::width; f.I = T::i_width; f.S = T::sign ? 1 : 0;
  f.Q = q_code(T::q_mode); f.O = o_code(T::o_mode);
  return f;
}

template <class T> inline int64_t raw_of(const T &v) {
  return (int64_t)v.template slc<T::width>(0).to_int64();  // ac_int<W,S>: already sign/zero extended
}
template <class T> inline T from_raw(int64_t r) {
  T v;
  v.set_slc(0, ac_int<T::width, T::sign>((long long)r));
  return v;
}

// One ac_fixed value <-> its engine container (int16 / int32 / int64 by width; 16 bytes, low quadword first, above 64 bits:
// ACC_TYPE / OUT_TYPE of the FIR classes and the CIC INT_TYPE may be that wide, reference ac_cic_dec_full.h:116-137).
template <class T, bool WIDE = (T::width > 64)> struct container_io {
  static void store(const T &v, unsigned char *p, int eb) {
    const int64_t r = raw_of(v);
    if (eb == 2) { int16_t t = (int16_t)r; memcpy(p, &t, 2); }
    else if (eb == 4) { int32_t t = (int32_t)r; memcpy(p, &t, 4); }
    else { memcpy(p, &r, 8); }
  }
  static T load(const unsigned char *p, int eb) {
    int64_t r;
    if (eb == 2) { int16_t t; memcpy(&t, p, 2); r = T::sign ? (int64_t)t : (int64_t)(uint16_t)t; }
    else if (eb == 4) { int32_t t; memcpy(&t, p, 4); r = T::sign ? (int64_t)t : (int64_t)(uint32_t)t; }
    else { memcpy(&r, p, 8); }
    return from_raw<T>(r);
  }
};
template <class T> struct container_io<T, true> {
  enum { HW = T::width - 64 };
  static void store(const T &v, unsigned char *p, int) {
    const uint64_t lo = (uint64_t)v.template slc<64>(0).to_int64();
    const int64_t hi = (int64_t)v.template slc<HW>(64).to_int64();   // ac_int<HW, S>: sign / zero extended
    memcpy(p, &lo, 8); memcpy(p + 8, &hi, 8);
  }
  static T load(const unsigned char *p, int) {
    uint64_t lo; int64_t hi;
    memcpy(&lo, p, 8); memcpy(&hi, p + 8, 8);
    T v;
    v.set_slc(0, ac_int<64, false>((unsigned long long)lo));
    v.set_slc(64, ac_int<HW, T::sign>((long long)hi));
    return v;
  }
};

// pack / unpack raw words into the engine's containers (int16 / int32 / int64 by width)
inline void pack(const std::vector<int64_t> &src, int eb, std::vector<unsigned char> &dst) {
  dst.resize(src.size() * (size_t)eb);
  for (size_t i = 0; i < src.size(); i++) {
    if (eb == 2) { int16_t t = (int16_t)src[i]; memcpy(&dst[i * 2], &t, 2); }
    else if (eb == 4) { int32_t t = (int32_t)src[i]; memcpy(&dst[i * 4], &t, 4); }
    else { memcpy(&dst[i * 8], &src[i], 8); }
  }
}
inline int64_t unpack_one(const unsigned char *p, int eb, bool is_signed) {
  if (eb == 2) { int16_t t; memcpy(&t, p, 2); return is_signed ? (int64_t)t : (int64_t)(uint16_t)t; }
  if (eb == 4) { int32_t t; memcpy(&t, p, 4); return is_signed ? (int64_t)t : (int64_t)(uint32_t)t; }
  int64_t t; memcpy(&t, p, 8); return t;
}

// ---------------------------------------------------------------------------------------------
// Batched FIR: n_channels independent filters behind one engine handle.
// ---------------------------------------------------------------------------------------------
template <class IN_TYPE, class OUT_TYPE, class COEFF_TYPE, class ACC_TYPE>
class fir_engine {
public:
  fir_engine(int kind, int ftype, int n_taps, int n_channels = 1, bool coeffs_per_channel = false, int device = -1)
      : h_(0), kind_(kind), ftype_(ftype), n_taps_(n_taps), n_ch_(n_channels), per_ch_(coeffs_per_channel),
        device_(device < 0 ? default_device() : device), head_(0), host_valid_(true), host_ahead_(false) {}
  ~fir_engine() { if (h_) { acdsp_fir_destroy(h_); } }
  // The reference objects are plain aggregates: copying one copies its state.
  fir_engine(const fir_engine &o)
      : h_(0), kind_(o.kind_), ftype_(o.ftype_), n_taps_(o.n_taps_), n_ch_(o.n_ch_), per_ch_(o.per_ch_), device_(o.device_),
        coeffs_(o.coeffs_), ring_(o.ring_), rt_(o.rt_), head_(o.head_), host_valid_(o.host_valid_), host_ahead_(o.host_ahead_) {
    if (o.h_) { check(acdsp_fir_clone(o.h_, &h_), "acdsp_fir_clone"); }
  }
  fir_engine &operator=(const fir_engine &o) {
    if (this != &o) {
      if (h_) { acdsp_fir_destroy(h_); h_ = 0; }
      kind_ = o.kind_; ftype_ = o.ftype_; n_taps_ = o.n_taps_; n_ch_ = o.n_ch_; per_ch_ = o.per_ch_; device_ = o.device_;
      coeffs_ = o.coeffs_;
      ring_ = o.ring_; rt_ = o.rt_; head_ = o.head_; host_valid_ = o.host_valid_; host_ahead_ = o.host_ahead_;
      if (o.h_) { check(acdsp_fir_clone(o.h_, &h_), "acdsp_fir_clone"); }
    }
    return *this;
  }

  int n_channels() const { return n_ch_; }
  int in_bytes() const { return acdsp_elem_bytes(IN_TYPE::width); }
  int out_bytes() const { return acdsp_elem_bytes(OUT_TYPE::width); }
  // the raw engine handle (state blobs, kernel statistics ...): the device becomes the authority on the filter state
  acdsp_fir_t handle() { ensure(); to_device(); host_valid_ = false; return h_; }

  // raw coefficient words, [n_taps] or [n_channels][n_taps]; uploaded only when they changed
  void set_coeffs_raw(const std::vector<int64_t> &c) {
    ensure();
    if (c == coeffs_) { return; }
    check(acdsp_fir_set_coeffs(h_, c.data()), "acdsp_fir_set_coeffs");
    coeffs_ = c;
  }
  void set_coeffs(const COEFF_TYPE *c) {
    std::vector<int64_t> r((size_t)n_taps_ * (per_ch_ ? n_ch_ : 1));
    for (size_t i = 0; i < r.size(); i++) { r[i] = raw_of(c[i]); }
    set_coeffs_raw(r);
  }
  // device-resident streams (the hot path): containers of in_bytes()/out_bytes(), strides in elements
  void run_device(const void *d_in, int64_t in_stride, int64_t n, void *d_out, int64_t out_stride, void *stream = 0) {
    ensure(); to_device(); host_valid_ = false;
    check(acdsp_fir_run(h_, d_in, in_stride, n, d_out, out_stride, stream), "acdsp_fir_run");
  }
  // host-resident dense [n_channels][n] containers
  void run_host(const void *h_in, int64_t n, void *h_out) {
    ensure(); to_device(); host_valid_ = false;
    check(acdsp_fir_run_host(h_, h_in, n, h_out), "acdsp_fir_run_host");
  }
  // one channel of ac_fixed values in, ac_fixed values out (used by the drop-in run() bodies)
  void run_values(const std::vector<IN_TYPE> &x, std::vector<OUT_TYPE> &y) {
    std::vector<unsigned char> bi(x.size() * (size_t)in_bytes()), bo(x.size() * (size_t)out_bytes());
    for (size_t i = 0; i < x.size(); i++) { container_io<IN_TYPE>::store(x[i], &bi[i * in_bytes()], in_bytes()); }
    run_host(bi.data(), (int64_t)x.size(), bo.data());
    y.resize(x.size());
    for (size_t i = 0; i < x.size(); i++) { y[i] = container_io<OUT_TYPE>::load(&bo[i * out_bytes()], out_bytes()); }
  }
  void reset() {
    if (h_) { check(acdsp_fir_reset(h_), "acdsp_fir_reset"); }
    ring_.clear(); rt_.clear(); head_ = 0; host_valid_ = true; host_ahead_ = false;
  }

  // ---- the drop-in classes' entry point: one channel, the coefficient array of THIS call --------------------------------------------
  // A burst below the break-even of a kernel launch (~25 us: ac_fir_prog_coeffs::run is ONE sample per call, reference
  // ac_fir_prog_coeffs.h:281) runs HERE, on the caller's own ac_fixed arithmetic -- `acc += x * c` in the tap order of the
  // architecture, over a ring of the last N_TAPS samples (or the partial sums of TRANSPOSED) -- and the filter state moves between the
  // two sides as an engine state blob whenever the other side runs next.  Larger bursts (and every batched entry point above) go to
  // the GPU.  ACDSP_HOST_SMALL_MACS = samples x taps below which a call stays on the host (default 8192; 0: never).
  void run_values_c(const std::vector<IN_TYPE> &x, std::vector<OUT_TYPE> &y, const COEFF_TYPE *c) {
    if (host_path_ok() && (int64_t)x.size() * n_taps_ <= small_macs()) {
      to_host();
      y.resize(x.size());
      for (size_t i = 0; i < x.size(); i++) { y[i] = host_step(x[i], c); }
      host_ahead_ = true;
      return;
    }
    set_coeffs(c);
    run_values(x, y);
  }
  static int64_t small_macs() {
    static const int64_t v = getenv("ACDSP_HOST_SMALL_MACS") ? atoll(getenv("ACDSP_HOST_SMALL_MACS")) : 8192;
    return v;
  }

private:
  bool host_path_ok() const {
    return n_ch_ == 1 && !per_ch_ && (kind_ == ACDSP_FIR_CONST || kind_ == ACDSP_FIR_LOAD || kind_ == ACDSP_FIR_PROG) &&
           ftype_ >= ACDSP_SHIFT_REG && ftype_ <= ACDSP_TRANSPOSED && n_taps_ >= 1;
  }
  // reg_trans partial sums are the state of TRANSPOSED with coefficients that can change (the const class keeps an input history: DESIGN 3)
  bool chain_state() const { return ftype_ == ACDSP_TRANSPOSED && kind_ != ACDSP_FIR_CONST; }
  // sample x[n-k] of the ring, k = 0 .. N-1
  const IN_TYPE &w(int k) const { return ring_[(size_t)((head_ + k) % n_taps_)]; }
  OUT_TYPE host_step(const IN_TYPE &xin, const COEFF_TYPE *c) {
    const int N = n_taps_;
    OUT_TYPE out;
    if (chain_state()) {   // every partial sum takes this sample's product at its own coefficient (ac_fir_load_coeffs.h:265-278)
      for (int i = N - 1; i >= 0; i--) {
        ACC_TYPE below = 0;
        if (i > 0) { below = rt_[(size_t)(i - 1)]; }
        rt_[(size_t)i] = xin * c[N - 1 - i] + below;
      }
      out = rt_[(size_t)(N - 1)];
      return out;
    }
    head_ = (head_ + N - 1) % N;
    ring_[(size_t)head_] = xin;
    ACC_TYPE acc = 0;
    switch (ftype_) {
      case ACDSP_C_BUFF:                                   // newest sample first
        for (int k = 0; k < N; k++) { acc += w(k) * c[k]; }
        break;
      case ACDSP_FOLD_EVEN:                                // mirrored samples share a coefficient: exact pre-add
        for (int k = N / 2 - 1; k >= 0; k--) { acc += c[k] * (w(k) + w(N - 1 - k)); }
        break;
      case ACDSP_FOLD_ODD: {                               // ... with the pre-add held in an ACC_TYPE variable, the centre tap alone
        const int mid = (N - 1) / 2;
        for (int k = 0; k <= mid; k++) {
          ACC_TYPE folded;
          if (k == mid) { folded = w(k); } else { folded = w(k) + w(N - 1 - k); }
          acc += c[k] * folded;
        }
        break;
      }
      default:                                             // SHIFT_REG, ROTATE_SHIFT, TRANSPOSED with bound coefficients: oldest sample first
        for (int k = N - 1; k >= 0; k--) { acc += w(k) * c[k]; }
        break;
    }
    out = acc;
    return out;
  }
  // host mirror <- engine state blob (64-byte header, then the words of the one channel: include/acdsp.h)
  void to_host() {
    const size_t N = (size_t)n_taps_;
    if (host_valid_ && (ring_.size() == N || rt_.size() == N)) { return; }
    ring_.assign(chain_state() ? 0 : N, IN_TYPE(0));
    rt_.assign(chain_state() ? N : 0, ACC_TYPE(0));
    head_ = 0;
    if (h_ && !host_valid_) {
      const int64_t sz = acdsp_fir_state_size(h_);
      std::vector<unsigned char> blob((size_t)sz);
      check(acdsp_fir_state_get(h_, blob.data(), (uint64_t)sz), "acdsp_fir_state_get");
      const unsigned char *pay = blob.data() + 64;
      if (chain_state()) {
        const size_t eb = ((size_t)sz - 64) / N;           // 8-byte ACC raw words, 16 above 64 bits
        for (size_t i = 0; i < N; i++) { rt_[i] = container_io<ACC_TYPE>::load(pay + i * eb, (int)eb); }
      } else {
        const size_t eb = (size_t)in_bytes(), hl = ((size_t)sz - 64) / eb;   // oldest first: entry hl-1-k is x[n-k]
        for (size_t k = 0; k < N && k < hl; k++) { ring_[k] = container_io<IN_TYPE>::load(pay + (hl - 1 - k) * eb, (int)eb); }
      }
    }
    host_valid_ = true;
  }
  // engine state <- host mirror, when the host ran last
  void to_device() {
    if (!host_ahead_) { return; }
    ensure_raw();
    const size_t N = (size_t)n_taps_;
    const int64_t sz = acdsp_fir_state_size(h_);
    std::vector<unsigned char> blob((size_t)sz);
    check(acdsp_fir_state_get(h_, blob.data(), (uint64_t)sz), "acdsp_fir_state_get");   // the header of this handle's blobs
    unsigned char *pay = blob.data() + 64;
    memset(pay, 0, (size_t)sz - 64);
    if (chain_state()) {
      const size_t eb = ((size_t)sz - 64) / N;
      for (size_t i = 0; i < N; i++) { container_io<ACC_TYPE>::store(rt_[i], pay + i * eb, (int)eb); }
    } else {
      const size_t eb = (size_t)in_bytes(), hl = ((size_t)sz - 64) / eb;
      for (size_t k = 0; k < N && k < hl; k++) { container_io<IN_TYPE>::store(w((int)k), pay + (hl - 1 - k) * eb, (int)eb); }
    }
    check(acdsp_fir_state_set(h_, blob.data(), (uint64_t)sz), "acdsp_fir_state_set");
    host_ahead_ = false;
  }
  void ensure() { ensure_raw(); }
  void ensure_raw() {
    if (h_) { return; }
    acdsp_fir_desc_t d;
    d.kind = kind_; d.ftype = ftype_; d.n_taps = n_taps_; d.n_channels = n_ch_; d.coeffs_per_channel = per_ch_ ? 1 : 0;
    d.in = fmt_of<IN_TYPE>(); d.coeff = fmt_of<COEFF_TYPE>(); d.acc = fmt_of<ACC_TYPE>(); d.out = fmt_of<OUT_TYPE>();
    d.device = device_; d.flags = 0;
    check(acdsp_fir_create(&d, &h_), "acdsp_fir_create");
  }
  acdsp_fir_t h_;
  int kind_, ftype_, n_taps_, n_ch_;
  bool per_ch_;
  int device_;
  std::vector<int64_t> coeffs_;
  // host-side mirror of the one-channel state (tiny calls): ring of the last N_TAPS samples, or the TRANSPOSED partial sums
  std::vector<IN_TYPE> ring_;
  std::vector<ACC_TYPE> rt_;
  int head_;
  bool host_valid_;   // the mirror holds the current state
  bool host_ahead_;   // ... and the device does not (the host ran last)
};

// ---------------------------------------------------------------------------------------------
// Batched CIC
// ---------------------------------------------------------------------------------------------
template <class IN_TYPE, class OUT_TYPE>
class cic_engine {
public:
  cic_engine(bool interp, unsigned R, unsigned M, unsigned N, int n_channels = 1, int device = -1)
      : h_(0), interp_(interp), R_(R), M_(M), N_(N), n_ch_(n_channels), device_(device < 0 ? default_device() : device) {}
  ~cic_engine() { if (h_) { acdsp_cic_destroy(h_); } }
  cic_engine(const cic_engine &o) : h_(0), interp_(o.interp_), R_(o.R_), M_(o.M_), N_(o.N_), n_ch_(o.n_ch_), device_(o.device_) {
    if (o.h_) { check(acdsp_cic_clone(o.h_, &h_), "acdsp_cic_clone"); }
  }
  cic_engine &operator=(const cic_engine &o) {
    if (this != &o) {
      if (h_) { acdsp_cic_destroy(h_); h_ = 0; }
      interp_ = o.interp_; R_ = o.R_; M_ = o.M_; N_ = o.N_; n_ch_ = o.n_ch_; device_ = o.device_;
      if (o.h_) { check(acdsp_cic_clone(o.h_, &h_), "acdsp_cic_clone"); }
    }
    return *this;
  }

  int in_bytes() const { return acdsp_elem_bytes(IN_TYPE::width); }
  int out_bytes() const { return acdsp_elem_bytes(OUT_TYPE::width); }
  acdsp_cic_t handle() { ensure(); return h_; }
  int64_t out_count(int64_t n_in) { ensure(); return acdsp_cic_out_count(h_, n_in); }

  void run_device(const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride, int64_t *n_out, void *stream = 0) {
    ensure();
    check(acdsp_cic_run(h_, d_in, in_stride, n_in, d_out, out_stride, n_out, stream), "acdsp_cic_run");
  }
  void run_values(const std::vector<IN_TYPE> &x, std::vector<OUT_TYPE> &y) {
    ensure();
    y.clear();
    if (x.empty()) { return; }
    std::vector<int64_t> r(x.size());
    for (size_t i = 0; i < x.size(); i++) { r[i] = raw_of(x[i]); }
    std::vector<unsigned char> bi;
    pack(r, in_bytes(), bi);
    const int64_t cap = out_count((int64_t)x.size());
    std::vector<unsigned char> bo((size_t)(cap > 0 ? cap : 1) * (size_t)out_bytes());
    int64_t n_out = 0;
    check(acdsp_cic_run_host(h_, bi.data(), (int64_t)x.size(), bo.data(), cap, &n_out), "acdsp_cic_run_host");
    y.resize((size_t)n_out);
    for (int64_t i = 0; i < n_out; i++) { y[(size_t)i] = container_io<OUT_TYPE>::load(&bo[(size_t)i * out_bytes()], out_bytes()); }
  }

private:
  void ensure() {
    if (h_) { return; }
    acdsp_cic_desc_t d;
    d.interp = interp_ ? 1 : 0; d.R = (int32_t)R_; d.M = (int32_t)M_; d.N = (int32_t)N_; d.n_channels = n_ch_;
    d.in = fmt_of<IN_TYPE>(); d.out = fmt_of<OUT_TYPE>(); d.device = device_; d.flags = 0;
    check(acdsp_cic_create(&d, &h_), "acdsp_cic_create");
  }
  acdsp_cic_t h_;
  bool interp_;
  unsigned R_, M_, N_;
  int n_ch_, device_;
};

// DDC cascade on device-resident streams: n_channels x { ac_cic_dec_full<IN, INT, R, M, N> -> FIR<INT, OUT, COEFF, ACC> }
// with INT = the decimator's lossless INT_TYPE (INT_TYPE::width etc. are checked by the engine).  The BASELINE
// config-5 shape class runs as ONE kernel (the INT_TYPE stream stays on chip); others as the two stage kernels.
template <class IN_TYPE, class INT_TYPE, class OUT_TYPE, class COEFF_TYPE, class ACC_TYPE>
class ddc_engine {
public:
  ddc_engine(unsigned R, unsigned M, unsigned N, int fir_kind, int ftype, int n_taps, int n_channels = 1, int device = -1) : h_(0) {
    acdsp_cic_desc_t c;
    c.interp = 0; c.R = (int32_t)R; c.M = (int32_t)M; c.N = (int32_t)N; c.n_channels = n_channels;
    c.in = fmt_of<IN_TYPE>(); c.out = fmt_of<INT_TYPE>(); c.device = device < 0 ? default_device() : device; c.flags = 0;
    acdsp_fir_desc_t f;
    f.kind = fir_kind; f.ftype = ftype; f.n_taps = n_taps; f.n_channels = n_channels; f.coeffs_per_channel = 0;
    f.in = fmt_of<INT_TYPE>(); f.coeff = fmt_of<COEFF_TYPE>(); f.acc = fmt_of<ACC_TYPE>(); f.out = fmt_of<OUT_TYPE>();
    f.device = c.device; f.flags = 0;
    n_taps_ = n_taps;
    check(acdsp_ddc_create(&c, &f, &h_), "acdsp_ddc_create");
  }
  ~ddc_engine() { if (h_) { acdsp_ddc_destroy(h_); } }
  void set_coeffs(const COEFF_TYPE *c) {
    std::vector<int64_t> r((size_t)n_taps_);
    for (size_t i = 0; i < r.size(); i++) { r[i] = raw_of(c[i]); }
    check(acdsp_ddc_set_coeffs(h_, r.data()), "acdsp_ddc_set_coeffs");
  }
  int64_t out_count(int64_t n_in) { return acdsp_ddc_out_count(h_, n_in); }
  bool fused() { return acdsp_ddc_path(h_) == 1; }
  void run_device(const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride, int64_t *n_out, void *stream = 0) {
    check(acdsp_ddc_run(h_, d_in, in_stride, n_in, d_out, out_stride, n_out, stream), "acdsp_ddc_run");
  }
  void reset() { check(acdsp_ddc_reset(h_), "acdsp_ddc_reset"); }
  acdsp_ddc_t handle() { return h_; }

private:
  ddc_engine(const ddc_engine &);
  ddc_engine &operator=(const ddc_engine &);
  acdsp_ddc_t h_;
  int n_taps_;
};

// ---------------------------------------------------------------------------------------------
// Node level: one FIR bank sharded over the GPUs of a node (acdsp_node_fir_*).  Contiguous channel slices, one engine
// handle + stream + host thread per device, coefficients replicated, no collective.  BASELINE config 4's 8192 channels on 8 GPUs:
//     acdsp::node_fir_engine<IN, OUT, COEFF, ACC> bank(ACDSP_FIR_PROG, ACDSP_SHIFT_REG, 1023, 8192, 8);
//     bank.set_coeffs(c);  bank.run_device(d_in, stride, n, d_out, stride);   // d_in[s] / d_out[s]: shard s's block on device s
// ---------------------------------------------------------------------------------------------
template <class IN_TYPE, class OUT_TYPE, class COEFF_TYPE, class ACC_TYPE>
class node_fir_engine {
public:
  // devices == 0: devices 0 .. n_devices-1; an entry may repeat (concurrent streams of one GPU)
  node_fir_engine(int kind, int ftype, int n_taps, int n_channels, int n_devices, const int32_t *devices = 0, bool coeffs_per_channel = false)
      : h_(0), n_taps_(n_taps), n_ch_(n_channels), per_ch_(coeffs_per_channel) {
    acdsp_fir_desc_t d;
    d.kind = kind; d.ftype = ftype; d.n_taps = n_taps; d.n_channels = n_channels; d.coeffs_per_channel = per_ch_ ? 1 : 0;
    d.in = fmt_of<IN_TYPE>(); d.coeff = fmt_of<COEFF_TYPE>(); d.acc = fmt_of<ACC_TYPE>(); d.out = fmt_of<OUT_TYPE>();
    d.device = 0; d.flags = 0;
    check(acdsp_node_fir_create(&d, n_devices, devices, &h_), "acdsp_node_fir_create");
  }
  ~node_fir_engine() { if (h_) { acdsp_node_destroy(h_); } }
  int n_shards() const { return acdsp_node_n_shards(h_); }
  // channel slice [lo, hi) and device of shard s; `engine` = its ordinary acdsp_fir_t (state get / set, reset, kernel_stats ...)
  void shard(int s, int32_t *device, int64_t *lo, int64_t *hi, acdsp_fir_t *engine = 0) {
    void *e = 0;
    check(acdsp_node_shard_info(h_, s, device, lo, hi, &e, 0), "acdsp_node_shard_info");
    if (engine) { *engine = (acdsp_fir_t)e; }
  }
  void set_coeffs(const COEFF_TYPE *c) {
    std::vector<int64_t> r((size_t)n_taps_ * (per_ch_ ? n_ch_ : 1));
    for (size_t i = 0; i < r.size(); i++) { r[i] = raw_of(c[i]); }
    check(acdsp_node_fir_set_coeffs(h_, r.data()), "acdsp_node_fir_set_coeffs");
  }
  // d_in[s] / d_out[s]: device pointers on shard s's device to its [hi - lo][stride] block; returns when every shard has finished
  void run_device(const void *const *d_in, int64_t in_stride, int64_t n, void *const *d_out, int64_t out_stride) {
    check(acdsp_node_fir_run(h_, d_in, in_stride, n, d_out, out_stride), "acdsp_node_fir_run");
  }
  // dense host block [n_channels][n]
  void run_host(const void *h_in, int64_t n, void *h_out) { check(acdsp_node_fir_run_host(h_, h_in, n, h_out), "acdsp_node_fir_run_host"); }
  // kernel time of the slowest shard in the last run (aggregate rate = n_channels * n / that)
  float last_ms_max() { float m = 0; check(acdsp_node_last_ms(h_, 0, &m), "acdsp_node_last_ms"); return m; }
  acdsp_node_t handle() { return h_; }

private:
  node_fir_engine(const node_fir_engine &);
  node_fir_engine &operator=(const node_fir_engine &);
  acdsp_node_t h_;
  int n_taps_, n_ch_;
  bool per_ch_;
};

}  // namespace acdsp

#endif
