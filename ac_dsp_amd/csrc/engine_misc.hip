// engine_misc.hip -- acdsp_intgdump_* / acdsp_mvavg_*: ac_intg_dump and ac_mv_avg behind the C ABI
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

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
  bool temp_zero = true;        // d_temp[cur] is known to be all zero (create / reset / zeroed behind a general-kernel call that dumped everything):
                                // what the tile / stream kernels rely on when they leave temp[] alone (advisor, round 5)
  int last_path = 0;            // acdsp_intgdump_path
  // block table of the last call: a stream that dumps on a fixed schedule passes the same n_sample[] every call, and then
  // neither the table is rebuilt nor uploaded and run() stays asynchronous (no stream synchronisation)
  std::vector<int64_t> last_ns;
  void *last_stream = nullptr;
  int64_t tbl_grp = 0, tbl_uni_rounds = 0, tbl_max_rounds = 0;
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
    h->tbl_max_rounds = 0;
    for (int64_t b = 0; b < n_blocks; b++) { if (blk[(size_t)(n_blocks + b)] > h->tbl_max_rounds) { h->tbl_max_rounds = blk[(size_t)(n_blocks + b)]; } }
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
  // A saturating ACC_TYPE whose bounds no block sum of this call can reach is a wrapping one (round 5; cf. acdsp_fir::sat_free): every block
  // dumps and nothing is carried in, so a sum has at most max-rounds terms of at most max|x| each.
  bool sat_free = false;
  if (d.acc.O != ACDSP_WRAP && !h->pending && grp == n_blocks && p.acc.F >= p.in.F && p.acc.F - p.in.F < 64 - d.in.W && (d.acc.S || !d.in.S)) {
    static const bool no_sat_free = getenv("ACDSP_NO_SAT_FREE") != nullptr;   // A/B knob
    const unsigned __int128 xmax = d.in.S ? ((unsigned __int128)1 << (d.in.W - 1)) : (((unsigned __int128)1 << d.in.W) - 1);
    const unsigned __int128 top = d.acc.S ? (((unsigned __int128)1 << (d.acc.W - 1)) - 1) : (((unsigned __int128)1 << d.acc.W) - 1);
    sat_free = !no_sat_free && (((unsigned __int128)h->tbl_max_rounds * xmax) << (p.acc.F - p.in.F)) <= top;   // rounds < 2^63, xmax <= 2^64 - W .. : inside 128 bits
  }
  p.lossless = (d.acc.O == ACDSP_WRAP || sat_free) && p.acc.F >= p.in.F && p.acc.F - p.in.F < 64 - d.in.W;
  p.tile_ok = p.lossless && !h->pending && h->temp_zero && grp == n_blocks;
  if (p.tile_ok) { p.uni_rounds = h->tbl_uni_rounds; }
  p.x = d_in; p.y = d_out; p.temp = h->d_temp[h->cur];
  p.blk_off = h->d_blk; p.blk_rounds = h->d_blk + n_blocks; p.blk_out = h->d_blk + 2 * n_blocks; p.blk_chain = h->d_chain;
  bool temp_written = true;
  hipError_t e = launch_intg_dump(p, h->d_temp[h->cur ^ 1], s, &temp_written, &h->last_path);
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "intg_dump kernel launch failed: %s", hipGetErrorString(e)); }
  if (temp_written) { h->cur ^= 1; }   // (else: nothing carried in, every block dumped -- the all-zero temp[] of this side stays the state)
  h->pending = start != (int32_t)n_blocks;   // the call ended on blocks that did not dump: their sums sit in temp[]
  if (temp_written) {
    // the general kernel wrote the next temp[]: non-zero while sums are pending; when every chain dumped it is zeroed HERE rather than trusted
    h->temp_zero = false;
    if (!h->pending) {
      HIP_TRY(hipMemsetAsync(h->d_temp[h->cur], 0, (size_t)d.n_objects * d.chn * sizeof(int64_t), s));
      h->temp_zero = true;
    }
  }
  return ACDSP_OK;
}

int32_t acdsp_intgdump_path(acdsp_intgdump_t h) { return h ? h->last_path : -1; }

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
  h->temp_zero = true;
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
  uint32_t *d_frag = nullptr;   // matrix-core form of the streaming kernel: fragments of the coefficient set (mv_avg_build_frags)
  int frag_nb = 0;
  int64_t frag_csum = 0;
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
  if (hipMalloc((void **)&h->d_coeffs, (size_t)desc->taps * sizeof(int64_t)) != hipSuccess ||
      hipMalloc((void **)&h->d_frag, (size_t)kMvAvgFragWords * sizeof(uint32_t)) != hipSuccess) {
    if (h->d_coeffs) { (void)hipFree(h->d_coeffs); }
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
  if (h->d_frag) { (void)hipFree(h->d_frag); }
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
  {
    std::vector<uint32_t> fr((size_t)kMvAvgFragWords, 0u);
    h->frag_nb = mv_avg_build_frags(coeffs, h->d.taps, h->d.win_mode, fr.data(), &h->frag_csum);
    if (h->frag_nb > 0) { HIP_TRY(hipMemcpy(h->d_frag, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice)); }
  }
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
  // A saturating ACC_TYPE that cannot saturate is a wrapping one (cf. acdsp_fir::sat_free): the cast of a sample is exact, and every partial sum
  // is bounded by sum|c| max|x| 2^d / 2^sh plus one LSB per tap, inside the type's symmetric range.  The kernels then see AC_WRAP.
  if (d.acc.O != ACDSP_WRAP && d.acc.S && !h->h_coeffs.empty() && !p.force_generic && (d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && d.acc.W <= 64) {
    static const bool no_sat_free = getenv("ACDSP_NO_SAT_FREE") != nullptr;   // A/B knob
    const int dc = p.acc.F - p.in.F, sh = p.cf.F, i_in = d.in.I + (d.in.S ? 0 : 1);
    if (!no_sat_free && dc >= 0 && dc < 40 && sh >= 0 && sh < 64 && d.acc.I >= i_in) {
      unsigned __int128 sa = 0;
      for (int64_t c : h->h_coeffs) { sa += (unsigned __int128)(c < 0 ? -(__int128)c : (__int128)c); }
      const unsigned __int128 xmax = d.in.S ? ((unsigned __int128)1 << (d.in.W - 1)) : (((unsigned __int128)1 << d.in.W) - 1);
      const unsigned __int128 top = ((unsigned __int128)1 << (d.acc.W - 1)) - 1;
      unsigned __int128 b = sa * xmax;                     // sa < 2^74 (1025 taps of 64 bits), xmax <= 2^64: may leave 128 bits
      if (sa == 0 || b / sa == xmax) {
        if ((b >> (127 - dc)) == 0) {
          b = ((b << dc) >> sh) + (unsigned __int128)h->h_coeffs.size() + 1;
          if (b <= top) { p.acc.O = ACDSP_WRAP; }
        }
      }
    }
  }
  // order-free class: products (ACC_TYPE) w[j] * coeffs[j] inside 2^62.  The cast sample has at most min(W_acc, W_in + max(F_acc - F_in, 0))
  // bits and the coefficients as many as the set on the handle needs (round 5: the bound by W_acc + W_coeff sent <32,16> samples into a
  // <56,30> accumulator to the 128-bit per-tap kernel)
  int cbits = d.coeff.W;
  if (!h->h_coeffs.empty()) {
    cbits = 1;
    for (int64_t c : h->h_coeffs) {
      const uint64_t m = (uint64_t)(c < 0 ? ~c : c);
      int b = 1;
      while (b < 64 && (m >> (b - 1)) != 0) { b++; }
      if (b > cbits) { cbits = b; }
    }
    if (!d.coeff.S) { cbits++; }
  }
  const int dcast = p.acc.F - p.in.F, xbits_in = d.in.W + (d.in.S ? 0 : 1) + (dcast > 0 ? dcast : 0);
  const int xbits = xbits_in < d.acc.W + (d.acc.S ? 0 : 1) ? xbits_in : d.acc.W + (d.acc.S ? 0 : 1);
  p.fast = !p.force_generic && p.acc.O == ACDSP_WRAP && (d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && p.cf.F >= 0 &&
           p.cf.F < 62 && xbits + cbits <= 62;
  p.n_sample = n_sample; p.n_frames = n_frames; p.out_per_frame = opf; p.in_stride = in_stride; p.out_stride = out_stride;
  p.x = d_in; p.y = d_out; p.coeffs = h->d_coeffs; p.h_coeffs = h->h_coeffs.data();
  p.frag = h->frag_nb > 0 ? h->d_frag : nullptr; p.frag_nb = h->frag_nb; p.frag_csum = h->frag_csum;
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

