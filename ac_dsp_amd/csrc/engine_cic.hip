// engine_cic.hip -- acdsp_cic_*: ac_cic_dec_full / ac_cic_intr_full behind the C ABI
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

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
  h->warm = h->hl;
  {
    // two-stage decimator (cic2.hip) where R = R1 R2 with a compiled stage-1 rate: its chunk 0 starts `wu` steps of 256 R1 inputs early,
    // so the handle keeps that much input history (the samples beyond the filter memory only ever feed warm-up values the combs cancel)
    // (below R = 32 the one-stage FIR identity serves the rates it has a plan for; where it has none -- R M N too many taps, e.g. R 24 M 2 N 4 -- the
    // two-stage kernel takes over there as well instead of the recurrence kernel)
    bool one_stage = false;
    if (!desc->interp && !h->wide && desc->R < 32 && (desc->in.W + (desc->in.S ? 0 : 1) + 7) / 8 <= h->in_eb && getenv("ACDSP_NO_GEN") == nullptr) {
      const int L = desc->R * h->me;
      std::vector<uint64_t> c(1, 1);
      for (int st = 0; st < desc->N; st++) {
        std::vector<uint64_t> nx(c.size() + L - 1, 0);
        for (size_t i = 0; i < c.size(); i++) { for (int j = 0; j < L; j++) { nx[i + j] += c[i]; } }
        c.swap(nx);
      }
      std::vector<int64_t> t((size_t)desc->N - 1, 0);
      for (uint64_t v : c) { t.push_back((int64_t)v); }
      FirGenPlan probe;
      std::vector<uint32_t> fr;
      one_stage = fir_gen_plan(t.data(), (int)t.size(), desc->R, 15, &probe, &fr) && fir_gen_plan(t.data(), (int)t.size(), desc->R, 0, &probe, &fr);
    }
    static const bool no_c2 = getenv("ACDSP_NO_CIC2") != nullptr;         // A/B knob: recurrence kernel as before
    static const bool c2_all = getenv("ACDSP_CIC2_ALL") != nullptr;       // A/B knob: also where the one-stage FIR identity fits (R < 32)
    const int in_bits = desc->in.W + (desc->in.S ? 0 : 1);
    if (!desc->interp && !h->wide && !no_c2 && (desc->R >= 32 || c2_all || !one_stage) && (in_bits + 7) / 8 <= h->in_eb && !(desc->flags & ACDSP_FLAG_FORCE_GENERIC) &&
        cic2_factor(h->in_eb, desc->R, h->me, desc->N, &h->c2_R1, &h->c2_R2, &h->c2_wu)) {
      cic2_stage1_taps(h->c2_R1, desc->N, &h->c2_taps);
      FirGenPlan probe;
      std::vector<uint32_t> fr;
      h->c2_ok = fir_gen_plan(h->c2_taps.data(), (int)h->c2_taps.size(), h->c2_R1, 15, &probe, &fr) &&
                 fir_gen_plan(h->c2_taps.data(), (int)h->c2_taps.size(), h->c2_R1, 0, &probe, &fr);
      if (h->c2_ok) {
        const int need = round_up(cic2_hist_len(h->in_eb, h->c2_R1, desc->N, h->c2_wu), kCicTile);
        if (need > h->hl) { h->hl = need; }
      }
    }
  }
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
    if (h->c2_ok && e == hipSuccess) { e = hipMalloc((void **)&h->d_c2frag, (size_t)16 * 3 * 8 * 64 * 4 * sizeof(uint32_t)); }
  }
  if (e != hipSuccess || h->tm.init() != ACDSP_OK) {
    acdsp_cic_destroy(h);
    return fail(ACDSP_EHIP, "CIC state allocation failed: %s", hipGetErrorString(e));
  }
  *out = h;
  if (trace_handles()) { fprintf(stderr, "[acdsp] cic_create interp=%d R=%d M=%d N=%d n_channels=%d\n", desc->interp, desc->R, desc->M, desc->N, desc->n_channels); }
  return ACDSP_OK;
}

int32_t acdsp_cic_destroy(acdsp_cic_t h) {
  if (!h) { return ACDSP_OK; }
  if (trace_handles()) { fprintf(stderr, "[acdsp] cic_destroy kernel_runs=%lld\n", (long long)h->tm.count); }
  (void)hipSetDevice(h->d.device);
  if (h->d_gfrag) { (void)hipFree(h->d_gfrag); }
  if (h->d_c2frag) { (void)hipFree(h->d_c2frag); }
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
  p.hl = h->hl; p.warm_tiles = h->warm / kCicTile;
  p.t_from = 0; p.emit_from = 0;
  p.vec_ok = ((uintptr_t)d_in % 16 == 0) && ((in_stride * h->in_eb) % 16 == 0);
  p.out_simple = (p.out.F == p.in.F && p.out.O == ACDSP_WRAP) ? ((p.out.S && p.out.W >= h->it.W) ? 2 : 1) : 0;
  p.in_stride = in_stride; p.out_stride = out_stride; p.n_in = n_in;
  p.x = d_in; p.y = d_out; p.hist = h->d_hist[h->cur];
  // chunking: aim at >= 4096 waves, keep the warm-up below ~6 % of a chunk
  const int64_t groups = (d.n_channels + 63) / 64;
  int64_t chunk = (n_in * groups + 4095) / 4096;
  const int64_t floor_chunk = (int64_t)16 * h->warm > 1024 ? (int64_t)16 * h->warm : 1024;
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
  // ... or in two stages (R = R1 R2, cic2.hip) where the one-stage window does not fit: complete chunks there, the ragged end on the recurrence kernel
  bool use_c2 = h->c2_ok && !use_gen && p.vec_ok && in_stride >= (n_in + 15) / 16 * 16;
  const uint32_t *c2frag = nullptr;
  if (use_c2) {
    const int fm = (int)(p.first % 16);
    if (!h->c2_have[fm]) {
      std::vector<uint32_t> fr;
      if (!fir_gen_plan(h->c2_taps.data(), (int)h->c2_taps.size(), h->c2_R1, fm, &h->c2_plan[fm], &fr)) { use_c2 = false; }
      else {
        HIP_TRY(hipMemcpyAsync(h->d_c2frag + (size_t)fm * 3 * 8 * 64 * 4, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));   // fr is a stack vector
        h->c2_have[fm] = true;
      }
    }
    if (use_c2) { c2frag = h->d_c2frag + (size_t)fm * 3 * 8 * 64 * 4; }
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
    int64_t covered = 0;
    e = hipSuccess;
    if (use_c2) { e = launch_cic2(p, h->c2_plan[p.first % 16], c2frag, h->c2_R1, h->c2_R2, h->c2_wu, no, s, &covered); }
    if (e == hipSuccess && covered < no) {
      if (covered > 0) {            // the ragged end of the call: outputs from `covered` on, i.e. emitting samples from first + covered R on
        p.emit_from = p.first + covered * d.R;
        p.t_from = p.emit_from / kCicTile * kCicTile;
        p.chunk = (floor_chunk + kCicTile - 1) / kCicTile * kCicTile;      // a short rest: many short chunks
      }
      e = launch_cic(p, s);
      p.t_from = 0; p.emit_from = 0;
    }
    if (covered > 0) { h->last_path = ACDSP_PATH_CIC_2STAGE; }
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

