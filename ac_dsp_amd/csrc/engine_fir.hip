// engine_fir.hip -- acdsp_fir_*: the FIR classes (reference ac_fir_{const,load,prog}_coeffs.h, ac_fir_reg_share.h) behind the C ABI
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

// ACC_TYPE as the kernels see it: a saturating accumulator that cannot saturate for the handle's coefficient set is a wrapping one
static inline DFmt fir_acc_fmt(const acdsp_fir *h) {
  DFmt a = make_dfmt(h->d.acc);
  if (h->sat_free) { a.O = ACDSP_WRAP; }
  return a;
}

// ---------------------------------------------------------------------------------------------
// FIR
// ---------------------------------------------------------------------------------------------
namespace acdsp {
namespace eng {
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
}  // namespace eng
}  // namespace acdsp

namespace {

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
  bool lossless = fa >= fi + fc && fa - fi - fc < 64 && (!h->use_rt || h->rt_hybrid) && !h->wide;   // (the overflow mode: below)
  const int ift = internal_ftype(desc->kind, desc->ftype);
  if (is_fold_odd(ift)) {
    // the ACC_TYPE `fold` must also keep every fraction bit of the pre-add (fc < 0 would let fa >= fi + fc pass with fa < fi)
    lossless = lossless && fa >= fi;
    // the ACC_TYPE `fold` variable must hold x[i] +/- x[N-1-i] without wrapping (a difference needs a signed type)
    int need_i = desc->in.I + 1 + ((desc->acc.S && !desc->in.S) ? 1 : 0);
    lossless = lossless && desc->acc.I >= need_i && (desc->acc.S || (!desc->in.S && ift != kRsFoldOddAnti));
  }
  h->lossless_shape = lossless;
  h->sat_free = false;
  h->lossless = lossless && desc->acc.O == ACDSP_WRAP;
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
  if (trace_handles()) { fprintf(stderr, "[acdsp] fir_create kind=%d ftype=%d n_taps=%d n_channels=%d\n", desc->kind, desc->ftype, desc->n_taps, desc->n_channels); }
  return ACDSP_OK;
}

int32_t acdsp_fir_destroy(acdsp_fir_t h) {
  if (!h) { return ACDSP_OK; }
  if (trace_handles()) { fprintf(stderr, "[acdsp] fir_destroy kernel_runs=%lld\n", (long long)h->n_runs); }
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
  delete h;
  return ACDSP_OK;
}

// rt_hybrid: reg_trans[] of every channel from the input history and the coefficients in d_coeffs (the reference's recurrence unrolled
// in time: fir_rt_update_kernel with the history as a call of hl >= n_taps samples, so that no older partial sum enters).  Synchronous.
}  // extern "C" (C++ linkage: the state blobs of engine.hip call it)
namespace acdsp {
namespace eng {
int32_t fir_rt_from_hist(acdsp_fir *h) {
  const acdsp_fir_desc_t &d = h->d;
  FirParams k;
  memset(&k, 0, sizeof k);
  k.n_taps = d.n_taps; k.ftype = internal_ftype(d.kind, d.ftype); k.n_ch = d.n_channels; k.coeffs_per_channel = d.coeffs_per_channel;
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out);
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
}  // namespace eng
}  // namespace acdsp
extern "C" {

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
  {
    // Saturating accumulators: |every partial sum, in any order| <= sum|c| max|x| (the folds' pre-added pairs included: effective_coeffs lists
    // both taps of a pair), plus one LSB per tap where a tap's product is quantised.  Inside the type's SYMMETRIC range the three saturating
    // modes never act and equal AC_WRAP.  Signed accumulators, no carried partial sums (reg_trans), at most 64 bits.  ACDSP_NO_SAT_FREE: A/B knob.
    static const bool no_sat_free = getenv("ACDSP_NO_SAT_FREE") != nullptr;
    const int fi_ = d.in.W - d.in.I, fc_ = d.coeff.W - d.coeff.I, fa_ = d.acc.W - d.acc.I;
    bool sf = !no_sat_free && d.acc.O != ACDSP_WRAP && d.acc.S && !h->wide && !h->use_rt && d.acc.W >= 2 && d.acc.W <= 64 &&
              (fa_ >= fi_ + fc_ ? fa_ - fi_ - fc_ < 64 : ((d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && fi_ + fc_ - fa_ < 64));
    const int ift_ = internal_ftype(d.kind, d.ftype);
    if (sf && is_fold_odd(ift_)) {   // the `fold` pre-add variable is an ACC_TYPE too
      const int need_i = d.in.I + 1 + (!d.in.S ? 1 : 0);
      sf = fa_ >= fi_ && d.acc.I >= need_i;
    }
    const unsigned __int128 xmax = d.in.S ? ((unsigned __int128)1 << (d.in.W - 1)) : (((unsigned __int128)1 << d.in.W) - 1);
    const unsigned __int128 top = ((unsigned __int128)1 << (d.acc.W - 1)) - 1;
    for (size_t st = 0; st < n_sets && sf; st++) {
      const std::vector<int64_t> eff = effective_coeffs(coeffs + st * d.n_taps, d.n_taps, ift_);
      unsigned __int128 sa = 0;
      for (int64_t v : eff) { sa += (unsigned __int128)(v < 0 ? -(__int128)v : (__int128)v); }
      unsigned __int128 b = sa * xmax;                       // < 2^64 * 2^64 would overflow: W_in, W_coeff <= 64 but sums of 2^10 taps -- guard
      if (sa != 0 && b / sa != xmax) { sf = false; break; }
      if (fa_ >= fi_ + fc_) {
        const int ls = fa_ - fi_ - fc_;
        if (ls > 0 && (b >> (127 - ls)) != 0) { sf = false; break; }
        b <<= ls;
      } else {
        b = (b >> (fi_ + fc_ - fa_)) + (unsigned __int128)eff.size() + 1;
      }
      sf = b <= top;
    }
    h->sat_free = sf;
    h->lossless = h->lossless_shape && (d.acc.O == ACDSP_WRAP || sf);
  }
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
    bool ok = true;
    // cshift: the sets go through the kernel scaled by 2^cshift (see below); one attempt with the shift the formats ask for, one without
    auto build = [&](int cshift) {
      std::fill(frag.begin(), frag.end(), 0u);
      memset(&worst, 0, sizeof worst);
      ok = true;
      for (size_t st = 0; st < n_sets && ok; st++) {
        std::vector<int64_t> eff = effective_coeffs(coeffs + st * d.n_taps, d.n_taps, internal_ftype(d.kind, d.ftype));
        for (int64_t &v : eff) { v = (int64_t)((uint64_t)v << cshift); }
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
    };
    // Narrow types (<8,1> samples and coefficients into an <8,1> output: 7 dropped bits) leave the 32-bit epilogue classes less than one bit
    // to shift by once the 16 - W_out bits of the packed finish are taken off (fir_mfma_epilogue_class: rse >= 1), and ran the generic
    // epilogue with element-wise stores (0.18 of the roofline).  The sum is linear in the set: the fragments are built from c << cshift and the
    // kernel is told F_coeff + cshift -- every bound scales with it, and the classes are asked again with the scaled plan.
    h->mfma_cshift = 0;
    {
      const int fi_ = d.in.W - d.in.I, fc_ = d.coeff.W - d.coeff.I, fo_ = d.out.W - d.out.I;
      const int nar_d = (h->out_eb == 2 && d.out.W >= 2 && d.out.W < 16) ? 16 - d.out.W : 0, rse = fi_ + fc_ - fo_ - nar_d;
      static const bool no_cshift = getenv("ACDSP_NO_CSHIFT") != nullptr;   // A/B knob
      const int want = (h->out_eb == 2 && rse < 1 && rse > -14 && !no_cshift) ? 1 - rse : 0;
      if (want > 0) {
        build(want);
        FirParams k;
        memset(&k, 0, sizeof k);
        k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out);
        k.cf.F += want;
        k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
        const int epi = ok ? fir_mfma_epilogue_class(k, worst) : 0;
        if (epi == 1 || epi == 2) { h->mfma_cshift = want; }
      }
      if (!h->mfma_cshift) { build(0); }
    }
    if (ok && d.coeffs_per_channel && worst.nb > fir_mfma_max_reg_blocks()) {
      // a set per channel needs the register-resident kernels: beyond 9 K-blocks only band-limited sets with the fast int16 epilogue
      FirParams k;
      memset(&k, 0, sizeof k);
      k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out);
      k.cf.F += h->mfma_cshift;
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
  if (!h->wide) { kq.acc = fir_acc_fmt(h); kq.out = make_dfmt(d.out); }
  kq.in_eb = h->in_eb; kq.out_eb = h->out_eb; kq.hl = h->hl; kq.use_rt = (h->use_rt && !h->rt_hybrid) ? 1 : 0;
  kq.lossless_shift = kq.acc.F - kq.in.F - kq.cf.F;
  static const bool no_lz = getenv("ACDSP_NO_MFMA_LOSSY") != nullptr;        // A/B knob: class B stays on the VALU kernels
  // 16-bit types: fir_lossy_kernel (int32 VALU, 12 instructions per tap and four outputs) serves them too; the matrix-core kernel needs 6
  // (measured 1.7 - 1.8 x faster at 63 / 127 taps, profiles/r5_shapes.txt) and goes first; ACDSP_LOSSY16_FIRST restores round 4's choice (A/B knob)
  static const bool lz_first = getenv("ACDSP_LOSSY16_FIRST") == nullptr;
  {
    const int ift = internal_ftype(d.kind, d.ftype);
    const int fi = kq.in.F, fc = kq.cf.F, fa = kq.acc.F, sbits = fi + fc - fa;
    const bool fold_odd = is_fold_odd(ift), fold_even = ift == ACDSP_FOLD_EVEN || ift == kRsFoldEven || ift == kRsFoldEvenAnti;
    const bool anti = ift == kRsFoldEvenAnti || ift == kRsFoldOddAnti;
    bool ok = !no_lz && !no_gen && !h->wide && !h->lossless && !h->use_rt && !(d.flags & ACDSP_FLAG_FORCE_GENERIC) && !d.coeffs_per_channel &&
              (d.acc.O == ACDSP_WRAP || h->sat_free) && (d.acc.Q == ACDSP_TRN || d.acc.Q == ACDSP_RND) && d.acc.S && d.acc.W <= 64 && sbits >= 1 && sbits <= 15 &&
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
  k.hist_next = nullptr; k.t_begin = 0; k.in_flip = 0;
  k.n_taps = d.n_taps; k.ftype = internal_ftype(d.kind, d.ftype); k.n_ch = d.n_channels; k.coeffs_per_channel = d.coeffs_per_channel;
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff);
  if (h->wide) { memset(&k.acc, 0, sizeof k.acc); memset(&k.out, 0, sizeof k.out); k.acc.F = d.acc.W - d.acc.I; }
  else { k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out); }
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
    // unsigned 16-bit samples: the kernel flips the top bit of every sample as it splits the rows and the history into byte planes
    // (FirParams::in_flip; round 4 and the first half of round 5 wrote a flipped image of both first: a second pass over the call's input,
    // 0.21 - 0.33 of the roofline where the signed types run at 0.55)
    k.in_flip = 1;
    k.in.S = 1; k.in.lo = -32768; k.in.hi = 32767;
  }
  if (path == ACDSP_PATH_MFMA_I8 || path == ACDSP_PATH_MFMA_GEN || path == ACDSP_PATH_MFMA_LOSSY) {
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
                                                (path == ACDSP_PATH_MFMA_I8 && !h->mfma_cshift && fir_mfma_register_resident(k, h->plan)));   // single-wave workgroups
  const int nxt_fused = hist_next_index(h->cur, false);
  if (fuse_hist) { k.hist_next = h->d_hist[nxt_fused]; }
  if (!small) { HIP_TRY(hipEventRecord(h->tm.start(), s)); }
  hipError_t e;
  if (path == ACDSP_PATH_MFMA_I8) {
    FirParams km = k;    // (the fragments hold c << mfma_cshift: the kernel's shifts follow; the state kernels below keep the handle's formats)
    km.cf.F += h->mfma_cshift; km.lossless_shift -= h->mfma_cshift;
    e = launch_fir_mfma(km, h->plan, d.coeffs_per_channel, h->d_frag, h->d_corr, s);
  }
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
  h->n_runs++;
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

int32_t acdsp_fir_mfma_epilogue(acdsp_fir_t h) {
  if (!h || !h->coeffs_set || h->path != ACDSP_PATH_MFMA_I8) { return -1; }
  const acdsp_fir_desc_t &d = h->d;
  FirParams k;
  memset(&k, 0, sizeof k);
  k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out);
  k.cf.F += h->mfma_cshift;
  k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
  return fir_mfma_epilogue_class(k, h->plan) | (h->mfma_cshift << 8) | (h->in_flip ? 1 << 16 : 0);
}

int32_t acdsp_fir_mfma_issued(acdsp_fir_t h, int32_t *per_1024_samples) {
  if (!h || !per_1024_samples) { return fail(ACDSP_EINVAL, "null argument"); }
  if (!h->coeffs_set) { return fail(ACDSP_ESTATE, "acdsp_fir_mfma_issued before acdsp_fir_set_coeffs"); }
  *per_1024_samples = 0;
  if (h->path == ACDSP_PATH_MFMA_I8) {
    const acdsp_fir_desc_t &d = h->d;
    FirParams k;
    memset(&k, 0, sizeof k);
    k.in = make_dfmt(d.in); k.cf = make_dfmt(d.coeff); k.acc = fir_acc_fmt(h); k.out = make_dfmt(d.out);
    k.cf.F += h->mfma_cshift;
    k.in_eb = h->in_eb; k.out_eb = h->out_eb; k.lossless_shift = k.acc.F - k.in.F - k.cf.F;
    *per_1024_samples = fir_mfma_issued_per_step(k, h->plan);
  }
  return ACDSP_OK;
}

}  // extern "C"

