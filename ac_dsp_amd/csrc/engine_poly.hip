// engine_poly.hip -- acdsp_polydec_* / acdsp_polyintr_*: ac_poly_dec and ac_poly_intr behind the C ABI
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

// ---------------------------------------------------------------------------------------------
// polyphase decimator
// ---------------------------------------------------------------------------------------------
struct acdsp_polydec {
  acdsp_polydec_desc_t d;
  int in_eb, out_eb, hl;
  bool lossless = false, coeffs_set = false, gen_ok = false;
  bool lossless_shape = false, sat_free = false;   // a saturating ACC_TYPE no partial sum of the current set can reach is a wrapping one (cf. acdsp_fir::sat_free)
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
  h->lossless_shape = fa >= fi + fc && fa - fi - fc < 64;
  h->lossless = h->lossless_shape && desc->acc.O == ACDSP_WRAP;
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
  {
    // |every partial sum of an output| <= sum|c| max|x| over all NTAPS * DF coefficients: inside the symmetric range of a signed saturating
    // ACC_TYPE the saturation never acts (ACDSP_NO_SAT_FREE: A/B knob)
    static const bool no_sat_free = getenv("ACDSP_NO_SAT_FREE") != nullptr;
    bool sf = !no_sat_free && d.acc.O != ACDSP_WRAP && d.acc.S && h->lossless_shape && d.acc.W >= 2 && d.acc.W <= 64;
    if (sf) {
      const int ls = (d.acc.W - d.acc.I) - (d.in.W - d.in.I) - (d.coeff.W - d.coeff.I);
      unsigned __int128 sa = 0;
      for (int i = 0; i < n; i++) { sa += (unsigned __int128)(coeffs[i] < 0 ? -(__int128)coeffs[i] : (__int128)coeffs[i]); }
      const unsigned __int128 xmax = d.in.S ? ((unsigned __int128)1 << (d.in.W - 1)) : (((unsigned __int128)1 << d.in.W) - 1);
      const unsigned __int128 top = ((unsigned __int128)1 << (d.acc.W - 1)) - 1;
      unsigned __int128 b = sa * xmax;
      sf = (sa == 0 || b / sa == xmax) && (ls == 0 || (b >> (127 - ls)) == 0) && (b << ls) <= top;
    }
    h->sat_free = sf;
    h->lossless = h->lossless_shape && (d.acc.O == ACDSP_WRAP || sf);
  }
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
  if (h->sat_free) { k.acc.O = ACDSP_WRAP; }
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
// polyphase interpolator (second half): ac_poly_intr
// ---------------------------------------------------------------------------------------------
struct acdsp_polyintr {
  acdsp_polyintr_desc_t d;
  int in_eb, out_eb, hl;
  bool ctrl_set = false;
  void *d_hist[2] = {nullptr, nullptr};
  int64_t *d_saved[2] = {nullptr, nullptr};   // sums of the last sample, emitted by the next call (folded cores)
  SideStream side;                            // head / tail kernels of a call beside its matrix-core kernel (fir_kernels.hpp)
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
  h->side.destroy();
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
  bool forked = false;
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
      // the head and the tail of the call (two launches of a few thousand waves, 16 us each, + the saved sums) neither feed nor follow the
      // matrix-core kernel: forked here, they overlap it on the handle's side stream (4.7 % of the bench row when they queued behind it)
      if (SideStream::enabled()) {
        const hipError_t ef = h->side.fork(s);
        if (ef != hipSuccess) { return fail(ACDSP_EHIP, "poly_intr side stream: %s", hipGetErrorString(ef)); }
        forked = true;
      }
      FirParams k;
      memset(&k, 0, sizeof k);
      k.n_ch = d.n_channels; k.in = p.in; k.cf = p.cf; k.acc = p.acc; k.out = p.out; k.in_eb = h->in_eb; k.out_eb = h->out_eb;
      k.lossless_shift = p.lossless_shift; k.in_stride = in_stride; k.out_stride = out_stride; k.n = n_in; k.x = d_in; k.y = d_out;
      e = launch_fir_up(k, h->up_plan, h->up_px, h->d_upfrag, h->d_upcorr, 0, 0, 0, h->up_shmask, h->up_max_abs, slot_a, n_steps, out_off, s);
      if (e == hipSuccess) {
        o_a = 16 * slot_a * L + out_off; o_b = 16 * (slot_a + 32 * n_steps) * L + out_off;
        h->last_path = ACDSP_PATH_MFMA_GEN;
      } else if (e != hipErrorNotSupported) {
        if (forked) { (void)h->side.join(s); }   // (an unjoined fork invalidates a stream capture and leaves the side stream waiting: advisor, round 5)
        return fail(ACDSP_EHIP, "poly_intr matrix-core kernel launch failed: %s", hipGetErrorString(e));
      }
    }
  }
  hipStream_t vs = (forked && o_b > o_a) ? h->side.s : s;
  if (o_b > o_a) {
    p.o_begin = 0; p.o_end = o_a;
    e = launch_polyintr(p, nullptr, vs);
    if (e == hipSuccess) { p.o_begin = o_b; p.o_end = no; e = launch_polyintr(p, h->d_saved[h->cur ^ 1], vs); }
  } else {
    e = launch_polyintr(p, h->d_saved[h->cur ^ 1], s);
  }
  hipError_t eh = hipSuccess;
  if (e == hipSuccess) {
    FirParams k;
    memset(&k, 0, sizeof k);
    k.n_ch = d.n_channels; k.in = p.in; k.in_eb = h->in_eb; k.hl = h->hl; k.in_stride = in_stride; k.n = n_in; k.x = d_in; k.hist = p.hist;
    eh = launch_fir_hist_update(k, h->d_hist[h->cur ^ 1], vs);   // (always the other buffer: the saved sums flip with it -- so it may run beside the main kernel too)
  }
  if (forked) {   // (always joined, used or not: an unjoined fork is an error under stream capture)
    const hipError_t ej = h->side.join(s);
    if (e == hipSuccess && ej != hipSuccess) { e = ej; }
  }
  if (e != hipSuccess) { return fail(ACDSP_EHIP, "poly_intr kernel launch failed: %s", hipGetErrorString(e)); }
  if (eh != hipSuccess) { return fail(ACDSP_EHIP, "poly_intr state kernel launch failed: %s", hipGetErrorString(eh)); }
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

