// engine_ddc.hip -- acdsp_ddc_*: the fused decimator -> FIR cascade behind the C ABI
#include "engine_common.hpp"

using namespace acdsp;
using namespace acdsp::eng;

// ---------------------------------------------------------------------------------------------
// DDC cascade: ac_cic_dec_full -> ac_fir_* on the decimator's lossless INT_TYPE words
// ---------------------------------------------------------------------------------------------
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

