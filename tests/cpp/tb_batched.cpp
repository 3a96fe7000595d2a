// tb_batched.cpp -- the many-channel C++ API (acdsp::fir_engine / cic_engine) on device-resident streams,
// using only include/acdsp.h helpers for memory (no HIP headers on the caller's side).  Each channel of the
// batch must equal a one-channel drop-in object fed the same samples.
#include <ac_dsp/ac_fir_load_coeffs.h>
#include <ac_dsp/ac_fir_const_coeffs.h>
#include <ac_dsp/ac_cic_dec_full.h>

#include <iostream>
#include <vector>

typedef ac_fixed<16, 2, true> T16;
typedef ac_fixed<40, 12, true> ACC;
typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT16;

int main() {
  int fails = 0;
  const int NCH = 48, N = 2000, TAPS = 63;
  const int dev = acdsp::default_device();
  // coefficients
  std::vector<T16> c(TAPS);
  for (int i = 0; i < TAPS; i++) { c[i] = T16(0.9 * (((i * 37) % 19) - 9) / 64.0); }
  // device buffers + on-device stimulus
  void *d_in = 0, *d_out = 0;
  acdsp::check(acdsp_dev_alloc(dev, (uint64_t)NCH * N * 2, &d_in), "alloc");
  acdsp::check(acdsp_dev_alloc(dev, (uint64_t)NCH * N * 2, &d_out), "alloc");
  acdsp::check(acdsp_fill_stimulus(dev, d_in, 2, NCH, N, N, 0xACD5, 16, 0, 0, 0), "fill");
  acdsp::fir_engine<T16, OUT16, T16, ACC> batch(ACDSP_FIR_LOAD, SHIFT_REG, TAPS, NCH);
  batch.set_coeffs(c.data());
  batch.run_device(d_in, N, N, d_out, N);
  std::vector<int16_t> hx((size_t)NCH * N), hy((size_t)NCH * N);
  acdsp::check(acdsp_sync(dev, 0), "sync");
  acdsp::check(acdsp_copy_d2h(dev, hx.data(), d_in, hx.size() * 2), "d2h");
  acdsp::check(acdsp_copy_d2h(dev, hy.data(), d_out, hy.size() * 2), "d2h");
  for (int ch = 0; ch < NCH; ch += 13) {
    ac_fir_load_coeffs<T16, OUT16, T16, ACC, TAPS, SHIFT_REG> one;
    ac_channel<T16> in, cch;
    ac_channel<OUT16> out;
    ac_channel<bool> ld;
    for (int i = 0; i < TAPS; i++) { cch.write(c[i]); }
    ld.write(true);
    for (int t = 0; t < N; t++) { in.write(acdsp::from_raw<T16>(hx[(size_t)ch * N + t])); }
    one.run(in, cch, out, ld);
    for (int t = 0; t < N; t++) {
      if (acdsp::raw_of(out.read()) != hy[(size_t)ch * N + t]) { fails++; break; }
    }
  }
  std::cout << "batched FIR vs per-channel objects: " << (fails ? "MISMATCH" : "identical") << std::endl;
  acdsp::check(acdsp_dev_free(dev, d_in), "free");
  acdsp::check(acdsp_dev_free(dev, d_out), "free");

  // ---- DDC cascade (BASELINE config-5 shape): fused kernel on device streams vs the two drop-in classes chained ----
  {
    typedef ac_fixed<16, 1, true> DIN;
    typedef ac_fixed<36, 21, true> DINT;     // find_inter_type_cic_dec for R 16, M 1, N 5 on <16,1>
    typedef ac_fixed<16, 1, true> DCF;
    typedef ac_fixed<60, 30, true> DACC;
    typedef ac_fixed<24, 9, true, AC_RND, AC_SAT> DOUT;
    const int DCH = 5, DN = 16 * (256 * 8 + 40), DT = 127;
    std::vector<DCF> dc(DT);
    for (int i = 0; i < DT; i++) { dc[i] = DCF(0.4 * (((i * 29) % 23) - 11) / 32.0); }
    acdsp::ddc_engine<DIN, DINT, DOUT, DCF, DACC> ddc(16, 1, 5, ACDSP_FIR_CONST, SHIFT_REG, DT, DCH);
    ddc.set_coeffs(dc.data());
    const int64_t no = ddc.out_count(DN);
    void *dd_in = 0, *dd_out = 0;
    acdsp::check(acdsp_dev_alloc(dev, (uint64_t)DCH * DN * 2, &dd_in), "alloc");
    acdsp::check(acdsp_dev_alloc(dev, (uint64_t)DCH * (no + 8) * 4, &dd_out), "alloc");
    acdsp::check(acdsp_fill_stimulus(dev, dd_in, 2, DCH, DN, DN, 0xD0C, 16, 0, 0, 0), "fill");
    int64_t got = 0;
    ddc.run_device(dd_in, DN, DN, dd_out, no + 8, &got);
    std::vector<int16_t> dx((size_t)DCH * DN);
    std::vector<int32_t> dy((size_t)DCH * (no + 8));
    acdsp::check(acdsp_sync(dev, 0), "sync");
    acdsp::check(acdsp_copy_d2h(dev, dx.data(), dd_in, dx.size() * 2), "d2h");
    acdsp::check(acdsp_copy_d2h(dev, dy.data(), dd_out, dy.size() * 4), "d2h");
    int dfails = (got != no || !ddc.fused()) ? 1 : 0;
    for (int ch = 0; ch < DCH; ch += 2) {
      ac_cic_dec_full<DIN, DINT, 16, 1, 5> cic;
      ac_fir_const_coeffs<DINT, DOUT, DCF, DACC, DT, SHIFT_REG> fir(dc.data());
      ac_channel<DIN> in;
      ac_channel<DINT> mid;
      ac_channel<DOUT> out;
      for (int t = 0; t < DN; t++) { in.write(acdsp::from_raw<DIN>(dx[(size_t)ch * DN + t])); }
      cic.run(in, mid);
      fir.run(mid, out);
      for (int64_t t = 0; t < no; t++) {
        if (acdsp::raw_of(out.read()) != dy[(size_t)ch * (no + 8) + t]) { dfails++; break; }
      }
    }
    std::cout << "fused DDC cascade vs chained per-channel objects: " << (dfails ? "MISMATCH" : "identical") << std::endl;
    fails += dfails;
    acdsp::check(acdsp_dev_free(dev, dd_in), "free");
    acdsp::check(acdsp_dev_free(dev, dd_out), "free");
  }
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
