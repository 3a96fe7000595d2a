// tb_batched.cpp -- the many-channel C++ API (acdsp::fir_engine / cic_engine) on device-resident streams,
// using only include/acdsp.h helpers for memory (no HIP headers on the caller's side).  Each channel of the
// batch must equal a one-channel drop-in object fed the same samples.
#include <ac_dsp/ac_fir_load_coeffs.h>
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
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
