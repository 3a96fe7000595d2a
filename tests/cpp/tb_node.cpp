// tb_node.cpp -- acdsp::node_fir_engine (include/ac_dsp/acdsp_engine.h; C ABI acdsp_node_fir_*): ONE filter bank sharded over the
// GPUs of the node from a plain C++ caller that sees no HIP header.  Three shards (all on the default device when the box has one
// GPU, else spread over the devices); every checked channel must equal a one-channel drop-in ac_fir_load_coeffs object fed the
// same samples, through the device path (per-shard buffers) and through the host path (one dense [channel][time] block).
#include <ac_dsp/ac_fir_load_coeffs.h>

#include <iostream>
#include <vector>

typedef ac_fixed<16, 2, true> T16;
typedef ac_fixed<40, 12, true> ACC;
typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT16;

int main() {
  const int NCH = 29, N = 3000, TAPS = 95, NS = 3;
  std::vector<T16> c(TAPS);
  for (int i = 0; i < TAPS; i++) { c[i] = T16(0.8 * (((i * 41) % 23) - 11) / 128.0); }
  const int nd = acdsp_device_count();
  int32_t devs[NS];
  for (int s = 0; s < NS; s++) { devs[s] = nd >= NS ? s : acdsp::default_device(); }
  acdsp::node_fir_engine<T16, OUT16, T16, ACC> bank(ACDSP_FIR_LOAD, SHIFT_REG, TAPS, NCH, NS, devs);
  bank.set_coeffs(c.data());
  if (bank.n_shards() != NS) { std::cout << "shard count MISMATCH" << std::endl; return 1; }

  // ---- device path: every shard's block lives on its own device ----
  std::vector<void *> d_in(NS), d_out(NS);
  std::vector<const void *> d_in_c(NS);
  std::vector<int16_t> hx((size_t)NCH * N), hy((size_t)NCH * N);
  int64_t covered = 0;
  for (int s = 0; s < NS; s++) {
    int32_t dev = -1;
    int64_t lo = 0, hi = 0;
    bank.shard(s, &dev, &lo, &hi);
    if (lo != covered || dev != devs[s]) { std::cout << "slice MISMATCH" << std::endl; return 1; }
    covered = hi;
    acdsp::check(acdsp_dev_alloc(dev, (uint64_t)(hi - lo) * N * 2, &d_in[s]), "alloc");
    acdsp::check(acdsp_dev_alloc(dev, (uint64_t)(hi - lo) * N * 2, &d_out[s]), "alloc");
    acdsp::check(acdsp_fill_stimulus(dev, d_in[s], 2, hi - lo, N, N, 0xACD5, 16, (uint64_t)lo, 0, 0), "fill");
    acdsp::check(acdsp_sync(dev, 0), "sync");
    d_in_c[s] = d_in[s];
  }
  if (covered != NCH) { std::cout << "slices do not cover the bank" << std::endl; return 1; }
  bank.run_device(d_in_c.data(), N, N, d_out.data(), N);
  for (int s = 0; s < NS; s++) {
    int32_t dev = -1;
    int64_t lo = 0, hi = 0;
    bank.shard(s, &dev, &lo, &hi);
    acdsp::check(acdsp_copy_d2h(dev, &hx[(size_t)lo * N], d_in[s], (uint64_t)(hi - lo) * N * 2), "d2h");
    acdsp::check(acdsp_copy_d2h(dev, &hy[(size_t)lo * N], d_out[s], (uint64_t)(hi - lo) * N * 2), "d2h");
    acdsp::check(acdsp_dev_free(dev, d_in[s]), "free");
    acdsp::check(acdsp_dev_free(dev, d_out[s]), "free");
  }
  int fails = 0;
  std::vector<int16_t> want((size_t)NCH * N);
  for (int ch = 0; ch < NCH; ch++) {
    ac_fir_load_coeffs<T16, OUT16, T16, ACC, TAPS, SHIFT_REG> one;
    ac_channel<T16> in, cch;
    ac_channel<OUT16> out;
    ac_channel<bool> ld;
    for (int i = 0; i < TAPS; i++) { cch.write(c[i]); }
    ld.write(true);
    for (int t = 0; t < N; t++) { in.write(acdsp::from_raw<T16>(hx[(size_t)ch * N + t])); }
    one.run(in, cch, out, ld);
    for (int t = 0; t < N; t++) {
      want[(size_t)ch * N + t] = (int16_t)acdsp::raw_of(out.read());
      if (want[(size_t)ch * N + t] != hy[(size_t)ch * N + t]) { fails++; }
    }
  }
  std::cout << "sharded bank, device path vs per-channel objects: " << (fails ? "MISMATCH" : "identical") << std::endl;

  // ---- host path: a second bank (fresh state), one dense host block ----
  acdsp::node_fir_engine<T16, OUT16, T16, ACC> bank2(ACDSP_FIR_LOAD, SHIFT_REG, TAPS, NCH, NS, devs);
  bank2.set_coeffs(c.data());
  std::vector<int16_t> hy2((size_t)NCH * N);
  bank2.run_host(hx.data(), N, hy2.data());
  int fails2 = 0;
  for (size_t i = 0; i < hy2.size(); i++) { fails2 += hy2[i] != want[i]; }
  std::cout << "sharded bank, host path vs per-channel objects: " << (fails2 ? "MISMATCH" : "identical") << std::endl;
  std::cout << ((fails || fails2) ? "Test FAILED." : "Test PASSED.") << std::endl;
  return (fails || fails2) ? 1 : 0;
}
