// cref_bench.cpp -- "C-ref" CPU baseline: the REFERENCE's own header-only path
// (/root/reference/include/ac_dsp, compiled where it lies) over this repo's ac_types subset, one filter object per
// channel, channels split over the host threads.  Build container only; the numbers go into BASELINE.md.
//   cref_bench [threads=nproc] [samples per channel=65536] [what: both | fir | cic | cfg1]
// cfg1 = BASELINE configs[0] as SURVEY 8(d) fixes it: ac_fir_const_coeffs, 63 taps <16,2,true>, ACC = OUT <38,10>, FOLD_ODD (the ftype of
// the reference's own testbench), ONE channel, the testbench's two-tone stimulus, `samples` samples in calls of 1024 ("CREF cfg1 ...")
// Prebuilt by __graft_entry__.build() into tests/_bin/ (binary only: no reference source travels); bench.py runs `cref_bench T N fir`
// on the GPU node's host cores and reports the "CREF fir <Msamples/s> <threads> <samples> <seconds>" line as
// cpu_baseline_ref_headers (kind "reference-headers-over-own-ac_types").
#include <ac_dsp/ac_fir_const_coeffs.h>
#include <ac_dsp/ac_fir_load_coeffs.h>
#include <ac_dsp/ac_cic_dec_full.h>

#include <cmath>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

typedef ac_fixed<16, 2, true> IN16;
typedef ac_fixed<40, 12, true> ACC40;
typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT16;
typedef ac_fixed<32, 16, true> IN32;
typedef ac_fixed<47, 31, true> INT47;

template <class F> static double timed(int threads, F work) {
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int i = 0; i < threads; i++) { th.emplace_back(work, i); }
  for (auto &t : th) { t.join(); }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

typedef ac_fir_load_coeffs<IN16, OUT16, IN16, ACC40, 255, SHIFT_REG> Fir;
typedef ac_cic_dec_full<IN32, INT47, 8, 1, 5> Cic;
struct FirJob { Fir fir; ac_channel<IN16> in, cch; ac_channel<OUT16> out; ac_channel<bool> ld; };
struct CicJob { Cic cic; ac_channel<IN32> in; ac_channel<INT47> out; };

typedef ac_fixed<38, 10, true> ACC38;
typedef ac_fir_const_coeffs<IN16, ACC38, IN16, ACC38, 63, FOLD_ODD> FirCfg1;

static int run_cfg1(int n) {
  const double pi = 3.14159265358979323846, fc = 75.0 / 500.0, m = 31.0;
  static IN16 c[63];
  double h[63], sum = 0;
  for (int i = 0; i < 63; i++) {
    const double k = i - m, sinc = k == 0 ? 2 * fc : std::sin(2 * pi * fc * k) / (pi * k);
    h[i] = sinc * (0.54 - 0.46 * std::cos(2 * pi * i / 62.0));
    sum += h[i];
  }
  for (int i = 0; i < 63; i++) { c[i] = IN16::from_raw128((__int128)std::llround(h[i] / sum * 16384.0)); }
  for (int i = 0; i < 31; i++) { c[62 - i] = c[i]; }
  std::vector<IN16> x(1024);
  {
    IN16 probe;
    const double tmax = probe.template set_val<AC_VAL_MAX>().to_double();
    std::vector<double> tone(1024);
    double amax = 0;
    for (int i = 0; i < 1024; i++) { tone[i] = std::sin(2 * pi * 25 * i / 500.0) + std::sin(2 * pi * 150 * i / 500.0); amax = std::fabs(tone[i]) > amax ? std::fabs(tone[i]) : amax; }
    for (int i = 0; i < 1024; i++) { x[i] = (tone[i] / amax) * tmax; }
  }
  FirCfg1 fir(c);
  ac_channel<IN16> in;
  ac_channel<ACC38> out;
  long long sink = 0;
  const int calls = (n + 1023) / 1024;
  double dt = 0;
  for (int k = 0; k < calls; k++) {
    for (int i = 0; i < 1024; i++) { in.write(x[i]); }          // stimulus queued outside the timed region
    const auto t0 = std::chrono::steady_clock::now();
    fir.run(in, out);
    dt += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    while (out.available(1)) { sink += gg::raw(out.read()); }
  }
  printf("C-ref ac_fir_const_coeffs 63 taps <16,2,true>, ACC = OUT <38,10>, FOLD_ODD, 1 channel, two-tone stimulus: %d calls x 1024 samples in %.3f s = %.3f Msamples/s\n",
         calls, dt, (double)calls * 1024 / dt / 1e6);
  printf("CREF cfg1 %.6f 1 %d %.3f\n", (double)calls * 1024 / dt / 1e6, calls * 1024, dt);
  printf("(checksum %lld)\n", sink);
  return 0;
}

int main(int argc, char **argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : (int)std::thread::hardware_concurrency();
  const int n = argc > 2 ? atoi(argv[2]) : 65536;
  const std::string what = argc > 3 ? argv[3] : "both";
  if (what == "cfg1") { return run_cfg1(n); }
  std::vector<long long> sink((size_t)threads, 0);
  if (what != "cic") {   // BASELINE configs[1]: ac_fir_load_coeffs, 255 taps, SHIFT_REG; stimulus generation is outside the timed region
    std::vector<FirJob> jobs((size_t)threads);
    timed(threads, [&](int id) {
      uint64_t seed = 0xACD5 + id;
      FirJob &j = jobs[(size_t)id];
      for (int i = 0; i < 255; i++) { j.cch.write(gg::rnd_bits<IN16>(seed, 12)); }
      j.ld.write(true);
      for (int i = 0; i < n; i++) { j.in.write(gg::rnd<IN16>(seed)); }
    });
    const double dt = timed(threads, [&](int id) { FirJob &j = jobs[(size_t)id]; j.fir.run(j.in, j.cch, j.out, j.ld); });
    for (int id = 0; id < threads; id++) { while (jobs[(size_t)id].out.available(1)) { sink[(size_t)id] += gg::raw(jobs[(size_t)id].out.read()); } }
    printf("C-ref ac_fir_load_coeffs 255 taps <16,2>, ACC <40,12>, SHIFT_REG: %d threads x %d samples in %.2f s = %.3f Msamples/s\n",
           threads, n, dt, (double)threads * n / dt / 1e6);
    printf("CREF fir %.6f %d %d %.3f\n", (double)threads * n / dt / 1e6, threads, n, dt);
  }
  if (what != "fir") {   // BASELINE configs[2]: ac_cic_dec_full N5 R8 M1 on <32,16>
    const int nc = n * 16;
    std::vector<CicJob> jobs((size_t)threads);
    timed(threads, [&](int id) {
      uint64_t seed = 0xACD5 + id;
      for (int i = 0; i < nc; i++) { jobs[(size_t)id].in.write(gg::rnd<IN32>(seed)); }
    });
    const double dt = timed(threads, [&](int id) { CicJob &j = jobs[(size_t)id]; j.cic.run(j.in, j.out); });
    for (int id = 0; id < threads; id++) { while (jobs[(size_t)id].out.available(1)) { sink[(size_t)id] += gg::raw(jobs[(size_t)id].out.read()); } }
    printf("C-ref ac_cic_dec_full N5 R8 M1 <32,16> -> <47,31>: %d threads x %d samples in %.2f s = %.3f Msamples/s\n", threads, nc, dt,
           (double)threads * nc / dt / 1e6);
    printf("CREF cic %.6f %d %d %.3f\n", (double)threads * nc / dt / 1e6, threads, nc, dt);
  }
  long long t = 0;
  for (long long v : sink) { t += v; }
  printf("(checksum %lld)\n", t);
  return 0;
}
