// gen_cfg1.cpp -- BASELINE configs[0] exactly as SURVEY 8(d) fixes it: ac_fir_const_coeffs, 63 taps, IN = COEFF = ac_fixed<16,2,true>,
// ACC = OUT = ac_fixed<38,10> (lossless: 32 + 6 bits), ONE channel, 1024 samples of the two-tone stimulus the reference's own
// testbench drives (tests/rtest_ac_fir_const_coeffs.cpp:126-151: sin(2 pi 25 i / 500) + sin(2 pi 150 i / 500), normalised to the
// largest IN_TYPE value), all six FTYPEs, one run() call.  The stimulus recipe is restated here (double arithmetic + the
// double -> ac_fixed assignment of include/ac_types), the filter is the reference's header where it lies (see common.h).
// Coefficients: a 63-tap Hamming-windowed sinc, cutoff 75 Hz of 500 (passes the 25 Hz tone, stops the 150 Hz one), unit DC gain,
// rounded to <16,2> and symmetrised (so the FOLD_* types see a symmetric set).
// usage: gen_cfg1 <out dir>  ->  <out dir>/fir_cfg1_63.json
#include <ac_dsp/ac_fir_const_coeffs.h>

#include <cmath>

#include "common.h"

using namespace gg;

typedef ac_fixed<16, 2, true> IN_T;
typedef ac_fixed<16, 2, true> CF_T;
typedef ac_fixed<38, 10, true> ACC_T;
typedef ac_fixed<38, 10, true> OUT_T;
static const unsigned kTaps = 63;
static const int kSamples = 1024;

static void coefficients(CF_T (&c)[kTaps], std::vector<long long> &dump) {
  const double pi = 3.14159265358979323846, fc = 75.0 / 500.0, m = (kTaps - 1) / 2.0;
  double h[kTaps], sum = 0;
  for (unsigned i = 0; i < kTaps; i++) {
    const double k = i - m;
    const double sinc = k == 0 ? 2 * fc : std::sin(2 * pi * fc * k) / (pi * k);
    h[i] = sinc * (0.54 - 0.46 * std::cos(2 * pi * i / (kTaps - 1)));
    sum += h[i];
  }
  long long r[kTaps];
  for (unsigned i = 0; i < kTaps; i++) { r[i] = std::llround(h[i] / sum * 16384.0); }   // F = 14
  for (unsigned i = 0; i < kTaps / 2; i++) { r[kTaps - 1 - i] = r[i]; }
  for (unsigned i = 0; i < kTaps; i++) { c[i] = CF_T::from_raw128((__int128)r[i]); dump.push_back(raw(c[i])); }
}

static void stimulus(std::vector<IN_T> &x) {
  const double pi = 3.14159265358979323846, F1 = 25, F2 = 150, Fs = 500;
  IN_T probe;
  const double type_max = probe.template set_val<AC_VAL_MAX>().to_double();
  std::vector<double> tone(kSamples);
  double amax = 0;
  for (int i = 0; i < kSamples; i++) {
    tone[i] = std::sin(2 * pi * F1 * i / Fs) + std::sin(2 * pi * F2 * i / Fs);
    if (std::fabs(tone[i]) > amax) { amax = std::fabs(tone[i]); }
  }
  x.resize(kSamples);
  for (int i = 0; i < kSamples; i++) { x[i] = (tone[i] / amax) * type_max; }   // double -> ac_fixed<16,2>: AC_TRN, AC_WRAP
}

template <FTYPE ft> static void one(Json &j, const CF_T (&c)[kTaps], const std::vector<long long> &cd, const std::vector<IN_T> &x) {
  ac_fir_const_coeffs<IN_T, OUT_T, CF_T, ACC_T, kTaps, ft> dut(c);
  ac_channel<IN_T> in;
  ac_channel<OUT_T> out;
  std::vector<long long> xs, ys, calls;
  for (int i = 0; i < kSamples; i++) { in.write(x[i]); xs.push_back(raw(x[i])); }
  dut.run(in, out);
  calls.push_back(kSamples);
  while (out.available(1)) { ys.push_back(raw(out.read())); }
  j.begin(std::string("const_cfg1_63_") + kFtypeNames[ft]);
  j.str("class", "const");
  j.str("ftype", kFtypeNames[ft]);
  j.num("n_taps", kTaps);
  j.rawjson("in", fmt_json<IN_T>());
  j.rawjson("coeff", fmt_json<CF_T>());
  j.rawjson("acc", fmt_json<ACC_T>());
  j.rawjson("out", fmt_json<OUT_T>());
  j.arr("coeffs", cd); j.num("reload_at", -1);
  j.arr("calls", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  CF_T c[kTaps];
  std::vector<long long> cd;
  coefficients(c, cd);
  std::vector<IN_T> x;
  stimulus(x);
  Json j(dir + "/fir_cfg1_63.json");
  one<SHIFT_REG>(j, c, cd, x);
  one<ROTATE_SHIFT>(j, c, cd, x);
  one<C_BUFF>(j, c, cd, x);
  one<FOLD_EVEN>(j, c, cd, x);
  one<FOLD_ODD>(j, c, cd, x);
  one<TRANSPOSED>(j, c, cd, x);
  return 0;
}
