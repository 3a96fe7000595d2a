// tb_polydec.cpp -- C++ testbench for the drop-in ac_poly_dec class template (own code; the reference ships no
// testbench for this block).  The expected stream is computed here with the ac_fixed templates of
// include/ac_types by the loop nest the reference documents (ac_poly_dec.h:109-128), i.e. a third,
// template-level statement of the algorithm next to the integer oracle and the HIP kernels.
#include <ac_dsp/ac_poly_dec.h>

#include <iostream>
#include <vector>

static const int NTAPS = 6, DF = 4;
typedef ac_fixed<16, 2, true> IN_T;
typedef ac_fixed<16, 2, true> CF_T;
typedef ac_fixed<40, 12, true> ACC_T;
typedef ac_fixed<16, 3, true, AC_RND, AC_SAT> OUT_T;
typedef ac_fixed<18, 6, true, AC_RND_CONV, AC_SAT> ACC_LOSSY;
struct STR_CF { CF_T coeffs[NTAPS * DF]; };

template <class ACC, class OUT>
static std::vector<OUT> expected(const std::vector<IN_T> &x, const STR_CF &c) {
  std::vector<IN_T> taps(NTAPS * DF, IN_T(0));
  std::vector<OUT> y;
  size_t pos = 0;
  while (x.size() - pos >= (size_t)DF) {
    ACC acc = 0.0;
    for (int df = DF - 1; df >= 0; df--) {
      for (int i = NTAPS * DF - 1; i >= 0; i--) { taps[i] = (i == 0) ? x[pos] : taps[i - 1]; }
      pos++;
      ACC acc1 = 0.0;
      for (int tp = 0; tp < NTAPS; tp++) { acc1 = acc1 + taps[tp * DF] * c.coeffs[tp + NTAPS * df]; }
      acc = acc + acc1;
    }
    OUT o = acc;
    y.push_back(o);
  }
  return y;
}

template <class ACC, class OUT>
static int run_case(const char *name) {
  STR_CF c;
  for (int i = 0; i < NTAPS * DF; i++) { c.coeffs[i] = CF_T(0.9 * (((i * 29) % 23) - 11) / 16.0); }
  std::vector<IN_T> x;
  for (int i = 0; i < 403; i++) { x.push_back(IN_T(1.7 * ((((i * 7919) % 1009) / 1009.0) - 0.5))); }   // 403 = 100 groups + 3 left over
  ac_poly_dec<IN_T, CF_T, STR_CF, ACC, OUT, NTAPS, DF> dut;
  ac_channel<IN_T> in;
  ac_channel<OUT> out;
  ac_channel<STR_CF> cch;
  cch.write(c);
  for (size_t i = 0; i < 150; i++) { in.write(x[i]); }
  dut.run(in, out, cch);                       // 37 groups consumed, 2 samples stay queued
  int fails = 0;
  if (in.debug_size() != 2 || out.debug_size() != 37) { std::cout << name << ": burst-1 counts " << in.debug_size() << " " << out.debug_size() << std::endl; fails++; }
  for (size_t i = 150; i < x.size(); i++) { in.write(x[i]); }
  dut.run(in, out, cch);
  if (in.debug_size() != 3) { fails++; }
  std::vector<OUT> ref = expected<ACC, OUT>(x, c);
  if (out.debug_size() != ref.size()) { std::cout << name << ": " << out.debug_size() << " outputs, expected " << ref.size() << std::endl; fails++; }
  for (size_t i = 0; i < ref.size() && out.available(1); i++) {
    OUT got = out.read();
    if (!(got == ref[i])) { if (fails < 5) { std::cout << name << ": mismatch @" << i << " " << got << " vs " << ref[i] << std::endl; } fails++; }
  }
  std::cout << name << ": " << (fails ? "FAILED" : "ok") << std::endl;
  return fails;
}

int main() {
  int fails = run_case<ACC_T, OUT_T>("lossless accumulator") + run_case<ACC_LOSSY, OUT_T>("lossy RND_CONV/SAT accumulator");
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
