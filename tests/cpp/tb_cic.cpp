// tb_cic.cpp -- C++ testbench for the drop-in CIC class templates (own code, in the style of the
// reference's tests/rtest_ac_cic_{dec,intr}_full.cpp).  Exact comparison against the reference's MATLAB
// fixed-point vectors (tests/golden/ref_txt, opened by bare name), R=7 M=2, dec N=4 / intr N=5.
#include <ac_dsp/ac_cic_dec_full.h>
#include <ac_dsp/ac_cic_intr_full.h>   // both headers in one TU: possible here, not with the reference

#include <fstream>
#include <iostream>
#include <vector>

typedef ac_fixed<32, 16, true> IN_T;

static std::vector<long double> read_ld(const char *fn) {
  std::ifstream f(fn);
  std::vector<long double> v;
  long double d;
  while (f >> d) { v.push_back(d); }
  return v;
}

template <class OUT> static int compare(ac_channel<OUT> &out, const std::vector<long double> &ref, size_t ref_off, size_t want) {
  int errs = 0;
  size_t k = 0;
  while (out.available(1) && ref_off + k < ref.size()) {
    long double got = out.read().to_double();
    if (got != ref[ref_off + k]) {
      if (errs < 5) { std::cout << "  mismatch @" << k << " expected " << (double)ref[ref_off + k] << " got " << (double)got << std::endl; }
      errs++;
    }
    k++;
  }
  if (k < want) { std::cout << "  too few outputs: " << k << " < " << want << std::endl; errs++; }
  return errs;
}

int main() {
  int fails = 0;
  {
    typedef ac_fixed<48, 32, true> OUT_T;
    std::vector<long double> xin = read_ld("ac_cic_dec_full_input.txt"), ref = read_ld("ac_cic_dec_full_ref.txt");
    if (xin.size() != 10003 || ref.size() != 1429) { std::cerr << "missing dec vectors\n"; return 2; }
    ac_channel<IN_T> in;
    ac_channel<OUT_T> out;
    in.write(IN_T(0.0));  // the reference testbench prepends one zero (rtest_ac_cic_dec_full.cpp:84-85)
    for (size_t i = 0; i < xin.size(); i++) { in.write(IN_T((double)xin[i])); }
    ac_cic_dec_full<IN_T, OUT_T, 7, 2, 4> filter;
    filter.run(in, out);
    std::cout << "dec  outputs = " << out.debug_size() << std::endl;
    fails += compare(out, ref, 0, 1429);
  }
  {
    typedef ac_fixed<49, 33, true> OUT_T;
    std::vector<long double> xin = read_ld("ac_cic_intr_full_input.txt"), ref = read_ld("ac_cic_intr_full_ref.txt");
    if (xin.size() < 1000 || ref.size() < 6995) { std::cerr << "missing intr vectors\n"; return 2; }
    ac_channel<IN_T> in;
    ac_channel<OUT_T> out;
    ac_cic_intr_full<IN_T, OUT_T, 7, 2, 5> filter;
    // two bursts: the R-1 trailing results of the first burst appear with the second one
    for (int i = 0; i < 400; i++) { in.write(IN_T((double)xin[i])); }
    filter.run(in, out);
    if (out.debug_size() != 399 * 7 + 1 - 4) { std::cout << "  burst-1 count " << out.debug_size() << std::endl; fails++; }
    for (int i = 400; i < 1000; i++) { in.write(IN_T((double)xin[i])); }
    filter.run(in, out);
    std::cout << "intr outputs = " << out.debug_size() << std::endl;
    fails += compare(out, ref, 5, 6990);  // first N_TB refs are discarded by the reference test (:99)
  }
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
