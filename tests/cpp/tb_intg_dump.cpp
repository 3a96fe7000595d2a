// tb_intg_dump.cpp -- C++ testbench for the ac_intg_dump drop-in (own code; the reference ships no test for this class),
// driven like the reference's usage example (include/ac_dsp/ac_intg_dump.h:40-62) and compared bit for bit with the same
// block / round / channel loop written on the ac_fixed templates of include/ac_types.
#include <ac_dsp/ac_intg_dump.h>

#include <cstdio>
#include <vector>

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <class IN, class ACC, class OUT, int NS, int CHN> static int run_design(const char *name, unsigned seed) {
  typedef ac_int<16, false> N_T;
  ac_intg_dump<IN, ACC, OUT, N_T, NS, CHN> dut;
  ACC temp[CHN];
  for (int i = 0; i < CHN; i++) { temp[i] = 0; }
  int bad = 0;
  const int wbits = IN::width < 24 ? IN::width : 24;   // stimulus magnitude (full scale for the narrow types)
  const int span = 1 << wbits;
  for (int call = 0; call < 3; call++) {
    ac_channel<IN> in;
    ac_channel<OUT> out;
    ac_channel<N_T> ns;
    std::vector<OUT> want;
    const int n_blocks = 5 + call;
    for (int b = 0; b < n_blocks; b++) {
      int n = 1 + (int)(lcg(seed) % NS);
      if (b == 2) { n = NS + 3; }        // too large: NS rounds, no dump, sums carry into the next block
      if (b == 4 && call == 1) { n = 0; }
      ns.write(N_T(n));
      bool flag = false;
      for (int j = 1; j <= NS; j++) {
        for (int i = 0; i < CHN; i++) {
          IN x;
          x.set_slc(0, ac_int<IN::width, true>(IN::sign ? (int)(lcg(seed) % span) - span / 2 : (int)(lcg(seed) % span)));
          in.write(x);
          temp[i] = temp[i] + x;
          if (j == n) { OUT o = temp[i]; want.push_back(o); temp[i] = 0; flag = true; }
        }
        if (flag) { break; }
      }
    }
    dut.run(in, out, ns);
    for (size_t k = 0; k < want.size(); k++) {
      if (!out.available(1)) { bad++; break; }
      OUT got = out.read();
      if (!(got == want[k])) { bad++; }
    }
    if (out.available(1) || in.available(1)) { bad++; }
  }
  printf("%-56s %s\n", name, bad ? "FAILED" : "ok");
  return bad;
}

int main() {
  int bad = 0;
  bad += run_design<ac_fixed<32, 16, true>, ac_fixed<64, 32, true>, ac_fixed<64, 32, true>, 24, 4>("usage-example types, NS 24, CHN 4", 1);
  bad += run_design<ac_fixed<16, 2, true>, ac_fixed<20, 6, true, AC_TRN, AC_SAT>, ac_fixed<12, 6, true, AC_RND, AC_SAT>, 16, 3>(
      "saturating ACC, rounding OUT, NS 16, CHN 3", 2);
  bad += run_design<ac_fixed<12, 12, false>, ac_fixed<14, 14, false, AC_TRN, AC_WRAP>, ac_fixed<14, 14, false>, 9, 1>("unsigned wrapping types, CHN 1", 3);
  printf("%s\n", bad ? "Test FAILED." : "Test PASSED.");
  return bad ? 1 : 0;
}
