// gen_poly_dec.cpp -- golden vectors of the reference's ac_poly_dec and ac_intg_dump, produced by the reference's own
// headers (see common.h).  usage: gen_poly_dec <out dir>
#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_dsp/ac_poly_dec.h>
#include <ac_dsp/ac_intg_dump.h>

#include "common.h"

using namespace gg;

template <class IN, class CF, class ACC, class OUT, int NT, int DF> static void polydec(Json &j, const char *tag, int n_groups, int split_mode, int cf_bits, uint64_t seed) {
  struct coef_s { CF coeffs[NT * DF]; };
  ac_poly_dec<IN, CF, coef_s, ACC, OUT, NT, DF> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<coef_s> cch;
  coef_s co;
  std::vector<long long> cd, xs, ys, calls;
  for (int i = 0; i < NT * DF; i++) { co.coeffs[i] = rnd_bits<CF>(seed, cf_bits); cd.push_back(raw(co.coeffs[i])); }
  cch.write(co);
  for (int k : splits(n_groups, split_mode)) {
    for (int i = 0; i < k * DF; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, out, cch);
    calls.push_back((long long)k * DF);
    while (out.available(1)) { ys.push_back(raw(out.read())); }
  }
  char nm[160];
  snprintf(nm, sizeof nm, "poly_dec_%s_T%d_DF%d%s", tag, NT, DF, split_mode ? "_chunked" : "");
  j.begin(nm);
  j.str("class", "poly_dec"); j.num("n_taps", NT); j.num("df", DF);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("coeff", fmt_json<CF>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("coeffs", cd); j.arr("calls", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

template <class IN, class ACC, class OUT, int NS, int CHN> static void intgdump(Json &j, const char *tag, const std::vector<int> &n_sample, int blocks_per_call, uint64_t seed) {
  typedef ac_int<12, false> NT;
  ac_intg_dump<IN, ACC, OUT, NT, NS, CHN> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<NT> ns;
  std::vector<long long> xs, ys, nsv, calls;
  for (size_t b0 = 0; b0 < n_sample.size(); b0 += blocks_per_call) {
    long long nb = 0;
    for (size_t b = b0; b < n_sample.size() && b < b0 + blocks_per_call; b++, nb++) {
      const int v = n_sample[b];
      ns.write(NT(v));
      nsv.push_back(v);
      const int rounds = (v >= 1 && v <= NS) ? v : NS;
      for (int i = 0; i < rounds * CHN; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
    }
    dut.run(in, out, ns);
    calls.push_back(nb);
    while (out.available(1)) { ys.push_back(raw(out.read())); }
  }
  char nm[160];
  snprintf(nm, sizeof nm, "intg_dump_%s_NS%d_CHN%d", tag, NS, CHN);
  j.begin(nm);
  j.str("class", "intg_dump"); j.num("ns", NS); j.num("chn", CHN);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("n_sample", nsv); j.arr("blocks_per_call", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  typedef ac_fixed<16, 2, true> I16;
  typedef ac_fixed<40, 12, true> A40;
  typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> O16;
  {
    Json j(dir + "/poly_dec.json");
    polydec<I16, I16, A40, O16, 16, 8>(j, "base", 120, 0, 13, 1);
    polydec<I16, I16, A40, O16, 16, 8>(j, "base", 120, 1, 13, 2);
    polydec<I16, I16, A40, A40, 5, 3>(j, "wide_out", 100, 1, 14, 3);
    polydec<ac_fixed<12, 3, true>, ac_fixed<10, 2, true>, ac_fixed<20, 6, true, AC_TRN, AC_SAT>, ac_fixed<9, 5, true, AC_RND, AC_SAT>, 4, 4>(j, "satacc", 100, 0, 10, 4);
    polydec<ac_fixed<24, 8, true>, I16, ac_fixed<48, 20, true>, ac_fixed<24, 8, true, AC_RND_CONV, AC_SAT_SYM>, 8, 2>(j, "wide_in", 100, 1, 14, 5);
  }
  {
    Json j(dir + "/intg_dump.json");
    std::vector<int> a;
    for (int b = 0; b < 40; b++) { a.push_back(1 + (b * 7) % 16); }
    intgdump<ac_fixed<16, 8, true>, ac_fixed<32, 16, true>, ac_fixed<32, 16, true>, 16, 4>(j, "dumps", a, 5, 11);
    std::vector<int> b;
    const int pat[] = {8, 0, 3, 900, 8, 8, 0, 0, 5, 1, 4000, 2};   // 0 and > NS: NS rounds, no dump, the sums carry on
    for (int r = 0; r < 3; r++) { for (int v : pat) { b.push_back(v); } }
    intgdump<ac_fixed<16, 8, true>, ac_fixed<32, 16, true>, ac_fixed<20, 12, true, AC_RND, AC_SAT>, 8, 3>(j, "carry", b, 4, 12);
    intgdump<ac_fixed<12, 3, true>, ac_fixed<14, 5, true, AC_TRN, AC_SAT>, ac_fixed<9, 5, true, AC_RND, AC_SAT>, 8, 2>(j, "satacc", b, 7, 13);
    intgdump<ac_fixed<12, 4, false>, ac_fixed<18, 8, false>, ac_fixed<18, 8, false>, 8, 1>(j, "unsigned", b, 36, 14);
  }
  return 0;
}
