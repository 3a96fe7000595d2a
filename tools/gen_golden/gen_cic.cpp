// gen_cic.cpp -- golden vectors of the reference's ac_cic_dec_full / ac_cic_intr_full, produced by the reference's own
// headers (see common.h).  The two headers both define a global `power` template (ac_cic_dec_full.h:94,
// ac_cic_intr_full.h:90), so the file is compiled twice: -DGEN_INTR selects the interpolator.  usage: gen_cic_{dec,intr} <out dir>
#ifdef GEN_INTR
#include <ac_dsp/ac_cic_intr_full.h>
#define CIC_CLASS ac_cic_intr_full
#define CIC_NAME "cic_intr"
#else
#include <ac_dsp/ac_cic_dec_full.h>
#define CIC_CLASS ac_cic_dec_full
#define CIC_NAME "cic_dec"
#endif

#include "common.h"

using namespace gg;

template <class IN, class OUT, unsigned R, unsigned M, unsigned N> static void one(Json &j, const char *tag, int n, int split_mode, uint64_t seed) {
  CIC_CLASS<IN, OUT, R, M, N> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  std::vector<long long> xs, ys, calls, outs_per_call;
  for (int k : splits(n, split_mode)) {
    for (int i = 0; i < k; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, out);
    calls.push_back(k);
    long long c = 0;
    while (out.available(1)) { ys.push_back(raw(out.read())); c++; }
    outs_per_call.push_back(c);
  }
  char nm[160];
  snprintf(nm, sizeof nm, "%s_%s_R%u_M%u_N%u%s", CIC_NAME, tag, R, M, N, split_mode ? "_chunked" : "");
  j.begin(nm);
  j.str("class", CIC_NAME);
  j.num("R", R); j.num("M", M); j.num("N", N);
  j.rawjson("in", fmt_json<IN>());
  j.rawjson("out", fmt_json<OUT>());
  j.arr("calls", calls); j.arr("outs_per_call", outs_per_call); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  Json j(dir + "/" CIC_NAME ".json");
  typedef ac_fixed<32, 16, true> I32;
  typedef ac_fixed<16, 1, true> I16;
  typedef ac_fixed<12, 4, false> U12;
#ifndef GEN_INTR
  // BASELINE configs[2]: N5 R8 M1 on <32,16>, OUT = the lossless INT_TYPE <47,31>; configs[4] stage A: R16 on <16,1> -> <36,21>
  one<I32, ac_fixed<47, 31, true>, 8, 1, 5>(j, "cfg3", 1100, 0, 11);
  one<I32, ac_fixed<47, 31, true>, 8, 1, 5>(j, "cfg3", 1100, 1, 12);
  one<I16, ac_fixed<36, 21, true>, 16, 1, 5>(j, "cfg5", 1700, 0, 13);
  one<I16, ac_fixed<36, 21, true>, 16, 1, 5>(j, "cfg5", 1700, 1, 14);
  one<I32, ac_fixed<48, 32, true>, 7, 2, 4>(j, "reftest", 900, 1, 15);                      // the reference testbench's parameters
  one<I32, ac_fixed<24, 20, true, AC_RND, AC_SAT>, 8, 1, 5>(j, "narrow_out", 900, 0, 16);  // OUT_TYPE conversion (ac_cic_dec_full.h:219)
  one<U12, ac_fixed<30, 18, true, AC_RND_CONV, AC_SAT_SYM>, 5, 3, 3>(j, "unsigned_in_M3", 700, 1, 17);   // M = 3: the delay-line quirk
  one<I16, ac_fixed<40, 25, true>, 3, 4, 6>(j, "M4", 600, 0, 18);
#else
  one<I32, ac_fixed<44, 28, true>, 8, 1, 5>(j, "cfg", 200, 0, 21);      // INT_TYPE of N5 R8 M1 on <32,16>: 12 + 32 bits
  one<I32, ac_fixed<44, 28, true>, 8, 1, 5>(j, "cfg", 200, 1, 22);
  one<I32, ac_fixed<49, 33, true>, 7, 2, 5>(j, "reftest", 200, 1, 23);  // the reference testbench's parameters
  one<I16, ac_fixed<20, 6, true, AC_RND, AC_SAT>, 4, 1, 3>(j, "narrow_out", 260, 0, 24);
  one<U12, ac_fixed<34, 26, true>, 5, 3, 4>(j, "unsigned_in_M3", 200, 1, 25);
  one<I16, ac_fixed<30, 15, true>, 16, 1, 2>(j, "R16", 150, 0, 26);
#endif
  return 0;
}
