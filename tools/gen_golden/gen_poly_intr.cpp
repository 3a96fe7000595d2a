// gen_poly_intr.cpp -- golden vectors of the reference's ac_poly_intr (three cores), produced by the reference's own
// header (see common.h; that header has its own FTYPE enum, so it cannot share a translation unit with the FIR headers).
// usage: gen_poly_intr <out dir>
#include <ac_dsp/ac_poly_intr.h>

#include "common.h"

using namespace gg;

static const char *const kPolyNames[] = {"FOLD_EVEN", "FOLD_ODD", "FOLD_ANTI"};

template <class IN, class CF, class ACC, class OUT, int N, int CSZ, int IFAC, FTYPE ft>
static void one(Json &j, const char *tag, int n, int cf_bits, bool reload, uint64_t seed) {
  struct ctrl_s { bool sign[IFAC]; ac_int<8, false> corr[IFAC]; };
  struct coef_s { CF coeffs[CSZ]; };
  ac_poly_intr<IN, CF, ACC, OUT, ctrl_s, coef_s, N, CSZ, IFAC, ft> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<ctrl_s> cch;
  ac_channel<coef_s> kch;
  ac_channel<bool> flag;
  std::vector<long long> xs, ys, outs_per_sample;
  std::vector<std::vector<long long> > cds, sgs, crs;
  auto load_ctrl = [&]() {
    ctrl_s ct;
    coef_s co;
    std::vector<long long> cd, sg, cr;
    for (int q = 0; q < IFAC; q++) {
      ct.sign[q] = (splitmix64(seed) & 1) != 0;
      // symmetric-pair partner: mostly the mirrored phase, sometimes the phase itself
      const int partner = (splitmix64(seed) % 3 == 0) ? q : IFAC - 1 - q;
      ct.corr[q] = partner;
      sg.push_back(ct.sign[q]); cr.push_back(partner);
    }
    for (int i = 0; i < CSZ; i++) { co.coeffs[i] = rnd_bits<CF>(seed, cf_bits); cd.push_back(raw(co.coeffs[i])); }
    cch.write(ct); kch.write(co); flag.write(true);
    dut.run(in, out, cch, kch, flag);
    cds.push_back(cd); sgs.push_back(sg); crs.push_back(cr);
  };
  load_ctrl();
  int reload_at = -1;
  for (int t = 0; t < n; t++) {
    if (reload && t == n / 2) { load_ctrl(); reload_at = t; }
    IN x = rnd<IN>(seed);
    xs.push_back(raw(x));
    in.write(x); flag.write(false);
    dut.run(in, out, cch, kch, flag);
    long long c = 0;
    while (out.available(1)) { ys.push_back(raw(out.read())); c++; }
    outs_per_sample.push_back(c);
  }
  char nm[160];
  snprintf(nm, sizeof nm, "poly_intr_%s_%s_T%d_C%d_IF%d", tag, kPolyNames[ft], N, CSZ, IFAC);
  j.begin(nm);
  j.str("class", "poly_intr"); j.str("ftype", kPolyNames[ft]); j.num("n_taps", N); j.num("coeff_sz", CSZ); j.num("ifac", IFAC);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("coeff", fmt_json<CF>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("coeffs", cds[0]); j.arr("sign", sgs[0]); j.arr("corr", crs[0]);
  j.num("reload_at", reload_at);
  if (reload_at >= 0) { j.arr("coeffs2", cds[1]); j.arr("sign2", sgs[1]); j.arr("corr2", crs[1]); }
  j.arr("x", xs); j.arr("outs_per_sample", outs_per_sample); j.arr("y", ys);
  j.end();
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  Json j(dir + "/poly_intr.json");
  typedef ac_fixed<16, 2, true> I16;
  typedef ac_fixed<40, 12, true> A40;
  typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> O16;
  typedef ac_fixed<12, 3, true, AC_RND, AC_SAT> I12S;
  typedef ac_fixed<10, 2, true> C10;
  typedef ac_fixed<18, 7, true, AC_RND_CONV, AC_SAT_SYM> A18;
  typedef ac_fixed<9, 5, true, AC_RND, AC_SAT> O9;
  one<I16, I16, A40, O16, 16, 64, 8, FOLD_EVEN>(j, "base", 120, 13, false, 1);
  one<I16, I16, A40, O16, 15, 64, 8, FOLD_ODD>(j, "base", 120, 13, true, 2);
  one<I16, I16, A40, O16, 8, 64, 8, FOLD_ANTI>(j, "base", 120, 13, false, 3);
  one<I16, I16, A40, A40, 8, 16, 4, FOLD_EVEN>(j, "wide_out", 100, 14, true, 4);
  one<I12S, C10, A18, O9, 8, 16, 4, FOLD_EVEN>(j, "sat", 100, 10, false, 5);
  one<I12S, C10, A18, O9, 7, 16, 4, FOLD_ODD>(j, "sat", 100, 10, true, 6);
  one<I12S, C10, A18, O9, 6, 12, 2, FOLD_ANTI>(j, "sat", 100, 10, false, 7);
  one<I16, I16, A40, O16, 4, 2, 1, FOLD_EVEN>(j, "if1", 60, 14, false, 8);
  return 0;
}
