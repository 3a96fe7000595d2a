// gen_wide.cpp -- golden vectors at widths ABOVE 64 bits, produced by the reference's own headers (see common.h): ac_cic_dec_full
// whose derived INT_TYPE is 78 bits (<48,20> through R 16, M 2, N 6: reference ac_cic_dec_full.h:116-137) and ac_fir_load_coeffs
// with an 80-bit ACC_TYPE / 72-bit saturating OUT_TYPE (ac_fir_load_coeffs.h:180-278).  Words of more than 64 bits are written as
// plain decimal integers (Python's json reads them exactly).  usage: gen_wide <out dir>
// The two CIC headers both define a global `power` template, so the interpolator half is a second build of this file (-DGEN_INTR).
#ifdef GEN_INTR
#include <ac_dsp/ac_cic_intr_full.h>
#define CIC_CLASS ac_cic_intr_full
#define CIC_NAME "wide_cic_intr"
#else
#include <ac_dsp/ac_cic_dec_full.h>
#include <ac_dsp/ac_fir_load_coeffs.h>
#define CIC_CLASS ac_cic_dec_full
#define CIC_NAME "wide_cic_dec"
#endif

#include "common.h"

using namespace gg;

static std::string dec128(__int128 v) {
  if (v == 0) { return "0"; }
  const bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-(v + 1)) + 1 : (unsigned __int128)v;
  std::string s;
  while (u) { s.insert(s.begin(), (char)('0' + (int)(u % 10))); u /= 10; }
  return neg ? "-" + s : s;
}
static std::string arr128(const std::vector<__int128> &v) {
  std::string s = "[";
  for (size_t i = 0; i < v.size(); i++) { s += (i ? "," : "") + dec128(v[i]); }
  return s + "]";
}

template <class IN, class OUT, unsigned R, unsigned M, unsigned N> static void cic(Json &j, const char *tag, int n, int split_mode, uint64_t seed) {
  CIC_CLASS<IN, OUT, R, M, N> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  std::vector<long long> xs, calls, outs_per_call;
  std::vector<__int128> ys;
  for (int k : splits(n, split_mode)) {
    for (int i = 0; i < k; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, out);
    calls.push_back(k);
    long long c = 0;
    while (out.available(1)) { ys.push_back(out.read().raw128()); c++; }
    outs_per_call.push_back(c);
  }
  char nm[160];
  snprintf(nm, sizeof nm, "%s_%s_R%u_M%u_N%u%s", CIC_NAME, tag, R, M, N, split_mode ? "_chunked" : "");
  j.begin(nm);
  j.str("class", CIC_NAME);
  j.num("R", R); j.num("M", M); j.num("N", N);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("calls", calls); j.arr("outs_per_call", outs_per_call); j.arr("x", xs); j.rawjson("y", arr128(ys));
  j.end();
}

#ifndef GEN_INTR
template <class IN, class OUT, class CF, class ACC, unsigned NT, FTYPE ft> static void fir(Json &j, const char *tag, int n, int split_mode, uint64_t seed) {
  ac_fir_load_coeffs<IN, OUT, CF, ACC, NT, ft> dut;
  ac_channel<IN> in;
  ac_channel<CF> cch;
  ac_channel<OUT> out;
  ac_channel<bool> ld;
  std::vector<long long> xs, cs, calls;
  std::vector<__int128> ys;
  for (unsigned i = 0; i < NT; i++) { CF c = rnd<CF>(seed); cs.push_back(raw(c)); cch.write(c); }
  ld.write(true);
  dut.run(in, cch, out, ld);
  for (int k : splits(n, split_mode)) {
    for (int i = 0; i < k; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, cch, out, ld);
    calls.push_back(k);
    while (out.available(1)) { ys.push_back(out.read().raw128()); }
  }
  char nm[160];
  snprintf(nm, sizeof nm, "wide_fir_%s_%s%s", tag, kFtypeNames[(int)ft], split_mode ? "_chunked" : "");
  j.begin(nm);
  j.str("class", "wide_fir");
  j.str("ftype", kFtypeNames[(int)ft]);
  j.num("n_taps", NT);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("coeff", fmt_json<CF>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("calls", calls); j.arr("coeffs", cs); j.arr("x", xs); j.rawjson("y", arr128(ys));
  j.end();
}

#endif

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  typedef ac_fixed<48, 20, true> I48;
#ifdef GEN_INTR
  Json j(dir + "/wide_intr.json");
  cic<I48, ac_fixed<74, 46, true>, 16, 2, 6>(j, "int74", 60, 0, 51);                            // OUT = the derived INT_TYPE <74,46>
  cic<I48, ac_fixed<74, 46, true>, 16, 2, 6>(j, "int74", 60, 1, 52);
  cic<I48, ac_fixed<66, 40, true, AC_RND, AC_SAT>, 16, 2, 6>(j, "int74_sat66", 50, 1, 53);
  cic<ac_fixed<64, 30, true>, ac_fixed<100, 50, true, AC_RND_CONV, AC_SAT_SYM>, 8, 3, 5>(j, "int84_out100", 40, 0, 54);
  return 0;
#else
  Json j(dir + "/wide.json");
  cic<I48, ac_fixed<78, 50, true>, 16, 2, 6>(j, "int78", 700, 0, 31);                          // OUT = the derived INT_TYPE <78,50>
  cic<I48, ac_fixed<78, 50, true>, 16, 2, 6>(j, "int78", 700, 1, 32);
  cic<I48, ac_fixed<70, 40, true, AC_RND, AC_SAT>, 16, 2, 6>(j, "int78_sat70", 500, 1, 33);    // 78-bit INT_TYPE -> saturating 70-bit OUT
  cic<ac_fixed<45, 20, true>, ac_fixed<40, 20, true, AC_RND_CONV, AC_SAT_SYM>, 8, 1, 8>(j, "int69_narrow_out", 400, 0, 34);
  typedef ac_fixed<32, 16, true> S32;
  typedef ac_fixed<80, 40, true> A80;
  typedef ac_fixed<72, 40, true, AC_RND, AC_SAT> O72;
  fir<S32, O72, S32, A80, 27, SHIFT_REG>(j, "acc80", 300, 0, 41);
  fir<S32, O72, S32, A80, 27, C_BUFF>(j, "acc80", 300, 1, 42);
  fir<S32, O72, S32, A80, 27, FOLD_EVEN>(j, "acc80", 300, 0, 43);
  fir<S32, O72, S32, A80, 27, FOLD_ODD>(j, "acc80", 300, 1, 44);
  fir<S32, O72, S32, A80, 27, TRANSPOSED>(j, "acc80", 300, 1, 45);
  fir<S32, ac_fixed<80, 40, true>, S32, ac_fixed<80, 40, true, AC_TRN_ZERO, AC_SAT>, 16, ROTATE_SHIFT>(j, "acc80_sat", 200, 0, 46);   // saturating 80-bit accumulator
  return 0;
#endif
}
