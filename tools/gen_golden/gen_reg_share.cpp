// gen_reg_share.cpp -- golden vectors of the reference's ac_fir_reg_share (five cores, blocked coefficient addressing,
// caller-owned shift register, delay line), produced by the reference's own header (see common.h).  usage: gen_reg_share <out dir>
#include <ac_dsp/ac_fir_reg_share.h>

#include "common.h"

using namespace gg;

template <int N, class IN, class OUT, class CF, class ACC, int MWW, int BS, int BO, FTYPE ft>
static void one(Json &j, const char *tag, int n, int cf_bits, uint64_t seed) {
  IN reg[N];
  for (int i = 0; i < N; i++) { reg[i] = 0; }
  ac_fir_reg_share<N, IN, OUT, CF, ACC, MWW, BS, BO, ft> dut(reg);
  CF coeffs[N];
  std::vector<long long> cd, xs, ys, dl;
  for (int i = 0; i < N; i++) { coeffs[i] = rnd_bits<CF>(seed, cf_bits); cd.push_back(raw(coeffs[i])); }
  for (int t = 0; t < n; t++) {
    IN x = rnd<IN>(seed);
    OUT y, d;
    dut.run(x, coeffs, y);
    dut.ac_firProgCoeffs_delay_line(d);
    xs.push_back(raw(x)); ys.push_back(raw(y)); dl.push_back(raw(d));
  }
  char nm[160];
  snprintf(nm, sizeof nm, "reg_share_%s_%s_%d_w%d_b%d_o%d", tag, kFtypeNames[ft], N, MWW, BS, BO);
  j.begin(nm);
  j.str("class", "reg_share"); j.str("ftype", kFtypeNames[ft]); j.num("n_taps", N);
  j.num("mem_word_width", MWW); j.num("blk_sz", BS); j.num("blk_offset", BO);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("coeff", fmt_json<CF>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("coeffs", cd); j.arr("x", xs); j.arr("y", ys); j.arr("delay_line", dl);
  j.end();
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  Json j(dir + "/fir_reg_share.json");
  typedef ac_fixed<16, 2, true> I16;
  typedef ac_fixed<40, 12, true> A40;
  typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> O16;
  typedef ac_fixed<20, 6, true, AC_TRN, AC_SAT> ASAT;
  typedef ac_fixed<12, 3, true> I12;
  typedef ac_fixed<10, 2, true> C10;
  typedef ac_fixed<9, 5, true, AC_RND, AC_SAT> O9;
  one<64, I16, O16, I16, A40, 4, 4, 0, SHIFT_REG>(j, "base", 200, 13, 1);
  one<64, I16, O16, I16, A40, 4, 4, 0, FOLD_EVEN>(j, "base", 200, 13, 2);
  one<64, I16, O16, I16, A40, 4, 4, 0, FOLD_EVEN_ANTI>(j, "base", 200, 13, 3);
  one<63, I16, O16, I16, A40, 1, 1, 0, FOLD_ODD>(j, "base", 200, 13, 4);
  one<63, I16, O16, I16, A40, 1, 1, 0, FOLD_ODD_ANTI>(j, "base", 200, 13, 5);
  one<12, I16, O16, I16, A40, 2, 4, 0, SHIFT_REG>(j, "overlap", 60, 14, 6);        // MEM_WORD_WIDTH < BLK_SZ: overlapping words
  one<16, I16, O16, I16, A40, 4, 2, 1, FOLD_EVEN>(j, "offset", 60, 14, 7);         // BLK_OFFSET 1
  one<15, I12, O9, C10, ASAT, 2, 2, 0, FOLD_ODD>(j, "satacc", 80, 10, 8);          // saturating lossy ACC: order matters
  one<15, I12, O9, C10, ASAT, 1, 1, 0, FOLD_ODD_ANTI>(j, "satacc", 80, 10, 9);
  one<16, I12, O9, C10, ASAT, 2, 2, 0, FOLD_EVEN_ANTI>(j, "satacc", 80, 10, 10);
  one<12, I12, O9, C10, ASAT, 4, 4, 0, SHIFT_REG>(j, "satacc", 80, 10, 11);
  return 0;
}
