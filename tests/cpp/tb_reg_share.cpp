// tb_reg_share.cpp -- C++ testbench for the ac_fir_reg_share drop-in (own code; the reference ships no test for this
// class).  Each design is driven sample by sample like the reference's usage example
// (include/ac_dsp/ac_fir_reg_share.h:44-60) and compared, bit for bit, with the same MAC loop written directly on the
// ac_fixed types of include/ac_types (ascending taps, blocked coefficient addressing, ACC_TYPE `fold`).
#include <ac_dsp/ac_fir_reg_share.h>

#include <cstdio>
#include <cstdlib>

typedef ac_fixed<16, 2, true> IN_T;
typedef ac_fixed<16, 2, true> CF_T;
typedef ac_fixed<40, 12, true> ACC_T;
typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT_T;
typedef ac_fixed<20, 6, true, AC_TRN, AC_SAT> ACC_LOSSY;   // saturating, too few fraction bits: strict MAC order matters

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int N, class ACC, int MWW, int BS, int BO, FTYPE ft> static int run_design(const char *name, unsigned seed) {
  IN_T reg[N], reg_ref[N];
  for (int i = 0; i < N; i++) { reg[i] = 0; reg_ref[i] = 0; }
  ac_fir_reg_share<N, IN_T, OUT_T, CF_T, ACC, MWW, BS, BO, ft> dut(reg);
  CF_T coeffs[N];
  for (int i = 0; i < N; i++) { coeffs[i].set_slc(0, ac_int<16, true>((int)(lcg(seed) % 8000) - 4000)); }
  const int count = ft == SHIFT_REG ? N : (ft == FOLD_EVEN || ft == FOLD_EVEN_ANTI) ? N / 2 : (N - 1) / 2 + 1;
  int bad = 0;
  for (int n = 0; n < 40; n++) {
    IN_T x;
    x.set_slc(0, ac_int<16, true>((int)(lcg(seed) % 65536) - 32768));
    OUT_T y, yr;
    dut.run(x, coeffs, y);
    // the same loops on the template types
    for (int i = N - 1; i >= 0; i--) { reg_ref[i] = (i == 0) ? x : reg_ref[i - 1]; }
    ACC acc = 0;
    int ram = 0;
    for (int i = 0; i < count; i += BS, ram += MWW) {
      int index = 0;
      for (int bc = BO; bc < BO + BS; bc++, index++) {
        const int t = i + index;
        if (ft == SHIFT_REG) { acc += reg_ref[t] * coeffs[ram + bc]; }
        else if (ft == FOLD_EVEN) { acc += (reg_ref[t] + reg_ref[N - 1 - t]) * coeffs[ram + bc]; }
        else if (ft == FOLD_EVEN_ANTI) { acc += (reg_ref[t] - reg_ref[N - 1 - t]) * coeffs[ram + bc]; }
        else {
          ACC fold;
          if (t == (N - 1) / 2) { fold = ACC(reg_ref[t]); }
          else if (ft == FOLD_ODD) { fold = ACC(reg_ref[t] + reg_ref[N - 1 - t]); }
          else { fold = ACC(reg_ref[t] - reg_ref[N - 1 - t]); }
          acc += coeffs[ram + bc] * fold;
        }
      }
    }
    yr = acc;
    if (!(y == yr)) { bad++; }
    for (int i = 0; i < N; i++) { if (!(reg[i] == reg_ref[i])) { bad++; } }
    OUT_T dl, dlr;
    dut.ac_firProgCoeffs_delay_line(dl);
    dlr = reg_ref[N - 1];
    if (!(dl == dlr)) { bad++; }
  }
  printf("%-44s %s\n", name, bad ? "FAILED" : "ok");
  return bad;
}

int main() {
  int bad = 0;
  bad += run_design<12, ACC_T, 4, 4, 0, SHIFT_REG>("SHIFT_REG 12 taps, blocks of 4", 1);
  bad += run_design<12, ACC_T, 2, 4, 0, SHIFT_REG>("SHIFT_REG 12 taps, MWW 2 BLK 4 (overlapping words)", 2);
  bad += run_design<16, ACC_T, 4, 2, 1, FOLD_EVEN>("FOLD_EVEN 16 taps, MWW 4 BLK 2 OFFSET 1", 3);
  bad += run_design<16, ACC_T, 4, 4, 0, FOLD_EVEN_ANTI>("FOLD_EVEN_ANTI 16 taps", 4);
  bad += run_design<15, ACC_T, 2, 2, 0, FOLD_ODD>("FOLD_ODD 15 taps", 5);
  bad += run_design<15, ACC_T, 1, 1, 0, FOLD_ODD_ANTI>("FOLD_ODD_ANTI 15 taps", 6);
  bad += run_design<15, ACC_LOSSY, 1, 1, 0, FOLD_ODD_ANTI>("FOLD_ODD_ANTI 15 taps, saturating lossy ACC", 7);
  bad += run_design<12, ACC_LOSSY, 4, 4, 0, SHIFT_REG>("SHIFT_REG 12 taps, saturating lossy ACC", 8);
  bad += run_design<16, ACC_LOSSY, 2, 2, 0, FOLD_EVEN_ANTI>("FOLD_EVEN_ANTI 16 taps, saturating lossy ACC", 9);
  printf("%s\n", bad ? "Test FAILED." : "Test PASSED.");
  return bad ? 1 : 0;
}
