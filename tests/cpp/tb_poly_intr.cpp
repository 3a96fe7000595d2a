// tb_poly_intr.cpp -- C++ testbench for the ac_poly_intr drop-in (own code; the reference ships no test for this class).
// Driven like the reference's usage example (include/ac_dsp/ac_poly_intr.h:38-66): control + coefficient structs through
// their channels with read_ctrl = true, then one run() per input sample.  Reference values: the same three cores written
// directly on the ac_fixed templates of include/ac_types (shift register, IN_TYPE negation, ACC_TYPE fold, accumulator
// banks with the one-sample delay, symmetric-pair correction >> 1).
#include <ac_dsp/ac_poly_intr.h>

#include <cstdio>
#include <vector>

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <class IN, class CF, class ACC, class OUT, int N, int CSZ, int IFAC, FTYPE ft> static int run_design(const char *name, unsigned seed) {
  struct ctrl_s { bool sign[IFAC]; ac_int<8, false> corr[IFAC]; };
  struct coef_s { CF coeffs[CSZ]; };
  ac_poly_intr<IN, CF, ACC, OUT, ctrl_s, coef_s, N, CSZ, IFAC, ft> dut;
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<ctrl_s> cch;
  ac_channel<coef_s> kch;
  ac_channel<bool> flag;
  ctrl_s ct;
  coef_s co;
  for (int j = 0; j < IFAC; j++) { ct.sign[j] = (lcg(seed) & 1) != 0; ct.corr[j] = IFAC - 1 - j; }
  for (int i = 0; i < CSZ; i++) { co.coeffs[i].set_slc(0, ac_int<CF::width, true>((int)(lcg(seed) % 2000) - 1000)); }
  cch.write(ct); kch.write(co); flag.write(true);
  dut.run(in, out, cch, kch, flag);
  // reference model state
  IN taps[N];
  ACC acc_a[IFAC], acc_b[IFAC];
  for (int i = 0; i < N; i++) { taps[i] = 0; }
  for (int j = 0; j < IFAC; j++) { acc_a[j] = 0; acc_b[j] = 0; }
  bool flip = false, init = false;
  int bad = 0;
  const int span = 1 << IN::width;
  for (int n = 0; n < 60; n++) {
    IN x;
    x.set_slc(0, ac_int<IN::width, true>((int)(lcg(seed) % span) - span / 2));
    in.write(x); flag.write(false);
    dut.run(in, out, cch, kch, flag);
    std::vector<OUT> want;
    for (int i = N - 1; i >= 1; i--) { taps[i] = taps[i - 1]; }
    for (int j = 0; j < IFAC; j++) {
      if (j == 0) { taps[0] = x; flip = !flip; }
      ACC fold, acc;
      acc = 0;
      if (ft == FOLD_ANTI) {
        for (int i = N - 1; i >= 0; i--) { acc += taps[i] * co.coeffs[i + N * j]; }
        OUT o = acc;
        want.push_back(o);
        continue;
      }
      if (ft == FOLD_EVEN) {
        for (int i = (N / 2) - 1; i >= 0; i--) {
          IN tp;
          if (ct.sign[j]) { tp = taps[N - 1 - i]; } else { tp = -taps[N - 1 - i]; }
          fold = (taps[i] + tp);
          acc += co.coeffs[i + j * N / 2] * fold;
        }
      } else {
        for (int i = 0; i < (((N - 1) / 2) + 1); i++) {
          if (i == (N - 1) / 2) { fold = taps[i]; }
          else {
            IN tp;
            if (ct.sign[j]) { tp = taps[N - 1 - i]; } else { tp = -taps[N - 1 - i]; }
            fold = (taps[i] + tp);
          }
          acc += co.coeffs[i + (N / 2 + 1) * j] * fold;
        }
      }
      ACC t1, t2;
      const int cj = (int)ct.corr[j];
      if (flip) { acc_b[j] = acc; t1 = acc_a[j]; t2 = acc_a[cj]; } else { acc_a[j] = acc; t1 = acc_b[j]; t2 = acc_b[cj]; }
      if (init) {
        if (j != cj) {
          ACC tn;
          if (ct.sign[j]) { tn = -t2; } else { tn = t2; }
          OUT o = (t1 + tn) >> 1;
          want.push_back(o);
        } else {
          OUT o = t1;
          want.push_back(o);
        }
      }
    }
    init = true;
    for (size_t k = 0; k < want.size(); k++) {
      if (!out.available(1)) { bad++; break; }
      OUT got = out.read();
      if (!(got == want[k])) { bad++; }
    }
    if (out.available(1)) { bad++; while (out.available(1)) { out.read(); } }
  }
  printf("%-52s %s\n", name, bad ? "FAILED" : "ok");
  return bad;
}

typedef ac_fixed<16, 2, true> I16;
typedef ac_fixed<40, 12, true> A40;
typedef ac_fixed<12, 3, true, AC_RND, AC_SAT> I12S;
typedef ac_fixed<10, 2, true> C10;
typedef ac_fixed<18, 7, true, AC_RND_CONV, AC_SAT_SYM> A18;
typedef ac_fixed<9, 5, true, AC_RND, AC_SAT> O9;

int main() {
  int bad = 0;
  bad += run_design<I16, I16, A40, A40, 8, 16, 4, FOLD_EVEN>("FOLD_EVEN 8 taps x IF 4, lossless types", 1);
  bad += run_design<I16, I16, A40, A40, 9, 20, 4, FOLD_ODD>("FOLD_ODD 9 taps x IF 4, lossless types", 2);
  bad += run_design<I16, I16, A40, A40, 5, 15, 3, FOLD_ANTI>("FOLD_ANTI 5 taps x IF 3, lossless types", 3);
  bad += run_design<I12S, C10, A18, O9, 8, 16, 4, FOLD_EVEN>("FOLD_EVEN 8 taps x IF 4, saturating narrow types", 4);
  bad += run_design<I12S, C10, A18, O9, 7, 16, 4, FOLD_ODD>("FOLD_ODD 7 taps x IF 4, saturating narrow types", 5);
  bad += run_design<I12S, C10, A18, O9, 6, 12, 2, FOLD_ANTI>("FOLD_ANTI 6 taps x IF 2, saturating narrow types", 6);
  printf("%s\n", bad ? "Test FAILED." : "Test PASSED.");
  return bad ? 1 : 0;
}
