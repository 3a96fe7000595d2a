// tb_mv_avg.cpp -- C++ testbench for the ac_mv_avg drop-in (own code; the reference ships no test for this class).
// Driven like the reference's usage example (include/ac_dsp/ac_mv_avg.h:45-63): n_sample through its channel, whole frames
// through data_in, one run() call.  Reference values: the same frame loop and MAC loop written directly on the ac_fixed
// templates and the ac_window_1d_flag of include/ac_types (ac_mv_avg.h:111-123,164-189).
#include <ac_dsp/ac_mv_avg.h>

#include <cstdio>
#include <vector>

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MAXS, int TAPS, ac_window_mode WM, class IN, class OUT, class ACC, class CF>
static int run_design(const char *name, int n, int frames, unsigned seed) {
  typedef ac_int<12, false> ST;
  CF c[TAPS];
  for (int i = 0; i < TAPS; i++) { c[i].set_slc(0, ac_int<CF::width, CF::sign>((int)(lcg(seed) % 2000) - (CF::sign ? 1000 : 0))); }
  ac_mv_avg<MAXS, TAPS, WM, IN, OUT, ACC, CF, ST> dut(c);
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<ST> ns;
  ns.write(ST(n));
  std::vector<IN> x((size_t)(n * frames));
  const int span = 1 << IN::width;
  for (size_t i = 0; i < x.size(); i++) { x[i].set_slc(0, ac_int<IN::width, IN::sign>((int)(lcg(seed) % span) - (IN::sign ? span / 2 : 0))); in.write(x[i]); }
  dut.run(in, out, ns);
  // the reference's loops on the template types
  std::vector<OUT> want;
  ac_window_1d_flag<IN, TAPS, WM> w;
  size_t pos = 0;
  for (int f = 0; f < frames; f++) {
    const int sample = (WM == AC_WIN) ? n : n + TAPS / 2;
    IN d = 0;
    for (int cnt = 0; cnt < sample; cnt++) {
      if (cnt < n) { d = x[pos++]; }
      w.write(d, cnt == 0, cnt == n - 1);
      if (w.valid()) {
        ACC acc = 0;
        for (int j = -TAPS / 2; j <= TAPS / 2; j++) { acc = acc + (ACC)w[j] * c[j + TAPS / 2]; }
        OUT o = acc;
        want.push_back(o);
      }
    }
  }
  int bad = 0;
  for (size_t k = 0; k < want.size(); k++) {
    if (!out.available(1)) { bad++; break; }
    OUT got = out.read();
    if (!(got == want[k])) { bad++; }
  }
  if (out.available(1)) { bad++; }
  printf("%-56s %s (%zu outputs)\n", name, bad ? "FAILED" : "ok", want.size());
  return bad;
}

typedef ac_fixed<16, 8, true> I16;
typedef ac_fixed<16, 1, true> C16;
typedef ac_fixed<32, 14, true> A32;
typedef ac_fixed<20, 10, true, AC_RND, AC_SAT> O20;
typedef ac_fixed<16, 6, true, AC_RND_CONV, AC_SAT> ASAT;

int main() {
  int bad = 0;
  bad += run_design<1024, 9, AC_WIN, I16, O20, A32, C16>("AC_WIN 9 taps, 300-sample frames", 300, 2, 1);
  bad += run_design<1024, 9, AC_MIRROR, I16, O20, A32, C16>("AC_MIRROR 9 taps, 300-sample frames", 300, 2, 2);
  bad += run_design<1024, 9, AC_CLIP, I16, O20, A32, C16>("AC_CLIP 9 taps, 300-sample frames", 300, 2, 3);
  bad += run_design<64, 9, AC_MIRROR, I16, O20, A32, C16>("AC_MIRROR 9 taps, 5-sample frames", 5, 6, 4);
  bad += run_design<64, 9, AC_CLIP, I16, O20, A32, C16>("AC_CLIP 9 taps, 1-sample frames", 1, 6, 5);
  bad += run_design<256, 7, AC_MIRROR, ac_fixed<24, 12, true>, ac_fixed<14, 5, true, AC_RND, AC_SAT>, ASAT, ac_fixed<12, 2, true> >(
      "AC_MIRROR 7 taps, lossy ACC cast, saturating ACC", 120, 2, 6);
  bad += run_design<256, 7, AC_WIN, ac_fixed<24, 12, true>, ac_fixed<14, 5, true, AC_RND, AC_SAT>, ASAT, ac_fixed<12, 2, true> >(
      "AC_WIN 7 taps, lossy ACC cast, saturating ACC", 120, 2, 7);
  printf("%s\n", bad ? "Test FAILED." : "Test PASSED.");
  return bad ? 1 : 0;
}
