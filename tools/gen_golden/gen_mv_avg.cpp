// gen_mv_avg.cpp -- golden vectors of the reference's ac_mv_avg: the reference's own run() / mvAvgCore() source
// (frame loop, ACC_TYPE cast, MAC order) over this repo's ac_window_1d_flag subset (include/ac_types/ac_window.h -- the
// window class itself is not part of the reference tree, so its boundary rules are NOT pinned by these vectors).
// usage: gen_mv_avg <out dir>
#include <ac_dsp/ac_mv_avg.h>

#include "common.h"

using namespace gg;

static const char *mode_name(ac_window_mode m) { return m == AC_WIN ? "WIN" : (m == AC_MIRROR ? "MIRROR" : "CLIP"); }

template <int MAXS, int TAPS, ac_window_mode WM, class IN, class OUT, class ACC, class CF>
static void one(Json &j, const char *tag, int n_sample, int n_frames, int cf_bits, uint64_t seed) {
  typedef ac_int<12, false> ST;
  CF c[TAPS];
  std::vector<long long> cd, xs, ys;
  for (int i = 0; i < TAPS; i++) { c[i] = rnd_bits<CF>(seed, cf_bits); cd.push_back(raw(c[i])); }
  ac_mv_avg<MAXS, TAPS, WM, IN, OUT, ACC, CF, ST> dut(c);
  ac_channel<IN> in;
  ac_channel<OUT> out;
  ac_channel<ST> ns;
  ns.write(ST(n_sample));
  for (int i = 0; i < n_sample * n_frames; i++) { IN x = rnd<IN>(seed); xs.push_back(raw(x)); in.write(x); }
  dut.run(in, out, ns);
  while (out.available(1)) { ys.push_back(raw(out.read())); }
  char nm[160];
  snprintf(nm, sizeof nm, "mv_avg_%s_%s_T%d_n%d_f%d", tag, mode_name(WM), TAPS, n_sample, n_frames);
  j.begin(nm);
  j.str("class", "mv_avg"); j.str("win_mode", mode_name(WM)); j.num("taps", TAPS); j.num("max_sample", MAXS);
  j.num("n_sample", n_sample); j.num("n_frames", n_frames);
  j.rawjson("in", fmt_json<IN>()); j.rawjson("coeff", fmt_json<CF>()); j.rawjson("acc", fmt_json<ACC>()); j.rawjson("out", fmt_json<OUT>());
  j.arr("coeffs", cd); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

template <ac_window_mode WM> static void all(Json &j, uint64_t seed) {
  typedef ac_fixed<16, 8, true> I16;
  typedef ac_fixed<16, 1, true> C16;
  typedef ac_fixed<32, 14, true> A32;
  typedef ac_fixed<20, 10, true, AC_RND, AC_SAT> O20;
  one<1024, 9, WM, I16, O20, A32, C16>(j, "base", 300, 2, 14, seed + 1);
  one<1024, 31, WM, I16, A32, A32, C16>(j, "base", 200, 1, 12, seed + 2);
  one<64, 9, WM, I16, O20, A32, C16>(j, "short_frames", 5, 7, 14, seed + 3);     // frames shorter than the window
  one<64, 5, WM, I16, O20, A32, C16>(j, "single", 1, 4, 14, seed + 4);
  // the header's own usage example shapes: ACC narrower than IN (the cast loses bits), saturating accumulator
  one<256, 7, WM, ac_fixed<24, 12, true>, ac_fixed<16, 6, true, AC_RND, AC_SAT>, ac_fixed<16, 6, true, AC_RND_CONV, AC_SAT>, ac_fixed<12, 2, true> >(
      j, "lossy_cast", 120, 2, 11, seed + 5);
  one<256, 5, WM, ac_fixed<12, 4, false>, ac_fixed<24, 10, false>, ac_fixed<24, 10, false>, ac_fixed<10, 0, false> >(j, "unsigned", 90, 2, 10, seed + 6);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  Json j(dir + "/mv_avg.json");
  all<AC_WIN>(j, 10);
  all<AC_MIRROR>(j, 20);
  all<AC_CLIP>(j, 30);
  return 0;
}
