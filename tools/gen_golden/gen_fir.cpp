// gen_fir.cpp -- golden vectors of the reference's ac_fir_const_coeffs / ac_fir_load_coeffs / ac_fir_prog_coeffs
// (all six FTYPEs), produced by the reference's own headers (see common.h).  usage: gen_fir <out dir>
#include <ac_dsp/ac_fir_const_coeffs.h>
#include <ac_dsp/ac_fir_load_coeffs.h>
#include <ac_dsp/ac_fir_prog_coeffs.h>

#include "common.h"

using namespace gg;

// BASELINE types (configs[1]): <16,2> in / coefficients, ACC <40,12>, OUT <16,2,RND,SAT>
struct TBase {
  typedef ac_fixed<16, 2, true> IN; typedef ac_fixed<16, 2, true> CF; typedef ac_fixed<40, 12, true> ACC;
  typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT;
  static const char *name() { return "base"; }
  static const int in_bits = 16, cf_bits = 16;
};
// the types of the reference's own prog testbench (tests/rtest_ac_fir_prog_coeffs.cpp:50-53): lossy ACC (38 > 32 fraction bits)
struct TProg {
  typedef ac_fixed<28, 6, true, AC_TRN, AC_WRAP> IN; typedef ac_fixed<23, 7, true, AC_TRN, AC_WRAP> CF;
  typedef ac_fixed<64, 32, true, AC_TRN, AC_WRAP> ACC; typedef ac_fixed<64, 32, true, AC_TRN, AC_WRAP> OUT;
  static const char *name() { return "progtest"; }
  static const int in_bits = 28, cf_bits = 23;
};
// saturating, lossy accumulator: strict MAC order matters
struct TSat {
  typedef ac_fixed<12, 3, true> IN; typedef ac_fixed<10, 2, true> CF; typedef ac_fixed<20, 6, true, AC_TRN, AC_SAT> ACC;
  typedef ac_fixed<9, 5, true, AC_RND, AC_SAT> OUT;
  static const char *name() { return "satacc"; }
  static const int in_bits = 12, cf_bits = 10;
};
// sign-dependent rounding in the accumulator and symmetric saturation; OUT rounds to infinity and zeroes on overflow
struct TQO {
  typedef ac_fixed<16, 2, true> IN; typedef ac_fixed<16, 2, true> CF; typedef ac_fixed<26, 8, true, AC_RND_ZERO, AC_SAT_SYM> ACC;
  typedef ac_fixed<14, 3, true, AC_RND_INF, AC_SAT_ZERO> OUT;
  static const char *name() { return "qo"; }
  static const int in_bits = 16, cf_bits = 16;
};
// unsigned wrapping accumulator under signed inputs (a negative sum wraps to 2^W - |v| before OUT saturates), unsigned IN
struct TUAcc {
  typedef ac_fixed<16, 2, true> IN; typedef ac_fixed<16, 2, true> CF; typedef ac_fixed<40, 12, false> ACC;
  typedef ac_fixed<16, 2, true, AC_RND, AC_SAT> OUT;
  static const char *name() { return "uacc"; }
  static const int in_bits = 16, cf_bits = 13;
};
struct TUIn {
  typedef ac_fixed<12, 4, false> IN; typedef ac_fixed<14, 1, true> CF; typedef ac_fixed<36, 12, true> ACC;
  typedef ac_fixed<18, 6, true, AC_RND_CONV, AC_SAT> OUT;
  static const char *name() { return "uin"; }
  static const int in_bits = 12, cf_bits = 14;
};

template <class T> static void header(Json &j, const char *cls, int ft, int n_taps) {
  j.str("class", cls);
  j.str("ftype", kFtypeNames[ft]);
  j.num("n_taps", n_taps);
  j.rawjson("in", fmt_json<typename T::IN>());
  j.rawjson("coeff", fmt_json<typename T::CF>());
  j.rawjson("acc", fmt_json<typename T::ACC>());
  j.rawjson("out", fmt_json<typename T::OUT>());
}

template <class T, int N> static void make_coeffs(uint64_t &seed, typename T::CF (&c)[N], std::vector<long long> &dump, bool symmetric) {
  // low-pass-like magnitudes: centre taps use the full coefficient width, the tails a few bits (keeps sum |c| moderate)
  for (int i = 0; i < N; i++) {
    const int d = i < N - 1 - i ? i : N - 1 - i;
    int bits = T::cf_bits - (N / 2 - d) / 8;
    if (bits < 4) { bits = 4; }
    c[i] = rnd_bits<typename T::CF>(seed, bits);
  }
  if (symmetric) { for (int i = 0; i < N / 2; i++) { c[N - 1 - i] = c[i]; } }
  for (int i = 0; i < N; i++) { dump.push_back(raw(c[i])); }
}

// ac_fir_const_coeffs: coefficient pointer bound at construction, run() drains the input channel
template <class T, unsigned N, FTYPE ft> static void case_const(Json &j, int n, int split_mode, uint64_t seed) {
  typename T::CF c[N];
  std::vector<long long> cd, xs, ys, calls;
  make_coeffs<T, (int)N>(seed, c, cd, (seed & 1) != 0);
  ac_fir_const_coeffs<typename T::IN, typename T::OUT, typename T::CF, typename T::ACC, N, ft> dut(c);
  ac_channel<typename T::IN> in;
  ac_channel<typename T::OUT> out;
  for (int k : splits(n, split_mode)) {
    for (int i = 0; i < k; i++) { typename T::IN x = rnd<typename T::IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, out);
    calls.push_back(k);
    while (out.available(1)) { ys.push_back(raw(out.read())); }
  }
  char nm[128];
  snprintf(nm, sizeof nm, "const_%s_%s_%u%s", T::name(), kFtypeNames[ft], N, split_mode ? "_chunked" : "");
  j.begin(nm);
  header<T>(j, "const", ft, N);
  j.arr("coeffs", cd); j.num("reload_at", -1);
  j.arr("calls", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

// ac_fir_load_coeffs: coefficients arrive through the channel pair (coeffs_ch, ld); a second set is loaded mid-stream
template <class T, unsigned N, FTYPE ft> static void case_load(Json &j, int n, int split_mode, uint64_t seed) {
  typename T::CF c0[N], c1[N];
  std::vector<long long> cd, cd1, xs, ys, calls;
  make_coeffs<T, (int)N>(seed, c0, cd, false);
  make_coeffs<T, (int)N>(seed, c1, cd1, true);
  ac_fir_load_coeffs<typename T::IN, typename T::OUT, typename T::CF, typename T::ACC, N, ft> dut;
  ac_channel<typename T::IN> in;
  ac_channel<typename T::OUT> out;
  ac_channel<typename T::CF> cch;
  ac_channel<bool> ld;
  const std::vector<int> sp = splits(n, split_mode ? 1 : 2);
  // mode 2 (one-shot flavour): two calls, the reload between them
  std::vector<int> plan;
  if (split_mode) { plan = sp; } else { plan.push_back(n / 2); plan.push_back(n - n / 2); }
  int done = 0, reload_at = -1;
  for (size_t ci = 0; ci < plan.size(); ci++) {
    const int k = plan[ci];
    if (ci == 0) { for (unsigned i = 0; i < N; i++) { cch.write(c0[i]); } ld.write(true); }
    else if (reload_at < 0 && done >= n / 2) { for (unsigned i = 0; i < N; i++) { cch.write(c1[i]); } ld.write(true); reload_at = done; }
    for (int i = 0; i < k; i++) { typename T::IN x = rnd<typename T::IN>(seed); xs.push_back(raw(x)); in.write(x); }
    dut.run(in, cch, out, ld);
    calls.push_back(k);
    done += k;
    while (out.available(1)) { ys.push_back(raw(out.read())); }
  }
  char nm[128];
  snprintf(nm, sizeof nm, "load_%s_%s_%u%s", T::name(), kFtypeNames[ft], N, split_mode ? "_chunked" : "");
  j.begin(nm);
  header<T>(j, "load", ft, N);
  j.arr("coeffs", cd); j.arr("coeffs2", cd1); j.num("reload_at", reload_at);
  j.arr("calls", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

// ac_fir_prog_coeffs: one sample per run() call, coefficient array passed with every call; it changes mid-stream
template <class T, int N, FTYPE ft> static void case_prog(Json &j, int n, uint64_t seed) {
  typename T::CF c0[N], c1[N];
  std::vector<long long> cd, cd1, xs, ys, calls;
  make_coeffs<T, N>(seed, c0, cd, true);
  make_coeffs<T, N>(seed, c1, cd1, false);
  ac_fir_prog_coeffs<typename T::IN, typename T::OUT, typename T::CF, typename T::ACC, N, ft> dut;
  ac_channel<typename T::IN> in;
  ac_channel<typename T::OUT> out;
  const int reload_at = (2 * n) / 3;
  for (int i = 0; i < n; i++) {
    typename T::IN x = rnd<typename T::IN>(seed);
    xs.push_back(raw(x));
    in.write(x);
    dut.run(in, out, i < reload_at ? c0 : c1);
  }
  calls.push_back(n);
  while (out.available(1)) { ys.push_back(raw(out.read())); }
  char nm[128];
  snprintf(nm, sizeof nm, "prog_%s_%s_%d", T::name(), kFtypeNames[ft], N);
  j.begin(nm);
  header<T>(j, "prog", ft, N);
  j.arr("coeffs", cd); j.arr("coeffs2", cd1); j.num("reload_at", reload_at);
  j.arr("calls", calls); j.arr("x", xs); j.arr("y", ys);
  j.end();
}

template <class T, unsigned N> static void all_const(Json &j, int n, uint64_t seed, bool chunked_too) {
  case_const<T, N, SHIFT_REG>(j, n, 0, seed + 1);
  case_const<T, N, ROTATE_SHIFT>(j, n, 0, seed + 2);
  case_const<T, N, C_BUFF>(j, n, 0, seed + 3);
  case_const<T, N, FOLD_EVEN>(j, n, 0, seed + 4);
  case_const<T, N, FOLD_ODD>(j, n, 0, seed + 5);
  case_const<T, N, TRANSPOSED>(j, n, 0, seed + 6);
  if (chunked_too) {
    case_const<T, N, SHIFT_REG>(j, n, 1, seed + 7);
    case_const<T, N, FOLD_ODD>(j, n, 1, seed + 8);
    case_const<T, N, C_BUFF>(j, n, 1, seed + 9);
  }
}
template <class T, unsigned N> static void all_load(Json &j, int n, uint64_t seed, bool chunked_too) {
  case_load<T, N, SHIFT_REG>(j, n, 0, seed + 1);
  case_load<T, N, ROTATE_SHIFT>(j, n, 0, seed + 2);
  case_load<T, N, C_BUFF>(j, n, 0, seed + 3);
  case_load<T, N, FOLD_EVEN>(j, n, 0, seed + 4);
  case_load<T, N, FOLD_ODD>(j, n, 0, seed + 5);
  case_load<T, N, TRANSPOSED>(j, n, 0, seed + 6);
  if (chunked_too) {
    case_load<T, N, TRANSPOSED>(j, n, 1, seed + 7);
    case_load<T, N, SHIFT_REG>(j, n, 1, seed + 8);
  }
}
template <class T, int N> static void all_prog(Json &j, int n, uint64_t seed) {
  case_prog<T, N, SHIFT_REG>(j, n, seed + 1);
  case_prog<T, N, ROTATE_SHIFT>(j, n, seed + 2);
  case_prog<T, N, C_BUFF>(j, n, seed + 3);
  case_prog<T, N, FOLD_EVEN>(j, n, seed + 4);
  case_prog<T, N, FOLD_ODD>(j, n, seed + 5);
  case_prog<T, N, TRANSPOSED>(j, n, seed + 6);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  {
    Json j(dir + "/fir_const_255.json");
    all_const<TBase, 255>(j, 560, 100, true);
  }
  {
    Json j(dir + "/fir_load_255.json");
    all_load<TBase, 255>(j, 560, 200, true);
  }
  {
    Json j(dir + "/fir_prog_255.json");
    all_prog<TBase, 255>(j, 420, 300);
  }
  {
    Json j(dir + "/fir_types_const.json");   // short filters over the awkward type sets, odd and even tap counts
    all_const<TProg, 27>(j, 130, 400, false);
    all_const<TSat, 27>(j, 130, 500, true);
    all_const<TSat, 28>(j, 160, 600, false);
    all_const<TQO, 31>(j, 130, 700, true);
    all_const<TQO, 30>(j, 160, 800, false);
    all_const<TUAcc, 33>(j, 160, 900, false);
    all_const<TUIn, 21>(j, 160, 1000, false);
  }
  {
    Json j(dir + "/fir_types_load_prog.json");
    all_load<TProg, 27>(j, 160, 1100, false);
    all_load<TSat, 28>(j, 160, 1200, true);
    all_load<TQO, 31>(j, 160, 1300, false);
    all_prog<TProg, 27>(j, 120, 1400);
    all_prog<TSat, 27>(j, 120, 1500);
    all_prog<TUAcc, 33>(j, 120, 1600);
  }
  return 0;
}
