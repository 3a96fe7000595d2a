// tb_tiny.cpp -- the host-side path of tiny calls (include/ac_dsp/acdsp_engine.h: fir_engine::run_values_c) against the GPU path.
//
// ac_fir_prog_coeffs::run is ONE sample per call (reference include/ac_dsp/ac_fir_prog_coeffs.h:281): bursts below the break-even of a
// kernel launch run on the host, in the caller's ac_fixed arithmetic, and the filter state moves to the device (and back) as a state blob
// whenever the other side runs next.  Every configuration below streams the same samples through
//   (a) a filter that sees a schedule of bursts on both sides of the threshold (1, 1, 3, 500, 1, 2, 400, 7 ... samples), and
//   (b) an engine that only ever runs the GPU kernels (whose results the Python suite checks against the CPU oracle),
// with coefficient changes at the same burst boundaries, and compares the outputs bit for bit: all six FTYPEs, the three classes,
// exact / lossy / saturating accumulators, sign-dependent rounding, an 80-bit accumulator, unsigned samples.
// The last block times one-sample calls (the drop-in ac_fir_prog_coeffs) on both paths.
#include <ac_dsp/ac_fir_const_coeffs.h>
#include <ac_dsp/ac_fir_load_coeffs.h>
#include <ac_dsp/ac_fir_prog_coeffs.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}
template <class T> static T rnd_fixed() {
  T v;
  v.set_slc(0, ac_int<T::width, T::sign>((long long)rnd64()));
  return v;
}

static int fails = 0;

template <class IN, class OUT, class CF, class ACC>
static void check_one(const char *name, int kind, int ftype, int n_taps) {
  acdsp::fir_engine<IN, OUT, CF, ACC> mixed(kind, ftype, n_taps), gpu(kind, ftype, n_taps);
  std::vector<CF> c0((size_t)n_taps), c1((size_t)n_taps);
  for (int i = 0; i < n_taps; i++) { c0[(size_t)i] = rnd_fixed<CF>(); c1[(size_t)i] = rnd_fixed<CF>(); }
  const int bursts[] = {1, 1, 3, 500, 1, 2, 400, 7, 1, 350, 1, 1};
  size_t bad = 0, total = 0;
  for (size_t b = 0; b < sizeof bursts / sizeof bursts[0]; b++) {
    // load / prog classes: another set from the fifth burst on (TRANSPOSED then carries partial sums of both sets)
    const CF *c = (kind != ACDSP_FIR_CONST && b >= 4) ? c1.data() : c0.data();
    // calls of more than ~300 samples x 27 taps pass the default threshold (8192 MACs); with 255 taps everything above 32 samples does
    std::vector<IN> x((size_t)bursts[b]);
    for (size_t i = 0; i < x.size(); i++) { x[i] = rnd_fixed<IN>(); }
    if (b == 5) { for (size_t i = 0; i < x.size(); i++) { x[i].template set_val<AC_VAL_MIN>(); } }
    std::vector<OUT> ym, yg;
    mixed.run_values_c(x, ym, c);
    gpu.set_coeffs(c);
    gpu.run_values(x, yg);
    for (size_t i = 0; i < x.size(); i++) { total++; if (!(ym[i] == yg[i])) { bad++; } }
  }
  if (bad) { fails++; }
  printf("%-58s ftype %d kind %d taps %3d: %zu / %zu mismatches%s\n", name, ftype, kind, n_taps, bad, total, bad ? "  <-- FAIL" : "");
}

template <class IN, class OUT, class CF, class ACC>
static void sweep(const char *name) {
  static const int ftypes[] = {ACDSP_SHIFT_REG, ACDSP_ROTATE_SHIFT, ACDSP_C_BUFF, ACDSP_FOLD_EVEN, ACDSP_FOLD_ODD, ACDSP_TRANSPOSED};
  for (int f = 0; f < 6; f++) {
    check_one<IN, OUT, CF, ACC>(name, ACDSP_FIR_PROG, ftypes[f], 27);
    check_one<IN, OUT, CF, ACC>(name, ACDSP_FIR_CONST, ftypes[f], 28);
  }
  check_one<IN, OUT, CF, ACC>(name, ACDSP_FIR_LOAD, ACDSP_TRANSPOSED, 9);
  check_one<IN, OUT, CF, ACC>(name, ACDSP_FIR_LOAD, ACDSP_FOLD_ODD, 64);
}

int main() {
  typedef ac_fixed<64, 32, true, AC_TRN, AC_WRAP> A64;
  sweep<ac_fixed<28, 6, true>, A64, ac_fixed<23, 7, true>, A64>("prog testbench types <28,6> x <23,7> -> <64,32> (lossy)");
  sweep<ac_fixed<16, 2, true>, ac_fixed<16, 2, true, AC_RND, AC_SAT>, ac_fixed<16, 2, true>, ac_fixed<40, 12, true> >("<16,2> exact sums -> <16,2,RND,SAT>");
  sweep<ac_fixed<16, 2, true>, ac_fixed<12, 3, true, AC_RND_CONV, AC_SAT_SYM>, ac_fixed<14, 2, true>, ac_fixed<24, 6, true, AC_TRN_ZERO, AC_SAT> >(
      "saturating accumulator, sign-dependent rounding");
  sweep<ac_fixed<15, 3, false>, ac_fixed<32, 10, true, AC_RND, AC_SAT>, ac_fixed<16, 2, true>, ac_fixed<44, 16, true> >("unsigned samples");
  sweep<ac_fixed<32, 16, true>, ac_fixed<80, 40, true>, ac_fixed<32, 16, true>, ac_fixed<80, 40, true> >("80-bit accumulator");
  check_one<ac_fixed<16, 2, true>, ac_fixed<16, 2, true, AC_RND, AC_SAT>, ac_fixed<16, 2, true>, ac_fixed<40, 12, true> >("255 taps", ACDSP_FIR_PROG, ACDSP_SHIFT_REG, 255);
  check_one<ac_fixed<16, 2, true>, ac_fixed<16, 2, true, AC_RND, AC_SAT>, ac_fixed<16, 2, true>, ac_fixed<40, 12, true> >("255 taps", ACDSP_FIR_LOAD, ACDSP_TRANSPOSED, 255);

  // one-sample calls of the drop-in class: the reference's calling pattern (tests/rtest_ac_fir_prog_coeffs.cpp:109-113)
  {
    typedef ac_fixed<28, 6, true> IN;
    typedef ac_fixed<23, 7, true> CF;
    ac_fir_prog_coeffs<IN, A64, CF, A64, 27, FOLD_ODD> filt;
    CF c[27];
    for (int i = 0; i < 27; i++) { c[i] = rnd_fixed<CF>(); }
    ac_channel<IN> in;
    ac_channel<A64> out;
    const int calls = 20000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; i++) { in.write(rnd_fixed<IN>()); filt.run(in, out, c); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
    printf("ac_fir_prog_coeffs<<28,6>, <23,7>, 27 taps, FOLD_ODD>::run, one sample per call: %.3f us per call (threshold %lld MACs)\n", us,
           (long long)acdsp::fir_engine<IN, A64, CF, A64>::small_macs());
    while (out.available(1)) { out.read(); }
  }
  if (fails) { printf("Test FAILED (%d configurations).\n", fails); return 1; }
  printf("Test PASSED.\n");
  return 0;
}
