// tb_fir.cpp -- C++ testbench for the drop-in FIR class templates (own code, written in the style of the
// reference's tests/rtest_ac_fir_{const,load,prog}_coeffs.cpp so the parity checks read the same way).
// Stimulus: the reference two-tone (rtest_ac_fir_const_coeffs.cpp:126-151); pass criterion: SQNR >= 60 dB
// against the reference's MATLAB vectors (data files under tests/golden/ref_txt, opened by bare name).
#include <ac_dsp/ac_fir_const_coeffs.h>
#include <ac_dsp/ac_fir_load_coeffs.h>
#include <ac_dsp/ac_fir_prog_coeffs.h>

#include <cmath>
#include <fstream>
#include <iostream>
#include <vector>

static const double PI_V = 3.14159265358979323846;

template <class T> std::vector<T> two_tone(int n) {
  T probe;
  double tmax = probe.template set_val<AC_VAL_MAX>().to_double();
  std::vector<double> v(n);
  double amax = 0;
  for (int i = 0; i < n; i++) {
    v[i] = sin(2 * PI_V * 25 * i / 500.0) + sin(2 * PI_V * 150 * i / 500.0);
    if (fabs(v[i]) > amax) { amax = fabs(v[i]); }
  }
  std::vector<T> out(n);
  for (int i = 0; i < n; i++) { out[i] = (v[i] / amax) * tmax; }
  return out;
}

static std::vector<double> read_doubles(const char *fn) {
  std::ifstream f(fn);
  std::vector<double> v;
  std::string tok;
  while (f >> tok) {
    if (!tok.empty() && tok[tok.size() - 1] == ',') { tok.erase(tok.size() - 1); }
    v.push_back(atof(tok.c_str()));
  }
  return v;
}

template <class OUT> static double sqnr(ac_channel<OUT> &out, const std::vector<double> &ref, bool cast_ref) {
  double noise = 0, sig = 0;
  size_t k = 0;
  while (out.available(1) && k < ref.size()) {
    double d = out.read().to_double();
    double r = cast_ref ? ((OUT)ref[k]).to_double() : ref[k];
    noise += (d - r) * (d - r);
    sig += r * r;
    k++;
  }
  return 10 * log10(sig / noise);
}

typedef ac_fixed<64, 32, true, AC_TRN, AC_WRAP> ACC64;

// --- const coeffs: wrapper that owns the array, exactly the usage pattern of the reference test ---
typedef ac_fixed<16, 8, true, AC_TRN, AC_WRAP> C_IN;
typedef ac_fixed<32, 16, true, AC_TRN, AC_WRAP> C_CF;
template <unsigned N, FTYPE ft>
class const_wrapper : public ac_fir_const_coeffs<C_IN, ACC64, C_CF, ACC64, N, ft> {
public:
  C_CF coeffs[N];
  const_wrapper(const std::vector<double> &c) : ac_fir_const_coeffs<C_IN, ACC64, C_CF, ACC64, N, ft>(coeffs) {
    for (unsigned i = 0; i < N; i++) { coeffs[i] = c[i]; }  // initialised after the base: run() must read lazily
  }
};

int main() {
  int fails = 0;
  {
    std::vector<double> cfg = read_doubles("ac_fir_const_coeffs_cfg.txt"), ref = read_doubles("ac_fir_const_coeffs_ref.txt");
    if (cfg.size() != 29 || ref.size() < 1024) { std::cerr << "missing const vectors\n"; return 2; }
    ac_channel<C_IN> in;
    ac_channel<ACC64> out;
    std::vector<C_IN> x = two_tone<C_IN>(1024);
    for (size_t i = 0; i < x.size(); i++) { in.write(x[i]); }
    const_wrapper<29, FOLD_ODD> filter(cfg);
    filter.run(in, out);
    double s = sqnr(out, ref, true);
    std::cout << "const  SQNR = " << s << " dB" << std::endl;
    if (!(s >= 60.0) || fabs(s - 84.24) > 0.01) { fails++; }
  }
  {
    typedef ac_fixed<32, 16, true, AC_TRN, AC_WRAP> T;
    std::vector<double> cfg = read_doubles("ac_fir_load_coeffs_cfg.txt"), ref = read_doubles("ac_fir_load_coeffs_ref.txt");
    ac_channel<T> in, cch;
    ac_channel<ACC64> out;
    ac_channel<bool> ld;
    for (size_t i = 0; i < cfg.size(); i++) { cch.write(T(cfg[i])); }
    ac_fir_load_coeffs<T, ACC64, T, ACC64, 27, FOLD_ODD> filter;
    ld.write(true);
    filter.run(in, cch, out, ld);   // load phase: no data queued, no output
    if (out.available(1) || cch.available(1)) { fails++; }
    ld.write(false);
    std::vector<T> x = two_tone<T>(1024);
    // feed in two bursts: state must carry across run() calls
    for (int i = 0; i < 300; i++) { in.write(x[i]); }
    filter.run(in, cch, out, ld);
    for (int i = 300; i < 1024; i++) { in.write(x[i]); }
    filter.run(in, cch, out, ld);
    double s = sqnr(out, ref, true);
    std::cout << "load   SQNR = " << s << " dB" << std::endl;
    if (!(s >= 60.0) || fabs(s - 89.56) > 0.01) { fails++; }
  }
  {
    typedef ac_fixed<28, 6, true, AC_TRN, AC_WRAP> TI;
    typedef ac_fixed<23, 7, true, AC_TRN, AC_WRAP> TC;
    std::vector<double> cfg = read_doubles("ac_fir_prog_coeffs_cfg.txt"), ref = read_doubles("ac_fir_prog_coeffs_ref.txt");
    ac_channel<TI> in;
    ac_channel<ACC64> out;
    TC coeffs[27];
    for (int i = 0; i < 27; i++) { coeffs[i] = cfg[i]; }
    ac_fir_prog_coeffs<TI, ACC64, TC, ACC64, 27, FOLD_ODD> filter;
    std::vector<TI> x = two_tone<TI>(1024);
    for (int i = 0; i < 1024; i++) {   // one sample per call, as ac_fir_prog_coeffs.h:281 prescribes
      in.write(x[i]);
      filter.run(in, out, coeffs);
    }
    if (out.debug_size() != 1024) { fails++; }
    double s = sqnr(out, ref, false);
    std::cout << "prog   SQNR = " << s << " dB" << std::endl;
    if (!(s >= 60.0) || fabs(s - 89.56) > 0.01) { fails++; }
  }
  {
    // copying an object copies its state (the reference classes are plain aggregates)
    typedef ac_fixed<16, 2, true> T;
    typedef ac_fixed<40, 12, true> A;
    T c[5] = {0.25, -0.5, 1.0, -0.5, 0.25};
    ac_fir_prog_coeffs<T, A, T, A, 5> f1;
    ac_channel<T> in;
    ac_channel<A> o1, o2;
    for (int i = 0; i < 3; i++) { in.write(T(0.125 * (i + 1))); f1.run(in, o1, c); }
    ac_fir_prog_coeffs<T, A, T, A, 5> f2(f1);
    ac_channel<T> in2;
    in.write(T(1.0)); in2.write(T(1.0));
    f1.run(in, o1, c);
    f2.run(in2, o2, c);
    A last1; while (o1.available(1)) { last1 = o1.read(); }
    if (!(o2.read() == last1)) { std::cout << "copy semantics FAILED" << std::endl; fails++; }
  }
  {
    // TRANSPOSED with coefficients that change mid-stream (ac_fir_prog_coeffs hands a set to every one-sample call): reg_trans[] keeps the
    // partial sums made with the coefficients of their time (ac_fir_prog_coeffs.h:233-247).  The model below is that recurrence on the same
    // ac_fixed types; the engine serves the n_taps - 1 outputs behind a change from reg_trans and everything else from the input history.
    typedef ac_fixed<16, 2, true> T;
    typedef ac_fixed<40, 12, true> A;
    const int N = 9;
    T sets[3][N];
    for (int k = 0; k < 3; k++) { for (int i = 0; i < N; i++) { sets[k][i] = T(0.03125 * ((i * 7 + k * 5) % 23 - 11)); } }
    ac_fir_prog_coeffs<T, A, T, A, N, TRANSPOSED> f;
    A rt[N];
    for (int i = 0; i < N; i++) { rt[i] = 0; }
    ac_channel<T> in;
    ac_channel<A> out;
    int bad = 0;
    const int change_at[] = {0, 30, 33, 34, 60};            // changes closer together than n_taps - 1 samples, then a long steady run
    int which = 0, nxt = 0;
    for (int t = 0; t < 120; t++) {
      if (nxt < 5 && t == change_at[nxt]) { which = nxt % 3; nxt++; }
      const T x = T(0.001953125 * ((t * 37) % 1001 - 500));
      in.write(x);
      f.run(in, out, sets[which]);
      for (int i = N - 1; i >= 0; i--) { rt[i] = x * sets[which][N - 1 - i] + (i ? rt[i - 1] : A(0)); }
      if (!out.available(1) || !(out.read() == rt[N - 1])) { bad++; }
      if (t == 32) {                                         // a copy taken inside a transition carries reg_trans along
        ac_fir_prog_coeffs<T, A, T, A, N, TRANSPOSED> g(f);
        ac_channel<T> in2;
        ac_channel<A> out2;
        A rt2[N];
        for (int i = 0; i < N; i++) { rt2[i] = rt[i]; }
        const T x2 = T(0.25);
        in2.write(x2);
        g.run(in2, out2, sets[2]);
        for (int i = N - 1; i >= 0; i--) { rt2[i] = x2 * sets[2][N - 1 - i] + (i ? rt2[i - 1] : A(0)); }
        if (!(out2.read() == rt2[N - 1])) { bad++; }
      }
    }
    std::cout << "transposed, coefficient sets changing mid-stream: " << bad << " mismatches" << std::endl;
    if (bad) { fails++; }
  }
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
