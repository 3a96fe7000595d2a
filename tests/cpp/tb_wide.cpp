// tb_wide.cpp -- C++ testbench: the drop-in class templates with types WIDER than 64 bits (16-byte engine containers).
// The reference's templates take any width -- INT_TYPE of ac_cic_dec_full is derived (reference ac_cic_dec_full.h:116-137; here
// <48,20> through R 16, M 2, N 6 = 78 bits), ACC_TYPE / OUT_TYPE of the FIR classes are the user's (ac_fir_load_coeffs.h:246-259).
// Expected values: the reference's per-sample loops written out below on the ac_fixed templates (own code, the loop shapes of
// ac_cic_full_core.h:80-87,110-135,228-255 and ac_fir_load_coeffs.h:145-151,246-259), compared exactly.
#include <ac_dsp/ac_cic_dec_full.h>
#include <ac_dsp/ac_fir_load_coeffs.h>

#include <cstdint>
#include <iostream>
#include <vector>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
template <class T> static T rnd_fixed() {
  T v;
  v.set_slc(0, ac_int<T::width, T::sign>((long long)(rnd64() >> (64 - (T::width < 63 ? T::width : 63)))));
  if (rnd64() & 1) { v = (T)(-v); }
  return v;
}

static int test_cic() {
  typedef ac_fixed<48, 20, true> IN_T;
  typedef ac_fixed<78, 50, true> INT_T;   // find_inter_type_cic_dec: log2_ceil(16^6 * 2^6) + 48 = 30 + 48
  const unsigned R = 16, M = 2, N = 6;
  ac_cic_dec_full<IN_T, INT_T, R, M, N> filter;
  INT_T ig[N], dl[N][M];
  for (unsigned i = 0; i < N; i++) { ig[i] = 0; for (unsigned j = 0; j < M; j++) { dl[i][j] = 0; } }
  unsigned cnt = 0;
  int errs = 0;
  for (int call = 0; call < 3; call++) {
    const int n = call == 0 ? 5 * R + 7 : (call == 1 ? 3 : 9 * R);
    ac_channel<IN_T> in;
    ac_channel<INT_T> out;
    std::vector<INT_T> want;
    for (int t = 0; t < n; t++) {
      const IN_T x = rnd_fixed<IN_T>();
      in.write(x);
      const bool valid = cnt == 0;
      for (unsigned i = N - 1; i > 0; i--) { ig[i] = ig[i] + ig[i - 1]; }   // pipelined integrators
      ig[0] = (INT_T)x + ig[0];
      cnt = (cnt + 1 > R - 1) ? 0 : cnt + 1;
      if (valid) {
        INT_T v = ig[N - 1];
        for (unsigned k = 0; k < N; k++) {                                   // combs, ascending delay-line loop
          const INT_T o = v - dl[k][M - 1];
          for (unsigned i = 1; i < M; i++) { dl[k][i] = dl[k][i - 1]; }
          dl[k][0] = v;
          v = o;
        }
        want.push_back(v);
      }
    }
    filter.run(in, out);
    if ((size_t)out.debug_size() != want.size()) { std::cout << "  cic call " << call << ": " << out.debug_size() << " outputs, expected " << want.size() << std::endl; errs++; }
    for (size_t k = 0; k < want.size() && out.available(1); k++) {
      const INT_T got = out.read();
      if (got != want[k]) { if (errs < 5) { std::cout << "  cic mismatch call " << call << " @" << k << std::endl; } errs++; }
    }
  }
  std::cout << "wide CIC (INT_TYPE <78,50>): " << (errs ? "FAILED" : "ok") << std::endl;
  return errs;
}

static int test_fir() {
  typedef ac_fixed<32, 16, true> IN_T;
  typedef ac_fixed<32, 16, true> CF_T;
  typedef ac_fixed<80, 40, true> ACC_T;   // (the ac_types subset forms sums of up to 128 bits: 32 + 80 + 1 fits)
  typedef ac_fixed<72, 40, true, AC_RND, AC_SAT> OUT_T;
  const unsigned NT = 27;
  ac_fir_load_coeffs<IN_T, OUT_T, CF_T, ACC_T, NT, FOLD_ODD> filter;
  ac_channel<IN_T> in;
  ac_channel<CF_T> cch;
  ac_channel<OUT_T> out;
  ac_channel<bool> ld;
  CF_T c[NT];
  for (unsigned i = 0; i < NT; i++) { c[i] = rnd_fixed<CF_T>(); cch.write(c[i]); }
  ld.write(true);
  filter.run(in, cch, out, ld);
  IN_T reg[NT];
  for (unsigned i = 0; i < NT; i++) { reg[i] = 0; }
  std::vector<OUT_T> want;
  for (int t = 0; t < 200; t++) {
    const IN_T x = rnd_fixed<IN_T>();
    in.write(x);
    for (int i = NT - 1; i >= 0; i--) { reg[i] = (i == 0) ? x : reg[i - 1]; }
    ACC_T acc = 0.0;
    for (unsigned i = 0; i < ((NT - 1) / 2) + 1; i++) {
      ACC_T fold;
      if (i == (NT - 1) / 2) { fold = reg[i]; } else { fold = reg[i] + reg[(NT - 1) - i]; }
      acc += c[i] * fold;
    }
    OUT_T o = acc;
    want.push_back(o);
  }
  filter.run(in, cch, out, ld);
  int errs = 0;
  if ((size_t)out.debug_size() != want.size()) { std::cout << "  fir: " << out.debug_size() << " outputs" << std::endl; errs++; }
  for (size_t k = 0; k < want.size() && out.available(1); k++) {
    if (out.read() != want[k]) { if (errs < 5) { std::cout << "  fir mismatch @" << k << std::endl; } errs++; }
  }
  std::cout << "wide FIR (ACC <80,40>, OUT <72,40,RND,SAT>, FOLD_ODD): " << (errs ? "FAILED" : "ok") << std::endl;
  return errs;
}

int main() {
  const int fails = test_cic() + test_fir();
  std::cout << (fails ? "Test FAILED." : "Test PASSED.") << std::endl;
  return fails;
}
