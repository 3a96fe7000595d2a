// ac_int.h -- minimal arbitrary-width integer type for the ac_dsp_amd engine.
//
// This is NOT the hlslibs/ac_types implementation (that package is an
// un-vendored dependency of hlslibs/ac_dsp: see the `#include <ac_int.h>` at
// reference include/ac_dsp/ac_fir_const_coeffs.h:88-90 and the AC_TYPES_INC
// check at reference tests/Makefile:1-3).  It is an independent, from-scratch
// subset written from the published AC Datatypes semantics, sufficient for
//   * the drop-in FIR/CIC class templates in include/ac_dsp/, and
//   * testbenches written in the style of the reference's tests/rtest_*.cpp.
// Design: one __int128 holds the (sign- or zero-extended) value, so W <= 128.
// The engine itself never computes with these host types: run() moves raw
// two's-complement words to the GPU.  The arithmetic operators exist for API
// completeness and as a second, independent model in the parity tests.
#ifndef AC_DSP_AMD_AC_INT_H
#define AC_DSP_AMD_AC_INT_H
#define __AC_INT_H  // the guard name AC Datatypes users test for

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <string>

#ifndef __SIZEOF_INT128__
#error "ac_dsp_amd ac_int.h needs a compiler with __int128 (gcc / clang / hipcc)"
#endif

enum ac_base_mode { AC_BIN = 2, AC_OCT = 8, AC_DEC = 10, AC_HEX = 16 };
enum ac_special_val { AC_VAL_DC, AC_VAL_0, AC_VAL_MIN, AC_VAL_MAX, AC_VAL_QUANTUM };
enum ac_q_mode { AC_TRN, AC_RND, AC_TRN_ZERO, AC_RND_ZERO, AC_RND_INF, AC_RND_MIN_INF, AC_RND_CONV, AC_RND_CONV_ODD };
enum ac_o_mode { AC_WRAP, AC_SAT, AC_SAT_ZERO, AC_SAT_SYM };

typedef long long Slong;
typedef unsigned long long Ulong;

namespace ac_private {
typedef __int128 i128;
typedef unsigned __int128 u128;

// Wrap x to W bits and extend (sign if S, zero otherwise).
inline i128 wrap_bits(i128 x, int W, bool S) {
  if (W >= 128) { return x; }
  u128 m = (((u128)1) << W) - 1;
  u128 u = ((u128)x) & m;
  if (S && ((u >> (W - 1)) & 1)) { u |= ~m; }
  return (i128)u;
}
inline i128 min_val(int W, bool S) { return S ? -(((i128)1) << (W - 1)) : (i128)0; }
inline i128 max_val(int W, bool S) {
  if (S) { return (i128)((((u128)1) << (W - 1)) - 1); }
  return (W >= 128) ? (i128)(~(u128)0 >> 1) : (i128)((((u128)1) << W) - 1);
}
inline double to_double(i128 x) { return (double)x; }
inline std::string to_dec_string(i128 x) {
  if (x == 0) { return "0"; }
  bool neg = x < 0;
  u128 u = neg ? (u128)0 - (u128)x : (u128)x;
  std::string s;
  while (u) { s.insert(s.begin(), char('0' + (int)(u % 10))); u /= 10; }
  if (neg) { s.insert(s.begin(), '-'); }
  return s;
}
template <bool C, class A, class B> struct select { typedef A type; };
template <class A, class B> struct select<false, A, B> { typedef B type; };
template <int A, int B> struct imax { enum { val = (A > B) ? A : B }; };
template <int A, int B> struct imin { enum { val = (A < B) ? A : B }; };

// Implicit conversion to (unsigned) long long exists only for W <= 64, as in AC Datatypes.
template <class Derived, bool S, bool FITS64> struct int_conv {};
template <class Derived> struct int_conv<Derived, true, true> {
  operator Slong() const { return (Slong) static_cast<const Derived *>(this)->raw128(); }
};
template <class Derived> struct int_conv<Derived, false, true> {
  operator Ulong() const { return (Ulong) static_cast<const Derived *>(this)->raw128(); }
};
}  // namespace ac_private

namespace ac {
template <unsigned long long X> struct nbits { enum { val = X ? 1 + nbits<(X >> 1)>::val : 0 }; };
template <> struct nbits<0> { enum { val = 0 }; };
template <unsigned long long X> struct log2_floor { enum { val = nbits<X>::val - 1 }; };
template <> struct log2_floor<0> {};
template <unsigned long long X> struct log2_ceil { enum { lf = log2_floor<X>::val, val = (X == (1ull << lf)) ? lf : lf + 1 }; };
template <> struct log2_ceil<0> {};
}  // namespace ac

template <int W, bool S = true>
class ac_int : public ac_private::int_conv<ac_int<W, S>, S, (W <= 64)> {
  static_assert(W >= 1 && W <= 128, "ac_dsp_amd ac_int supports 1 <= W <= 128");
  typedef ac_private::i128 i128;
  i128 v;

public:
  static const int width = W;
  static const int i_width = W;
  static const bool sign = S;
  static const ac_q_mode q_mode = AC_TRN;
  static const ac_o_mode o_mode = AC_WRAP;
  static const int e_width = 0;

  template <int W2, bool S2> struct rt {
    enum {
      mult_w = W + W2, mult_s = S || S2,
      plus_w = ac_private::imax<W + (S2 && !S), W2 + (S && !S2)>::val + 1, plus_s = S || S2,
      minus_w = ac_private::imax<W + (S2 && !S), W2 + (S && !S2)>::val + 1, minus_s = true,
      div_w = W + S2, div_s = S || S2,
      mod_w = ac_private::imin<W, W2 + (!S2 && S)>::val, mod_s = S,
      logic_w = ac_private::imax<W + (S2 && !S), W2 + (S && !S2)>::val, logic_s = S || S2
    };
    typedef ac_int<mult_w, mult_s> mult;
    typedef ac_int<plus_w, plus_s> plus;
    typedef ac_int<minus_w, minus_s> minus;
    typedef ac_int<logic_w, logic_s> logic;
    typedef ac_int<div_w, div_s> div;
    typedef ac_int<mod_w, mod_s> mod;
    typedef ac_int<W, S> arg1;
  };
  struct rt_unary {
    typedef ac_int<W + 1, true> neg;
    typedef ac_int<W + !S, true> bnot;
  };

  ac_int() : v(0) {}
  ac_int(bool b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(char b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(signed char b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(unsigned char b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(short b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(unsigned short b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(int b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(unsigned b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(long b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(unsigned long b) : v(ac_private::wrap_bits((i128)b, W, S)) {}
  ac_int(Slong b) : v(ac_private::wrap_bits(b, W, S)) {}
  ac_int(Ulong b) : v(ac_private::wrap_bits((i128)b, W, S)) {}
  ac_int(double d) : v(ac_private::wrap_bits((i128)std::trunc(d), W, S)) {}
  template <int W2, bool S2> ac_int(const ac_int<W2, S2> &o) : v(ac_private::wrap_bits(o.raw128(), W, S)) {}

  // Raw access used by ac_fixed and by the engine glue.
  i128 raw128() const { return v; }
  static ac_int from_raw128(i128 x) { ac_int r; r.v = ac_private::wrap_bits(x, W, S); return r; }

  template <ac_special_val V> ac_int &set_val() {
    if (V == AC_VAL_MIN) { v = ac_private::min_val(W, S); }
    else if (V == AC_VAL_MAX) { v = ac_private::max_val(W, S); }
    else if (V == AC_VAL_QUANTUM) { v = 1; }
    else { v = 0; }  // AC_VAL_0, and a defined value for AC_VAL_DC
    return *this;
  }

  int to_int() const { return (int)v; }
  unsigned to_uint() const { return (unsigned)v; }
  long to_long() const { return (long)v; }
  unsigned long to_ulong() const { return (unsigned long)v; }
  Slong to_int64() const { return (Slong)v; }
  Ulong to_uint64() const { return (Ulong)v; }
  double to_double() const { return ac_private::to_double(v); }
  int length() const { return W; }
  std::string to_string(ac_base_mode = AC_DEC, bool = false) const { return ac_private::to_dec_string(v); }

  // arithmetic
  template <int W2, bool S2> typename rt<W2, S2>::mult operator*(const ac_int<W2, S2> &o) const {
    static_assert(W + W2 <= 128, "product wider than 128 bits");
    return rt<W2, S2>::mult::from_raw128(v * o.raw128());
  }
  template <int W2, bool S2> typename rt<W2, S2>::plus operator+(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::plus::from_raw128(v + o.raw128());
  }
  template <int W2, bool S2> typename rt<W2, S2>::minus operator-(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::minus::from_raw128(v - o.raw128());
  }
  template <int W2, bool S2> typename rt<W2, S2>::div operator/(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::div::from_raw128(o.raw128() ? v / o.raw128() : 0);
  }
  template <int W2, bool S2> typename rt<W2, S2>::mod operator%(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::mod::from_raw128(o.raw128() ? v % o.raw128() : 0);
  }
  template <int W2, bool S2> typename rt<W2, S2>::logic operator&(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::logic::from_raw128(v & o.raw128());
  }
  template <int W2, bool S2> typename rt<W2, S2>::logic operator|(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::logic::from_raw128(v | o.raw128());
  }
  template <int W2, bool S2> typename rt<W2, S2>::logic operator^(const ac_int<W2, S2> &o) const {
    return rt<W2, S2>::logic::from_raw128(v ^ o.raw128());
  }
  typename rt_unary::neg operator-() const { return rt_unary::neg::from_raw128(-v); }
  ac_int operator+() const { return *this; }
  typename rt_unary::bnot operator~() const { return rt_unary::bnot::from_raw128(~v); }
  bool operator!() const { return v == 0; }

  template <int W2, bool S2> ac_int &operator+=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v + o.raw128(), W, S); return *this; }
  template <int W2, bool S2> ac_int &operator-=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v - o.raw128(), W, S); return *this; }
  template <int W2, bool S2> ac_int &operator*=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v * o.raw128(), W, S); return *this; }
  template <int W2, bool S2> ac_int &operator&=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v & o.raw128(), W, S); return *this; }
  template <int W2, bool S2> ac_int &operator|=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v | o.raw128(), W, S); return *this; }
  template <int W2, bool S2> ac_int &operator^=(const ac_int<W2, S2> &o) { v = ac_private::wrap_bits(v ^ o.raw128(), W, S); return *this; }
  ac_int &operator++() { v = ac_private::wrap_bits(v + 1, W, S); return *this; }
  ac_int &operator--() { v = ac_private::wrap_bits(v - 1, W, S); return *this; }
  const ac_int operator++(int) { ac_int t = *this; ++*this; return t; }
  const ac_int operator--(int) { ac_int t = *this; --*this; return t; }

  // shifts keep the type of the left operand (AC Datatypes rule)
  ac_int operator<<(int s) const {
    if (s < 0) { return *this >> (-s); }
    return from_raw128(s >= 128 ? (i128)0 : (i128)((ac_private::u128)v << s));
  }
  ac_int operator>>(int s) const {
    if (s < 0) { return *this << (-s); }
    return from_raw128(s >= 128 ? (v < 0 ? (i128)-1 : (i128)0) : (v >> s));
  }
  ac_int &operator<<=(int s) { *this = *this << s; return *this; }
  ac_int &operator>>=(int s) { *this = *this >> s; return *this; }

  template <int W2, bool S2> bool operator==(const ac_int<W2, S2> &o) const { return v == o.raw128(); }
  template <int W2, bool S2> bool operator!=(const ac_int<W2, S2> &o) const { return v != o.raw128(); }
  template <int W2, bool S2> bool operator<(const ac_int<W2, S2> &o) const { return v < o.raw128(); }
  template <int W2, bool S2> bool operator>(const ac_int<W2, S2> &o) const { return v > o.raw128(); }
  template <int W2, bool S2> bool operator<=(const ac_int<W2, S2> &o) const { return v <= o.raw128(); }
  template <int W2, bool S2> bool operator>=(const ac_int<W2, S2> &o) const { return v >= o.raw128(); }

  // bit select / slices
  bool operator[](int i) const { return (i >= 0 && i < 128) ? (bool)((v >> i) & 1) : (v < 0); }
  template <int WS> ac_int<WS, S> slc(int lsb) const { return ac_int<WS, S>::from_raw128(v >> lsb); }
  template <int W2, bool S2> ac_int &set_slc(int lsb, const ac_int<W2, S2> &s) {
    ac_private::u128 m = (W2 >= 128) ? ~(ac_private::u128)0 : ((((ac_private::u128)1) << W2) - 1);
    ac_private::u128 u = ((ac_private::u128)v & ~(m << lsb)) | ((((ac_private::u128)s.raw128()) & m) << lsb);
    v = ac_private::wrap_bits((i128)u, W, S);
    return *this;
  }
};

// Mixed ac_int / C-integer operators: the C operand is promoted to the ac_int of its own width.
#define AC_DSP_AMD_INT_OPS(CT, CW, CS)                                                                              \
  template <int W, bool S> inline typename ac_int<W, S>::template rt<CW, CS>::plus operator+(const ac_int<W, S> &a, CT b) { return a + ac_int<CW, CS>(b); } \
  template <int W, bool S> inline typename ac_int<CW, CS>::template rt<W, S>::plus operator+(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) + a; } \
  template <int W, bool S> inline typename ac_int<W, S>::template rt<CW, CS>::minus operator-(const ac_int<W, S> &a, CT b) { return a - ac_int<CW, CS>(b); } \
  template <int W, bool S> inline typename ac_int<CW, CS>::template rt<W, S>::minus operator-(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) - a; } \
  template <int W, bool S> inline typename ac_int<W, S>::template rt<CW, CS>::mult operator*(const ac_int<W, S> &a, CT b) { return a * ac_int<CW, CS>(b); } \
  template <int W, bool S> inline typename ac_int<CW, CS>::template rt<W, S>::mult operator*(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) * a; } \
  template <int W, bool S> inline ac_int<W, S> &operator+=(ac_int<W, S> &a, CT b) { return a += ac_int<CW, CS>(b); } \
  template <int W, bool S> inline ac_int<W, S> &operator-=(ac_int<W, S> &a, CT b) { return a -= ac_int<CW, CS>(b); } \
  template <int W, bool S> inline bool operator==(const ac_int<W, S> &a, CT b) { return a == ac_int<CW, CS>(b); } \
  template <int W, bool S> inline bool operator!=(const ac_int<W, S> &a, CT b) { return a != ac_int<CW, CS>(b); } \
  template <int W, bool S> inline bool operator<(const ac_int<W, S> &a, CT b) { return a < ac_int<CW, CS>(b); }   \
  template <int W, bool S> inline bool operator>(const ac_int<W, S> &a, CT b) { return a > ac_int<CW, CS>(b); }   \
  template <int W, bool S> inline bool operator<=(const ac_int<W, S> &a, CT b) { return a <= ac_int<CW, CS>(b); } \
  template <int W, bool S> inline bool operator>=(const ac_int<W, S> &a, CT b) { return a >= ac_int<CW, CS>(b); } \
  template <int W, bool S> inline bool operator==(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) == a; } \
  template <int W, bool S> inline bool operator!=(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) != a; } \
  template <int W, bool S> inline bool operator<(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) < a; }   \
  template <int W, bool S> inline bool operator>(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) > a; }   \
  template <int W, bool S> inline bool operator<=(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) <= a; } \
  template <int W, bool S> inline bool operator>=(CT b, const ac_int<W, S> &a) { return ac_int<CW, CS>(b) >= a; }

AC_DSP_AMD_INT_OPS(bool, 1, false)
AC_DSP_AMD_INT_OPS(char, 8, true)
AC_DSP_AMD_INT_OPS(signed char, 8, true)
AC_DSP_AMD_INT_OPS(unsigned char, 8, false)
AC_DSP_AMD_INT_OPS(short, 16, true)
AC_DSP_AMD_INT_OPS(unsigned short, 16, false)
AC_DSP_AMD_INT_OPS(int, 32, true)
AC_DSP_AMD_INT_OPS(unsigned int, 32, false)
AC_DSP_AMD_INT_OPS(long, 64, true)
AC_DSP_AMD_INT_OPS(unsigned long, 64, false)
AC_DSP_AMD_INT_OPS(Slong, 64, true)
AC_DSP_AMD_INT_OPS(Ulong, 64, false)
#undef AC_DSP_AMD_INT_OPS

template <int W, bool S> inline std::ostream &operator<<(std::ostream &os, const ac_int<W, S> &x) {
  os << x.to_string(AC_DEC);
  return os;
}

namespace ac {
// ac::init_array<AC_VAL_*>(array, n): reference call sites ac_fir_const_coeffs.h:145-146,
// ac_fir_load_coeffs.h:311, ac_cic_full_core.h:101.
template <ac_special_val V, class T> inline bool init_array(T *a, int n) {
  T t;
  t.template set_val<V>();
  for (int i = 0; i < n; i++) { a[i] = t; }
  return true;
}
}  // namespace ac

#endif
