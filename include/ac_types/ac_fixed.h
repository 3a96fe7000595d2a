// ac_fixed.h -- minimal fixed-point type for the ac_dsp_amd engine.
//
// Independent from-scratch subset of the AC Datatypes `ac_fixed<W,I,S,Q,O>`
// (hlslibs/ac_types is an un-vendored dependency of hlslibs/ac_dsp; the
// reference includes it at include/ac_dsp/ac_fir_const_coeffs.h:88 and
// include/ac_dsp/ac_cic_full_core.h:50).  Semantics follow the published AC
// Datatypes rules:
//   value  = raw * 2^-(W-I), raw is a W-bit two's-complement (S) / unsigned word
//   a*b   -> ac_fixed<W1+W2, I1+I2, S1||S2>                       (exact)
//   a+b   -> ac_fixed<max(I1+(S2&&!S1), I2+(S1&&!S2))+1+max(F1,F2), ..+1, S1||S2>
//   a-b   -> same width, always signed
//   assignment / construction quantises with Q, then handles overflow with O;
//   a rounding carry takes part in the overflow decision.
// The same rules are restated on plain integers in oracle/acdsp_oracle.c; the
// test-suite cross-checks the two implementations against each other.
#ifndef AC_DSP_AMD_AC_FIXED_H
#define AC_DSP_AMD_AC_FIXED_H
#define __AC_FIXED_H

#include <ac_int.h>

namespace ac_private {

// Quantise x * 2^-f_src to f_dst fractional bits with mode Q (no overflow handling).
inline i128 quantize(i128 x, int f_src, int f_dst, ac_q_mode Q) {
  int sh = f_src - f_dst;
  if (sh <= 0) { return (-sh >= 128) ? (i128)0 : (i128)((u128)x << (-sh)); }
  bool neg = x < 0;
  bool sticky = false;
  if (sh > 126) {  // pre-shift with a sticky bit so the main path sees sh <= 126
    int k = sh - 126;
    i128 xs = (k >= 128) ? (neg ? (i128)-1 : (i128)0) : (x >> k);
    sticky = (k >= 128) ? (x != 0) : ((x - (i128)((u128)xs << k)) != 0);
    x = xs;
    sh = 126;
  }
  i128 q = x >> sh;                              // floor
  i128 rem = x - (i128)((u128)q << sh);          // 0 <= rem < 2^sh
  i128 half = ((i128)1) << (sh - 1);
  bool qb = rem >= half;                         // MSB of the dropped field
  bool r = ((rem & (half - 1)) != 0) || sticky;  // any lower dropped bit set
  bool lsb = (bool)(q & 1);
  bool inc = false;
  switch (Q) {
    case AC_TRN: inc = false; break;
    case AC_RND: inc = qb; break;
    case AC_TRN_ZERO: inc = neg && (qb || r); break;
    case AC_RND_ZERO: inc = qb && (r || neg); break;
    case AC_RND_INF: inc = qb && (r || !neg); break;
    case AC_RND_MIN_INF: inc = qb && r; break;
    case AC_RND_CONV: inc = qb && (r || lsb); break;
    case AC_RND_CONV_ODD: inc = qb && (r || !lsb); break;
  }
  return q + (inc ? 1 : 0);
}

// Fit q into a W-bit (S) word with overflow mode O.
inline i128 overflow(i128 q, int W, bool S, ac_o_mode O) {
  i128 lo = min_val(W, S), hi = max_val(W, S);
  bool ovf = (q < lo) || (q > hi);
  switch (O) {
    case AC_WRAP: return wrap_bits(q, W, S);
    case AC_SAT: return ovf ? ((q < lo) ? lo : hi) : q;
    case AC_SAT_ZERO: return ovf ? (i128)0 : q;
    case AC_SAT_SYM:
      if (S) {
        if (ovf) { return (q < 0) ? lo + 1 : hi; }
        return (q == lo && W > 1) ? lo + 1 : q;
      }
      return ovf ? ((q < lo) ? lo : hi) : q;
  }
  return q;
}

inline i128 convert(i128 x, int f_src, int W, int F, bool S, ac_q_mode Q, ac_o_mode O) {
  return overflow(quantize(x, f_src, F, Q), W, S, O);
}

// Exact decomposition of a finite double: d == m * 2^e with |m| < 2^53.
inline void split_double(double d, i128 &m, int &e) {
  if (d == 0.0 || d != d) { m = 0; e = 0; return; }
  int ex;
  double fr = frexp(d, &ex);           // d = fr * 2^ex, 0.5 <= |fr| < 1
  m = (i128)(long long)ldexp(fr, 53);  // exact
  e = ex - 53;
}
}  // namespace ac_private

template <int W, int I, bool S = true, ac_q_mode Q = AC_TRN, ac_o_mode O = AC_WRAP>
class ac_fixed {
  static_assert(W >= 1 && W <= 128, "ac_dsp_amd ac_fixed supports 1 <= W <= 128");
  typedef ac_private::i128 i128;
  i128 v;  // raw word, extended to 128 bits

  static i128 conv(i128 x, int f_src) { return ac_private::convert(x, f_src, W, W - I, S, Q, O); }
  static i128 conv_double(double d) {
    i128 m; int e;
    ac_private::split_double(d, m, e);
    // value = m * 2^e = m * 2^-(-e)
    if (-e < -(127 - 54)) {  // enormous magnitude: saturate/wrap as an out-of-range value
      i128 big = (d < 0) ? ac_private::min_val(128, true) / 2 : ac_private::max_val(128, true) / 2;
      return ac_private::overflow(O == AC_WRAP ? (i128)0 : big, W, S, O);
    }
    return conv(m, -e);
  }

public:
  static const int width = W;
  static const int i_width = I;
  static const bool sign = S;
  static const ac_q_mode q_mode = Q;
  static const ac_o_mode o_mode = O;
  static const int e_width = 0;

  template <int W2, int I2, bool S2> struct rt {
    enum {
      F = W - I, F2 = W2 - I2,
      mult_w = W + W2, mult_i = I + I2, mult_s = S || S2,
      plus_i = ac_private::imax<I + (S2 && !S), I2 + (S && !S2)>::val + 1,
      plus_w = plus_i + ac_private::imax<F, F2>::val, plus_s = S || S2,
      minus_w = plus_w, minus_i = plus_i, minus_s = true,
      logic_w = ac_private::imax<I + (S2 && !S), I2 + (S && !S2)>::val + ac_private::imax<F, F2>::val,
      logic_i = ac_private::imax<I + (S2 && !S), I2 + (S && !S2)>::val, logic_s = S || S2
    };
    typedef ac_fixed<mult_w, mult_i, mult_s> mult;
    typedef ac_fixed<plus_w, plus_i, plus_s> plus;
    typedef ac_fixed<minus_w, minus_i, minus_s> minus;
    typedef ac_fixed<logic_w, logic_i, logic_s> logic;
    typedef ac_fixed<W, I, S> arg1;
  };
  struct rt_unary {
    typedef ac_fixed<W + 1, I + 1, true> neg;
  };

  ac_fixed() : v(0) {}
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  ac_fixed(const ac_fixed<W2, I2, S2, Q2, O2> &o) : v(conv(o.raw128(), W2 - I2)) {}
  template <int W2, bool S2> ac_fixed(const ac_int<W2, S2> &o) : v(conv(o.raw128(), 0)) {}
  ac_fixed(bool b) : v(conv((i128)b, 0)) {}
  ac_fixed(char b) : v(conv((i128)b, 0)) {}
  ac_fixed(signed char b) : v(conv((i128)b, 0)) {}
  ac_fixed(unsigned char b) : v(conv((i128)b, 0)) {}
  ac_fixed(short b) : v(conv((i128)b, 0)) {}
  ac_fixed(unsigned short b) : v(conv((i128)b, 0)) {}
  ac_fixed(int b) : v(conv((i128)b, 0)) {}
  ac_fixed(unsigned b) : v(conv((i128)b, 0)) {}
  ac_fixed(long b) : v(conv((i128)b, 0)) {}
  ac_fixed(unsigned long b) : v(conv((i128)b, 0)) {}
  ac_fixed(Slong b) : v(conv((i128)b, 0)) {}
  ac_fixed(Ulong b) : v(conv((i128)b, 0)) {}
  ac_fixed(double d) : v(conv_double(d)) {}
  ac_fixed(float d) : v(conv_double((double)d)) {}

  // Raw word access (engine glue and tests).
  i128 raw128() const { return v; }
  static ac_fixed from_raw128(i128 x) { ac_fixed r; r.v = ac_private::wrap_bits(x, W, S); return r; }

  template <ac_special_val V> ac_fixed &set_val() {
    if (V == AC_VAL_MIN) { v = (O == AC_SAT_SYM && S && W > 1) ? ac_private::min_val(W, S) + 1 : ac_private::min_val(W, S); }
    else if (V == AC_VAL_MAX) { v = ac_private::max_val(W, S); }
    else if (V == AC_VAL_QUANTUM) { v = 1; }
    else { v = 0; }
    return *this;
  }

  double to_double() const { return ldexp(ac_private::to_double(v), -(W - I)); }
  long double to_long_double() const { return ldexpl((long double)v, -(W - I)); }
  int to_int() const { return (int)ac_private::quantize(v, W - I, 0, AC_TRN); }
  unsigned to_uint() const { return (unsigned)ac_private::quantize(v, W - I, 0, AC_TRN); }
  long to_long() const { return (long)ac_private::quantize(v, W - I, 0, AC_TRN); }
  Slong to_int64() const { return (Slong)ac_private::quantize(v, W - I, 0, AC_TRN); }
  Ulong to_uint64() const { return (Ulong)ac_private::quantize(v, W - I, 0, AC_TRN); }
  ac_int<ac_private::imax<I, 1>::val, S> to_ac_int() const {
    return ac_int<ac_private::imax<I, 1>::val, S>::from_raw128(ac_private::quantize(v, W - I, 0, AC_TRN));
  }
  int length() const { return W; }
  std::string to_string(ac_base_mode = AC_DEC, bool = false) const {
    char buf[80];
    snprintf(buf, sizeof buf, "%.21Lg", to_long_double());
    return std::string(buf);
  }

  // arithmetic (exact result types)
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  typename rt<W2, I2, S2>::mult operator*(const ac_fixed<W2, I2, S2, Q2, O2> &o) const {
    static_assert(W + W2 <= 128, "product wider than 128 bits");
    return rt<W2, I2, S2>::mult::from_raw128(v * o.raw128());
  }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  typename rt<W2, I2, S2>::plus operator+(const ac_fixed<W2, I2, S2, Q2, O2> &o) const {
    enum { F = W - I, F2 = W2 - I2, FM = ac_private::imax<F, F2>::val };
    static_assert((int)rt<W2, I2, S2>::plus_w <= 128, "sum wider than 128 bits");
    return rt<W2, I2, S2>::plus::from_raw128((i128)((ac_private::u128)v << (FM - F)) + (i128)((ac_private::u128)o.raw128() << (FM - F2)));
  }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  typename rt<W2, I2, S2>::minus operator-(const ac_fixed<W2, I2, S2, Q2, O2> &o) const {
    enum { F = W - I, F2 = W2 - I2, FM = ac_private::imax<F, F2>::val };
    static_assert((int)rt<W2, I2, S2>::minus_w <= 128, "difference wider than 128 bits");
    return rt<W2, I2, S2>::minus::from_raw128((i128)((ac_private::u128)v << (FM - F)) - (i128)((ac_private::u128)o.raw128() << (FM - F2)));
  }
  typename rt_unary::neg operator-() const { return rt_unary::neg::from_raw128(-v); }
  ac_fixed operator+() const { return *this; }
  bool operator!() const { return v == 0; }

  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  ac_fixed &operator+=(const ac_fixed<W2, I2, S2, Q2, O2> &o) { *this = this->operator+(o); return *this; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  ac_fixed &operator-=(const ac_fixed<W2, I2, S2, Q2, O2> &o) { *this = this->operator-(o); return *this; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2>
  ac_fixed &operator*=(const ac_fixed<W2, I2, S2, Q2, O2> &o) { *this = this->operator*(o); return *this; }

  // shifts keep the left operand's type: bits shifted out are lost (no rounding/saturation)
  ac_fixed operator<<(int s) const {
    if (s < 0) { return *this >> (-s); }
    return from_raw128(s >= 128 ? (i128)0 : (i128)((ac_private::u128)v << s));
  }
  ac_fixed operator>>(int s) const {
    if (s < 0) { return *this << (-s); }
    return from_raw128(s >= 128 ? (v < 0 ? (i128)-1 : (i128)0) : (v >> s));
  }
  ac_fixed &operator<<=(int s) { *this = *this << s; return *this; }
  ac_fixed &operator>>=(int s) { *this = *this >> s; return *this; }

  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> int cmp(const ac_fixed<W2, I2, S2, Q2, O2> &o) const {
    enum { F = W - I, F2 = W2 - I2, FM = ac_private::imax<F, F2>::val };
    // Compare by integer part first (no overflow), then by aligned fraction.
    i128 ai = v >> F, bi = o.raw128() >> F2;  // floor of each value (F,F2 < 128 given W<=128 in practice)
    if (ai != bi) { return ai < bi ? -1 : 1; }
    // (shifts on the unsigned image: `ai << F` of a negative ai is undefined before C++20 -- found by the UBSan build of tb_tiny)
    ac_private::u128 af = ((ac_private::u128)v - ((ac_private::u128)ai << F)) << (FM - F);
    ac_private::u128 bf = ((ac_private::u128)o.raw128() - ((ac_private::u128)bi << F2)) << (FM - F2);
    return af < bf ? -1 : (af > bf ? 1 : 0);
  }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator==(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) == 0; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator!=(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) != 0; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator<(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) < 0; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator>(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) > 0; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator<=(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) <= 0; }
  template <int W2, int I2, bool S2, ac_q_mode Q2, ac_o_mode O2> bool operator>=(const ac_fixed<W2, I2, S2, Q2, O2> &o) const { return cmp(o) >= 0; }
  bool operator==(double d) const { return to_long_double() == (long double)d; }
  bool operator!=(double d) const { return to_long_double() != (long double)d; }
  bool operator<(double d) const { return to_long_double() < (long double)d; }
  bool operator>(double d) const { return to_long_double() > (long double)d; }
  bool operator<=(double d) const { return to_long_double() <= (long double)d; }
  bool operator>=(double d) const { return to_long_double() >= (long double)d; }

  // bit select / slices over the raw word (LSB = bit 0), as in AC Datatypes
  bool operator[](int i) const { return (i >= 0 && i < 128) ? (bool)((v >> i) & 1) : (v < 0); }
  template <int WS> ac_int<WS, S> slc(int lsb) const { return ac_int<WS, S>::from_raw128(v >> lsb); }
  template <int W2, bool S2> ac_fixed &set_slc(int lsb, const ac_int<W2, S2> &s) {
    ac_private::u128 m = (W2 >= 128) ? ~(ac_private::u128)0 : ((((ac_private::u128)1) << W2) - 1);
    ac_private::u128 u = ((ac_private::u128)v & ~(m << lsb)) | ((((ac_private::u128)s.raw128()) & m) << lsb);
    v = ac_private::wrap_bits((i128)u, W, S);
    return *this;
  }
};

// Mixed ac_fixed / C-integer operators: the C operand becomes ac_fixed<bits,bits,signed>.
#define AC_DSP_AMD_FX_OPS(CT, CW, CS)                                                                                 \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<W, I, S, Q, O>::template rt<CW, CW, CS>::plus operator+(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a + ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<CW, CW, CS>::template rt<W, I, S>::plus operator+(CT b, const ac_fixed<W, I, S, Q, O> &a) { return ac_fixed<CW, CW, CS>(b) + a; } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<W, I, S, Q, O>::template rt<CW, CW, CS>::minus operator-(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a - ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<CW, CW, CS>::template rt<W, I, S>::minus operator-(CT b, const ac_fixed<W, I, S, Q, O> &a) { return ac_fixed<CW, CW, CS>(b) - a; } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<W, I, S, Q, O>::template rt<CW, CW, CS>::mult operator*(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a * ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>                                                           \
  inline typename ac_fixed<CW, CW, CS>::template rt<W, I, S>::mult operator*(CT b, const ac_fixed<W, I, S, Q, O> &a) { return ac_fixed<CW, CW, CS>(b) * a; } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline ac_fixed<W, I, S, Q, O> &operator+=(ac_fixed<W, I, S, Q, O> &a, CT b) { return a += ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline ac_fixed<W, I, S, Q, O> &operator-=(ac_fixed<W, I, S, Q, O> &a, CT b) { return a -= ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator==(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a == ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator!=(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a != ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator<(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a < ac_fixed<CW, CW, CS>(b); }   \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator>(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a > ac_fixed<CW, CW, CS>(b); }   \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator<=(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a <= ac_fixed<CW, CW, CS>(b); } \
  template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O> inline bool operator>=(const ac_fixed<W, I, S, Q, O> &a, CT b) { return a >= ac_fixed<CW, CW, CS>(b); }

AC_DSP_AMD_FX_OPS(bool, 1, false)
AC_DSP_AMD_FX_OPS(char, 8, true)
AC_DSP_AMD_FX_OPS(signed char, 8, true)
AC_DSP_AMD_FX_OPS(unsigned char, 8, false)
AC_DSP_AMD_FX_OPS(short, 16, true)
AC_DSP_AMD_FX_OPS(unsigned short, 16, false)
AC_DSP_AMD_FX_OPS(int, 32, true)
AC_DSP_AMD_FX_OPS(unsigned int, 32, false)
AC_DSP_AMD_FX_OPS(long, 64, true)
AC_DSP_AMD_FX_OPS(unsigned long, 64, false)
AC_DSP_AMD_FX_OPS(Slong, 64, true)
AC_DSP_AMD_FX_OPS(Ulong, 64, false)
#undef AC_DSP_AMD_FX_OPS

// ac_fixed (op) ac_int
template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O, int W2, bool S2>
inline typename ac_fixed<W, I, S, Q, O>::template rt<W2, W2, S2>::mult operator*(const ac_fixed<W, I, S, Q, O> &a, const ac_int<W2, S2> &b) { return a * ac_fixed<W2, W2, S2>(b); }
template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O, int W2, bool S2>
inline typename ac_fixed<W, I, S, Q, O>::template rt<W2, W2, S2>::plus operator+(const ac_fixed<W, I, S, Q, O> &a, const ac_int<W2, S2> &b) { return a + ac_fixed<W2, W2, S2>(b); }
template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O, int W2, bool S2>
inline typename ac_fixed<W, I, S, Q, O>::template rt<W2, W2, S2>::minus operator-(const ac_fixed<W, I, S, Q, O> &a, const ac_int<W2, S2> &b) { return a - ac_fixed<W2, W2, S2>(b); }

template <int W, int I, bool S, ac_q_mode Q, ac_o_mode O>
inline std::ostream &operator<<(std::ostream &os, const ac_fixed<W, I, S, Q, O> &x) {
  os << x.to_string(AC_DEC);
  return os;
}

#endif
