// xcheck_fixed.cpp -- dumps conversions and MAC steps computed with the ac_fixed templates of
// include/ac_types so that tests/test_oracle.py can compare them with the plain-integer rules of the
// oracle (two independent implementations of the AC Datatypes quantisation / overflow semantics).
// Output lines:  C <case> <src_raw> <dst_raw>      conversion  src type -> dst type
//                M <case> <acc_raw> <a_raw> <b_raw> <new_acc_raw>     acc += a * b
#include <ac_fixed.h>

#include <cstdio>
#include <cstdint>

static uint64_t s = 0x1234567ull;
static uint64_t rnd() {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class T> static T from_raw(long long r) { return T::from_raw128((__int128)r); }
template <class T> static long long raw(const T &v) { return (long long)v.raw128(); }
template <class T> static long long rand_raw() {
  // mix of uniform values and edge values (min, max, +-1 LSB, half-way points)
  const int W = T::width;
  uint64_t r = rnd();
  long long lo = T::sign ? (long long)(~0ull << (W - 1)) : 0;
  long long hi = T::sign ? (long long)((1ull << (W - 1)) - 1) : (long long)((1ull << (W < 63 ? W : 63)) - 1);
  switch (r % 11) {
    case 0: return lo;
    case 1: return hi;
    case 2: return T::sign ? -1 : 1;
    case 3: return 0;
    case 4: return lo + 1;
    default: break;
  }
  uint64_t span = (uint64_t)hi - (uint64_t)lo + 1;   // 0 means the full 64-bit range
  return span ? (long long)((uint64_t)lo + ((r >> 3) % span)) : (long long)r;
}

template <class SRC, class DST> static void conv_case(int id, int n) {
  for (int i = 0; i < n; i++) {
    SRC a = from_raw<SRC>(rand_raw<SRC>());
    DST d = a;
    printf("C %d %lld %lld\n", id, raw(a), raw(d));
  }
}
template <class A, class B, class ACC> static void mac_case(int id, int n) {
  for (int i = 0; i < n; i++) {
    ACC acc = from_raw<ACC>(rand_raw<ACC>());
    A a = from_raw<A>(rand_raw<A>());
    B b = from_raw<B>(rand_raw<B>());
    long long before = raw(acc);
    acc += a * b;
    printf("M %d %lld %lld %lld %lld\n", id, before, raw(a), raw(b), raw(acc));
  }
}

#define QS(X) X(AC_TRN, 0) X(AC_RND, 1) X(AC_TRN_ZERO, 2) X(AC_RND_ZERO, 3) X(AC_RND_INF, 4) X(AC_RND_MIN_INF, 5) X(AC_RND_CONV, 6) X(AC_RND_CONV_ODD, 7)

int main() {
  typedef ac_fixed<24, 9, true> SRC_S;
  typedef ac_fixed<20, 6, false> SRC_U;
#define CONV(Q, qi)                                                           \
  conv_case<SRC_S, ac_fixed<12, 5, true, Q, AC_WRAP> >(100 + qi * 4 + 0, 300);     \
  conv_case<SRC_S, ac_fixed<12, 5, true, Q, AC_SAT> >(100 + qi * 4 + 1, 300);      \
  conv_case<SRC_S, ac_fixed<12, 5, true, Q, AC_SAT_ZERO> >(100 + qi * 4 + 2, 300); \
  conv_case<SRC_S, ac_fixed<12, 5, true, Q, AC_SAT_SYM> >(100 + qi * 4 + 3, 300);  \
  conv_case<SRC_S, ac_fixed<11, 4, false, Q, AC_WRAP> >(200 + qi * 4 + 0, 300);    \
  conv_case<SRC_S, ac_fixed<11, 4, false, Q, AC_SAT> >(200 + qi * 4 + 1, 300);     \
  conv_case<SRC_S, ac_fixed<11, 4, false, Q, AC_SAT_ZERO> >(200 + qi * 4 + 2, 300);\
  conv_case<SRC_S, ac_fixed<11, 4, false, Q, AC_SAT_SYM> >(200 + qi * 4 + 3, 300); \
  conv_case<SRC_U, ac_fixed<10, 3, true, Q, AC_SAT> >(300 + qi * 4 + 1, 300);      \
  conv_case<SRC_U, ac_fixed<10, 8, true, Q, AC_SAT_SYM> >(300 + qi * 4 + 3, 300);  \
  mac_case<ac_fixed<12, 4, true>, ac_fixed<10, 2, true>, ac_fixed<18, 7, true, Q, AC_WRAP> >(400 + qi * 4 + 0, 300);     \
  mac_case<ac_fixed<12, 4, true>, ac_fixed<10, 2, true>, ac_fixed<18, 7, true, Q, AC_SAT> >(400 + qi * 4 + 1, 300);      \
  mac_case<ac_fixed<12, 4, true>, ac_fixed<10, 2, true>, ac_fixed<18, 7, true, Q, AC_SAT_ZERO> >(400 + qi * 4 + 2, 300); \
  mac_case<ac_fixed<12, 4, true>, ac_fixed<10, 2, true>, ac_fixed<18, 7, true, Q, AC_SAT_SYM> >(400 + qi * 4 + 3, 300);
  QS(CONV)
  // widening conversions and the reference-test accumulator
  conv_case<ac_fixed<16, 8, true>, ac_fixed<64, 32, true> >(900, 200);
  conv_case<ac_fixed<28, 6, true>, ac_fixed<64, 32, true> >(901, 200);
  mac_case<ac_fixed<28, 6, true>, ac_fixed<23, 7, true>, ac_fixed<64, 32, true> >(902, 300);
  mac_case<ac_fixed<64, 32, true>, ac_fixed<32, 16, true>, ac_fixed<64, 32, true> >(903, 300);
  return 0;
}
