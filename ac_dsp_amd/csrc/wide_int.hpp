// wide_int.hpp -- 256-bit two's-complement integer for the exact intermediates of types wider than 64 bits.
//
// The reference's `acc += reg[i] * coeffs[i]` forms an exact product and an exact aligned sum before the ACC_TYPE
// quantisation / overflow (reference include/ac_dsp/ac_fir_const_coeffs.h:196); with an accumulator of up to 128 bits
// (its FIR testbenches already multiply <32> x <64>, tests/rtest_ac_fir_const_coeffs.cpp:71-74, and ac_cic_dec_full derives
// INT_TYPE of any width, ac_cic_dec_full.h:132) those intermediates need up to 128 + 64 + alignment bits.  Four 64-bit limbs,
// little endian; only what the exact-order kernels use.  Correctness path: speed is not a goal here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acdsp {

struct i256 {
  uint64_t w[4];

  __host__ __device__ i256() : w{0, 0, 0, 0} {}
  __host__ __device__ i256(__int128 v) {
    const unsigned __int128 u = (unsigned __int128)v;
    const uint64_t ext = v < 0 ? ~uint64_t(0) : 0;
    w[0] = (uint64_t)u; w[1] = (uint64_t)(u >> 64); w[2] = ext; w[3] = ext;
  }
  __host__ __device__ i256(int64_t v) : i256((__int128)v) {}
  __host__ __device__ i256(int v) : i256((__int128)v) {}

  __host__ __device__ bool neg() const { return (w[3] >> 63) != 0; }
  __host__ __device__ bool zero() const { return (w[0] | w[1] | w[2] | w[3]) == 0; }
  // (limb picked by selects, not by a run-time index: an indexed private array lives in scratch memory on the device)
  __host__ __device__ int bit(int k) const {
    if (k < 0) { return 0; }
    if (k > 255) { return (int)neg(); }
    const int q = k >> 6;
    const uint64_t v = q == 0 ? w[0] : (q == 1 ? w[1] : (q == 2 ? w[2] : w[3]));
    return (int)((v >> (k & 63)) & 1);
  }
  // OR of bits [0, k)
  __host__ __device__ bool any_below(int k) const {
    if (k <= 0) { return false; }
    if (k > 256) { k = 256; }
    uint64_t acc = 0;
    for (int i = 0; i < 4; i++) {
      const int lo = 64 * i;
      if (k >= lo + 64) { acc |= w[i]; }
      else if (k > lo) { acc |= w[i] & ((uint64_t(1) << (k - lo)) - 1); }
    }
    return acc != 0;
  }
  // low 128 bits
  __host__ __device__ __int128 low128() const { return (__int128)(((unsigned __int128)w[1] << 64) | w[0]); }
  // does the value fit a signed 128-bit integer?
  __host__ __device__ bool fits128() const {
    const uint64_t ext = (w[1] >> 63) ? ~uint64_t(0) : 0;
    return w[2] == ext && w[3] == ext;
  }
};

__host__ __device__ inline i256 operator+(const i256 &a, const i256 &b) {
  i256 r;
  unsigned __int128 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
__host__ __device__ inline i256 operator~(const i256 &a) {
  i256 r;
  for (int i = 0; i < 4; i++) { r.w[i] = ~a.w[i]; }
  return r;
}
__host__ __device__ inline i256 operator-(const i256 &a) { return ~a + i256(1); }
__host__ __device__ inline i256 operator-(const i256 &a, const i256 &b) { return a + (-b); }
// product modulo 2^256 (two's complement: the low 256 bits of the signed product)
__host__ __device__ inline i256 operator*(const i256 &a, const i256 &b) {
  i256 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    unsigned __int128 c = 0;
#pragma unroll
    for (int j = 0; j < 4 - i; j++) {
      c += (unsigned __int128)a.w[i] * b.w[j] + r.w[i + j];
      r.w[i + j] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
// limb a.w[k] for a run-time k by selects (k outside 0..3: `ext`); see bit()
__host__ __device__ inline uint64_t limb_or(const i256 &a, int k, uint64_t ext) {
  return k == 0 ? a.w[0] : (k == 1 ? a.w[1] : (k == 2 ? a.w[2] : (k == 3 ? a.w[3] : ext)));
}
__host__ __device__ inline i256 shl(const i256 &a, int s) {
  i256 r;
  if (s <= 0) { return s == 0 ? a : r; }
  if (s >= 256) { return r; }
  const int q = s >> 6, b = s & 63;
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    uint64_t v = limb_or(a, i - q, 0) << b;
    if (b) { v |= limb_or(a, i - q - 1, 0) >> (64 - b); }
    r.w[i] = v;
  }
  return r;
}
// arithmetic right shift (floor division by 2^s)
__host__ __device__ inline i256 sar(const i256 &a, int s) {
  if (s <= 0) { return a; }
  const uint64_t ext = a.neg() ? ~uint64_t(0) : 0;
  i256 r;
  if (s >= 256) { for (int i = 0; i < 4; i++) { r.w[i] = ext; } return r; }
  const int q = s >> 6, b = s & 63;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint64_t lo = limb_or(a, i + q, ext), hi = limb_or(a, i + q + 1, ext);
    r.w[i] = b ? ((lo >> b) | (hi << (64 - b))) : lo;
  }
  return r;
}
__host__ __device__ inline bool operator==(const i256 &a, const i256 &b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}
__host__ __device__ inline bool operator<(const i256 &a, const i256 &b) {
  if (a.neg() != b.neg()) { return a.neg(); }
  for (int i = 3; i >= 0; i--) {
    if (a.w[i] != b.w[i]) { return a.w[i] < b.w[i]; }
  }
  return false;
}
__host__ __device__ inline bool operator>(const i256 &a, const i256 &b) { return b < a; }

}  // namespace acdsp
