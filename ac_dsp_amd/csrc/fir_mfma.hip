// fir_mfma.hip -- many-channel 16-bit FIR on the gfx950 matrix cores (exact integer arithmetic).
//
// Replaces the tap-MAC loops of the reference cores (`acc += reg[i] * coeffs[i]`,
// reference include/ac_dsp/ac_fir_const_coeffs.h:190-199 and the load_/prog_ twins) for the
// arithmetic class in which that loop is an exact integer dot product: accumulator with at least
// F_in + F_coeff fractional bits and AC_WRAP overflow (DESIGN.md "arithmetic classes").  The
// folded architectures (:244-275) enter through their effective direct-form coefficients.
//
// Why MFMA at all: a 255-tap int16 FIR carries 255 MAC per 4 algorithmic bytes; the VALU
// (v_dot2_i32_i16) tops out near 15 % of the HBM roofline, the int8 matrix pipe does not.
//
// Formulation.  For one block of 32 consecutive outputs of one channel,
//     y[T0+i] = sum_k c[k] x[T0+i-k]        i = 0..31
// is a [32 x 32*NB] Toeplitz matrix (built from c, the same for every output block of the
// channel) times the [32*NB] input samples ending at T0+31, NB = ceil((N-1)/32)+1.  The 32 MFMA
// columns are 32 CONSECUTIVE output blocks of the same channel:
//     D[i][n] = y[T0 + 32n + i] = sum_b sum_k A_b[i][k] * X_b[k][n],
//     A_b[i][k] = c[i - k + 32*(NB-1-b)],     X_b[k][n] = x[T0 - 32(NB-1) + 32(n+b) + k],
// so one step (4*NB MFMA 32x32x32) produces 1024 consecutive outputs of one channel from one
// contiguous (32+NB-1)*64-byte stretch of its row.
// MFMA has no int16 operand type, so both operands are split into two signed bytes:
//     c = 256*ch + cl            (cl = sign-extended low byte, ch = (c - cl)/256, both int8)
//     x = 256*xh + xl + 128      (xh = high byte, xl = low byte re-biased to signed)
//     y = 65536*S(ch,xh) + 256*(S(ch,xl) + S(cl,xh)) + S(cl,xl) + 128*sum(c)
// Each S is an int32 MFMA accumulation (|S| <= 32*NB*2^14 < 2^31).  The result is the exact
// integer dot product; rounding/saturation into OUT_TYPE happens once, in the epilogue.
//
// Data movement.  The 2*NB A fragments (4 VGPRs each) stay in registers for the whole kernel.
// Column n of K-block b is "input chunk n+b": the X fragments of the NB K-blocks are lane-shifted
// copies of each other.  The shift is done by LDS addressing: a step's chunks are fetched with fully
// coalesced 16-byte loads (every HBM visit of a row moves 2+ KB -- a first version that put 32
// channels in the columns touched 32 rows x 64 B per step, ~130k interleaved DRAM streams, and
// stalled near 2 TB/s), split into byte planes with v_perm_b32, staged in LDS (2.5 KB per wave) and
// each K-block's fragment is a contiguous, conflict-free ds_read_b128 at offset 16*(n+b).
// One wave = one channel x one time chunk; the Toeplitz fragments are per coefficient set, so
// per-channel coefficients cost nothing extra.
//
// Scheduling.  On gfx950 the int8 MFMA run and the rest of a SIMD's instruction stream serialise: time per
// step ~ (#MFMA x 32 cycles) + (#other instructions x ~4 cycles), whatever the wave pairing (measured:
// two free-running waves per SIMD, an 8-wave ping-pong with s_barrier role swaps, and a software-pipelined
// 1 MFMA : 4 VALU interleave all land within a few percent, the ping-pong 4 % behind).  The kernel therefore
// uses independent single-wave workgroups and spends its effort on instruction count: zero high-byte Toeplitz
// blocks are skipped, the epilogue is 4 VALU ops per output plus a clamping pack, outputs leave through a
// swizzled LDS tile as two 16-byte-per-lane stores.  The 8-wave ping-pong form stays selectable (kSmallWaves)
// and is what the large-tap kernel uses, where the Toeplitz fragments are shared through LDS.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fir_kernels.hpp"

namespace acdsp {

// AC_RND and AC_RND_MIN_INF are "add a constant, then floor" (2^(rs-1) and 2^(rs-1) - 1): the constant rides in the preloaded low-plane
// accumulator of every fast epilogue class, like AC_TRN's zero
__host__ __device__ static inline bool q_const_mode(int q) { return q == ACDSP_TRN || q == ACDSP_RND || q == ACDSP_RND_MIN_INF; }
__host__ __device__ static inline int64_t q_preload(int q, int rs) {
  if (rs <= 0 || rs > 62) { return 0; }
  return q == ACDSP_RND ? (int64_t(1) << (rs - 1)) : (q == ACDSP_RND_MIN_INF ? (int64_t(1) << (rs - 1)) - 1 : 0);
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));

constexpr int kMaxRegNB = 9;  // register-resident Toeplitz fragments: up to 32*8+1 = 257 taps
constexpr int kMaxNB = 33;    // Toeplitz fragments in LDS (shared coefficient set): up to 1025 taps

#ifndef ACDSP_FIR_TU_MID   // fir_mfma_mid.hip re-includes this file for the kernels and launch_nb_hs only
static bool use_reg33(int nb, uint64_t hi_mask, int epi);
static bool use_mid(int nb, uint64_t hi_mask, int epi);
int fir_mfma_max_blocks() { return kMaxNB; }
int fir_mfma_max_reg_blocks() { return kMaxRegNB; }

// Host: build frag[plane][b][lane][4 dwords]; plane 0 = high bytes, 1 = low bytes.
// Lane l holds row i = l & 31 and the 16 K-positions k = 16*(l>>5) + j, j = 0..15 (the same
// (lane-group, byte) -> k map is used for the data operand, so any K permutation inside the
// instruction cancels).
// K-blocks of the plan for n_taps: the Toeplitz band of 32 outputs spans n_taps + 31 inputs; 10 .. 32 blocks are padded to the next odd
// count (the register-resident shapes of fir_mfma_mid.hip exist for odd counts; the extra leading block holds zeros).
// ... and the count with the padding applied whatever the ACDSP_NO_MID knob says: the handle's state geometry (history length) must not
// depend on an A/B environment variable
int fir_mfma_plan_blocks_padded(int n_taps) {
  const int nb = (n_taps - 1 + 31) / 32 + 1;
  return (nb >= 10 && nb <= 32 && (nb & 1) == 0) ? nb + 1 : nb;
}
int fir_mfma_plan_blocks(int n_taps) {
  int nb = (n_taps - 1 + 31) / 32 + 1;
  static const bool no_mid = getenv("ACDSP_NO_MID") != nullptr;
  if (!no_mid && nb >= 10 && nb <= 32 && (nb & 1) == 0) { nb++; }
  return nb;
}

bool fir_mfma_build_fragments(const int64_t *c, int n_taps, FirMfmaPlan *plan, uint32_t *frag) {
  const int nb = fir_mfma_plan_blocks(n_taps);
  if (nb > kMaxNB) { return false; }
  std::vector<int8_t> chi(n_taps), clo(n_taps);
  int64_t sum = 0, sa = 0, sah = 0, sal = 0;
  for (int k = 0; k < n_taps; k++) {
    int64_t v = c[k];
    if (v < -32768 || v > 32767) { return false; }
    int64_t lo = ((v + 128) & 0xff) - 128;
    int64_t hi = (v - lo) / 256;
    if (hi < -128 || hi > 127) { return false; }  // c >= 32640: not representable as two signed bytes
    chi[k] = (int8_t)hi; clo[k] = (int8_t)lo;
    sum += v;
    sa += v < 0 ? -v : v; sah += hi < 0 ? -hi : hi; sal += lo < 0 ? -lo : lo;
  }
  plan->nb = nb;
  plan->sum_abs = sa; plan->sum_abs_hi = sah; plan->sum_abs_lo = sal;
  plan->corr = 128 * sum;
  plan->hi_mask = plan->lo_mask = 0;
  for (int pl = 0; pl < 2; pl++) {
    const int8_t *src = pl == 0 ? chi.data() : clo.data();
    for (int b = 0; b < nb; b++) {
      bool any = false;
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 31, h = lane >> 5;
        for (int dw = 0; dw < 4; dw++) {
          uint32_t word = 0;
          for (int bj = 0; bj < 4; bj++) {
            const int k = 16 * h + 4 * dw + bj;
            const int tap = i - k + 32 * (nb - 1 - b);
            int8_t val = (tap >= 0 && tap < n_taps) ? src[tap] : (int8_t)0;
            any = any || val != 0;
            word |= (uint32_t)(uint8_t)val << (8 * bj);
          }
          frag[(((size_t)pl * nb + b) * 64 + lane) * 4 + dw] = word;
        }
      }
      if (any) { (pl == 0 ? plan->hi_mask : plan->lo_mask) |= uint64_t(1) << b; }
    }
  }
  return true;
}

#endif  // !ACDSP_FIR_TU_MID

// Bytes of one staged [plane][half] array of nc 16-byte chunks.  The two halves of a plane are written by one
// ds_write_b64 (lanes alternate between them) and LDS stores see 32 banks: pad so that the arrays sit 16 banks apart
// (size = 64 mod 128), otherwise chunk c of both halves shares its banks (2-way conflict on every staging store:
// SQ_LDS_BANK_CONFLICT was 27 % of SQ_LDS_IDX_ACTIVE).
__host__ __device__ constexpr int staged_array_bytes(int nc) { return ((nc * 16 + 63) / 128) * 128 + 64; }

struct MfmaArgs {
  int64_t steps_per_wave;  // 1024-sample steps per wave
  int64_t n_steps;         // ceil(n / 1024)
  int64_t n8;              // n rounded up to a multiple of 8 (rows are readable that far)
  int32_t out_vec_ok;
  int32_t frag_per_channel;
  uint64_t hi_mask, lo_mask;  // bit b: K-block b of the hi / lo coefficient plane has a non-zero entry (any set)
  int32_t nb, hb0, hb1;       // big-NB kernel: K-blocks, and the range [hb0, hb1] of non-zero high-byte blocks
  int64_t step0;              // big-NB kernels: first 1024-sample step of this launch (split launches)
  const int64_t *corr;     // [n_sets] 128 * sum(c) per coefficient set
  // OUT_TYPEs of W < 16 bits in the 32-bit epilogue classes (round 4).  With d = 16 - W the epilogue shifts by rs - d instead of rs, packs
  // as for 16 bits (AC_SAT: saturating pack, AC_WRAP: truncating pack) and shifts the packed words right by d, arithmetically:
  // sat16(q') >> d == sat_W(q' >> d) and the sign bit of the truncated q' is bit W - 1 of q.  One v_pk_ashrrev_i16 per two outputs, in the
  // NAR instantiations of the pipelined body only; the edge chunks (fir_mfma_body) convert in registers: clamp to [nar_lo, nar_hi],
  // sign-extend the low 32 - nar_sh bits.  nar_on = 0: 16-bit OUT_TYPEs, nothing of this runs.
  int32_t nar_on, nar_d, nar_lo, nar_hi, nar_sh;
  // 16-bit OUT_TYPEs with a sign- / parity-dependent rounding mode or AC_SAT_SYM / AC_SAT_ZERO (round 5), also in the NAR instantiations
  // (nar_d = 0): the truncated quotient of the 32-bit epilogue plus the increment the dropped bits ask for (acdsp_dev.hpp: q_increment),
  // then a clamp to [gq_lo, gq_hi] (AC_SAT_SYM: +-(2^15 - 1)) or, gq_form = 2, zero outside the int16 range.  gq_on = 0: the constant modes
  // (AC_TRN / AC_RND / AC_RND_MIN_INF) into AC_WRAP / AC_SAT, whose rounding constant rides in ll.
  // gq_off / gq_c / gq_k: the mode's bit, increment and constant; gq_form: which copy of the loop (epi32_gq).
  int32_t gq_on, gq_off, gq_c, gq_k, gq_form, gq_lo, gq_hi;
  // unsigned 16-bit samples: 0x80808080 flips the top bit of every high byte as the planes are split (x - 32768 is a signed int16; the
  // host adds 32768 sum(c) to corr); 0 for signed samples.  One v_xor per four samples, in every instantiation.
  uint32_t hi_xor;
  // 4-byte containers in the wide class (W4 instantiations of the pipelined body): w4_sat = 1: AC_SAT bounds, 0: wrap to W_out bits in
  // 64 bits, 2: wrap in 32-bit arithmetic (2^8 mid + ll and, for rs > 16, hh + carry exact in int32: host-checked)
  int32_t w4_sat;
  int64_t w4_lo, w4_hi;
  int64_t *dbg;            // optional: per-wave {shader-clock ticks, 100 MHz real-time ticks} (ACDSP_DEBUG_CLOCK)
};

// 32-bit epilogue of the int16-output fast path.  V = 2^16 hh + 2^8 mid + ll is the exact dot product (ll
// already carries 128*sum(c) and the rounding constant); lo = 2^8 mid + ll fits int32 (host-checked), so
//   V >> rs = (hh << (16 - rs)) + (lo >> rs)            for rs <= 16   (2^16 hh is a multiple of 2^rs)
//   V >> rs = (hh + (lo >> 16)) >> (rs - 16)            for rs  > 16
// i.e. 3 (4) VALU ops per output; rs is wave-uniform.
// high-byte plane of unsigned 16-bit samples (MfmaArgs::hi_xor): in place and from an SGPR.  The nine-block kernels sit at the
// 256-register limit and the allocator's outcome there turns on details: written as `^` this spilled four VGPRs in the HS = 34 kernels of
// classes 1 / 2 and 117 in the dense class-3 one; as a movable asm only the latter (118); as a fixed one (volatile) the class-3 kernel drops
// to 2 spilled VGPRs (9 before the xor existed) but classes 1 / 2 spill 4 in their HS = 0 kernels -- so class 3 pins it, the others do not
// (tests/test_abi.py: test_no_kernel_uses_scratch is the judge of any other arrangement).
template <bool PINNED>
__device__ __forceinline__ unsigned hi_flip(unsigned v, unsigned m) {
  if constexpr (PINNED) { asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v) : "s"(m)); }
  else { asm("v_xor_b32 %0, %1, %0" : "+v"(v) : "s"(m)); }
  return v;
}
template <bool WIDE>   // WIDE: rs > 16
__device__ __forceinline__ void epi32_t(const v16i &hh, const v16i &mid, const v16i &ll, int rs, int (&o)[16]) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int lo = (int)(((unsigned)mid[r] << 8) + (unsigned)ll[r]);
    o[r] = WIDE ? (hh[r] + (lo >> 16)) >> (rs - 16) : (int)((unsigned)hh[r] << (16 - rs)) + (lo >> rs);
  }
}
// The same with the increment of a sign- or parity-dependent rounding mode (round 5, second form).  The dropped bits of V >> rs are the low
// rs bits of lo (rs <= 16: 2^16 hh is a multiple of 2^rs), or the low rs - 16 bits of hh + (lo >> 16) above the low half of lo; with
// rem = those bits, m = -1 where the floor quotient q0 is negative (gq_off = 31) or odd (gq_off = 0), every mode is
//   q = q0 + ((rem + (m & C) + K) >> rs)
// (acdsp_dev.hpp: q_increment, written as one carry): AC_TRN_ZERO C = 2^rs - 1, K = 0 on the sign; AC_RND_ZERO C = 1, K = half - 1 and
// AC_RND_INF C = -1, K = half on the sign; AC_RND_CONV C = 1, K = half - 1 and AC_RND_CONV_ODD C = -1, K = half on the parity; the
// constant modes (AC_TRN / AC_RND / AC_RND_MIN_INF: K rides in ll) come here only for AC_SAT_SYM / AC_SAT_ZERO, with C = K = 0.
// Six (seven) VALU per output on top of epi32_t's three (four); the first form tested mask bits per condition, ~24.
// For rs <= 16 the increment can go into lo BEFORE the shift -- (hh << (16 - rs)) + ((lo + (m & C) + K) >> rs), host-checked to stay inside
// int32 -- which saves the separate carry: the parity of q0 is bit rs of lo for rs < 16 (three VALU more than epi32_t), its sign needs q0
// first (five more, one less than the carry form: not worth a third copy of the loop in kernels that sit at the register limit -- with
// it the six- and nine-block NAR kernels spilled five VGPRs).  One uniform branch per step picks the form; the sign modes, rs = 16 on the
// parity and rs > 16 keep the carry form.
// fence between groups of four outputs of epi32_gq: holds the VALU (register pressure) but lets MFMA, SALU, VMEM and DS instructions cross
constexpr int kEpiFence = 0x0008 | 0x0004 | 0x0010 | 0x0080;
// FORM is a compile-time copy of MfmaArgs::gq_form: 0 = increment before the shift on the parity (rs < 16), 1 = carry form, 2 = carry form
// and AC_SAT_ZERO.  The pipelined loop must not branch: a uniform branch per emit splits its body into basic blocks, and the matrix
// products of a group no longer overlap the epilogue next to them (0.29 ms where AC_RND runs 0.21, VALU count almost equal) -- the kernel
// picks one of three copies of the loop instead (fir_mfma_kernel).  Forms 0 / 1 clamp to [gq_lo, gq_hi] (AC_SAT_SYM: +-(2^15 - 1); else the
// whole int32 range) with one v_med3.
template <bool WIDE, int FORM>
__device__ __forceinline__ void epi32_gq(const v16i &hh, const v16i &mid, const v16i &ll, int rs, const MfmaArgs &a, int (&o)[16]) {
  const unsigned C = (unsigned)a.gq_c, K = (unsigned)a.gq_k, off = (unsigned)a.gq_off, mask = (rs >= 32 ? 0u : (1u << rs)) - 1u;
  const int lo_b = a.gq_lo, hi_b = a.gq_hi;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int lo = (int)(((unsigned)mid[r] << 8) + (unsigned)ll[r]);
    int q;
    if constexpr (!WIDE && FORM == 0) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(lo, (unsigned)rs, 1u);
      q = (int)((unsigned)hh[r] << (16 - rs)) + ((int)((unsigned)lo + (m & C) + K) >> rs);
    } else {
      int q0;
      unsigned rem;
      if constexpr (!WIDE) {
        q0 = (int)((unsigned)hh[r] << (16 - rs)) + (lo >> rs);
        rem = (unsigned)lo & mask;
      } else {
        const int t = hh[r] + (lo >> 16);
        q0 = t >> (rs - 16);
        rem = __builtin_amdgcn_perm((unsigned)t, (unsigned)lo, 0x05040100u) & mask;   // t[15:0] : lo[15:0]
      }
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe(q0, off, 1u);
      q = q0 + (int)((rem + (m & C) + K) >> rs);
    }
    if constexpr (FORM == 2) { o[r] = (q < -32768 || q > 32767) ? 0 : q; }
    else { o[r] = q < lo_b ? lo_b : (q > hi_b ? hi_b : q); }
    if ((r & 3) == 3) { __builtin_amdgcn_sched_barrier(kEpiFence); }   // four outputs at a time: the kernels that carry this beside nine blocks of fragments have no registers to interleave sixteen
  }
}
// the same behind uniform branches, for the edge chunks (fir_mfma_body)
template <bool WIDE>
__device__ __forceinline__ void epi32_gq_rt(const v16i &hh, const v16i &mid, const v16i &ll, int rs, const MfmaArgs &a, int (&o)[16]) {
  if (a.gq_form == 0 && !WIDE) { epi32_gq<WIDE, 0>(hh, mid, ll, rs, a, o); }
  else if (a.gq_form == 2) { epi32_gq<WIDE, 2>(hh, mid, ll, rs, a, o); }
  else { epi32_gq<WIDE, 1>(hh, mid, ll, rs, a, o); }
}
__device__ __forceinline__ void epi32_narrow(const MfmaArgs &a, int (&o)[16]) {
  if (a.nar_on) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      int q = o[r] >> a.nar_d;
      q = q < a.nar_lo ? a.nar_lo : (q > a.nar_hi ? a.nar_hi : q);
      o[r] = (int)((unsigned)q << a.nar_sh) >> a.nar_sh;
    }
  }
}
// packed form of the last step (MfmaArgs): four dwords of int16 pairs >> d, arithmetically
__device__ __forceinline__ v4i pk16_ashr(const v4i &v, int d) {
  typedef short v2s_ __attribute__((ext_vector_type(2)));
  const v2s_ d2 = (v2s_){(short)d, (short)d};
  v4i r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // (the element goes through a scalar first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the
    // index -- clang 22 / ROCm 7.2 -- and every lane stored four copies of its first dword: round 5, found by disassembly; the round-4
    // parity tests of the narrow OUT_TYPEs never reached this pipelined body)
    const int e = v[i];
    r[i] = __builtin_bit_cast(int, __builtin_bit_cast(v2s_, e) >> d2);
  }
  return r;
}
template <bool GQ = true>   // GQ = false: the kernels of more than kMaxRegNB K-blocks, which the general-rounding class never reaches
__device__ __forceinline__ void epi32(const v16i &hh, const v16i &mid, const v16i &ll, int rs, const MfmaArgs &a, int (&o)[16]) {
  if (GQ && a.gq_on) {
    if (rs <= 16) { epi32_gq_rt<false>(hh, mid, ll, rs, a, o); } else { epi32_gq_rt<true>(hh, mid, ll, rs, a, o); }
    return;
  }
  if (rs <= 16) { epi32_t<false>(hh, mid, ll, rs, o); }   // one uniform branch per step, not one per output
  else { epi32_t<true>(hh, mid, ll, rs, o); }
  epi32_narrow(a, o);
}

// EPI 4 (round 4; the LDS-resident kernels only): int16 OUT containers when the 32-bit epilogue's bounds fail -- dense or
// high-gain sets of 258+ taps, where 2^8 mid + ll leaves int32 or the sum may wrap the accumulator.  V = 2^16 hh + 2^8 mid + ll + C
// in 64 bits, then the reference's two conversions branch-free: acc = wrap_ACC(V << lossless_shift); q = (acc + rnd) >> rs2 with
// rs2 = F_acc - F_out >= 1; AC_SAT clamps, AC_WRAP keeps the low W_out bits (host-derived constants in Epi64).  ~35 VALU per output
// instead of 3, still in the shadow of the 132 MFMAs of such a step; results leave through the same LDS tile and 16-byte stores as
// EPI 1 / 2 (the generic class, EPI 0, converts with uniform branches and stores element by element: 4.6 - 6.1 ms on the dense rows
// of profiles/r3_taps_sweep.txt).
struct Epi64 { int64_t corr, rnd, lo, hi; int ls, ka, rs, ko; };
__device__ __forceinline__ Epi64 make_epi64(const FirParams &p, int64_t corr) {
  Epi64 e;
  e.corr = corr; e.ls = p.lossless_shift; e.ka = 64 - p.acc.W; e.rs = p.acc.F - p.out.F;
  e.rnd = q_preload(p.out.Q, e.rs);   // (EPI 4 guarantees the range; the other classes never read it)
  if (p.out.O == ACDSP_SAT) { e.lo = p.out.lo; e.hi = p.out.hi; e.ko = 0; }
  else { e.lo = INT64_MIN; e.hi = INT64_MAX; e.ko = 64 - p.out.W; }
  return e;
}
__device__ __forceinline__ void epi64(const v16i &hh, const v16i &mid, const v16i &ll, const Epi64 &e, int (&o)[16]) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int64_t v = ((int64_t)hh[r] << 16) + ((int64_t)mid[r] << 8) + (int64_t)ll[r] + e.corr;
    int64_t acc = (int64_t)((uint64_t)v << e.ls);
    acc = (int64_t)((uint64_t)acc << e.ka) >> e.ka;                   // wrap to ACC_TYPE (signed)
    int64_t q = (acc + e.rnd) >> e.rs;                                // W_acc <= 62: the rounding add cannot leave int64
    q = q < e.lo ? e.lo : (q > e.hi ? e.hi : q);
    o[r] = (int)((int64_t)((uint64_t)q << e.ko) >> e.ko);
  }
}

// B-fragment read-ahead group size and target waves per SIMD of the register-resident kernel (tuning knobs:
// (3, 2) measured 1.117 ms on config 2, (1, 3) 1.068 ms but spills on dense coefficient sets -> 2.2 ms).
#ifndef ACDSP_GS
#define ACDSP_GS 3
#endif
#ifndef ACDSP_OCC
#define ACDSP_OCC 2
#endif
// non-temporal accesses (A/B knob): bit 0 = ring loads, bit 1 = int16 tile stores, bit 2 = wide tile stores of the pipelined body,
// bit 3 = stores of the double-wide 1023-tap kernel
#ifndef ACDSP_FIR_NT
#define ACDSP_FIR_NT 15
#endif
#ifndef ACDSP_FIR_PRIO
#define ACDSP_FIR_PRIO 0
#endif
// B-fragment read-ahead group of the 33-block shape (one wave per SIMD, 512 registers to spend): same-box A/B on config 4, two passes
// (round 5): 2: 2.067 / 2.068 ms, 3: 2.078, 4: 2.024 / 2.023, 6: 2.029 / 2.030, 8: 2.056 / 2.054.  The 12 + 12 band (nine high-plane blocks) keeps 2: at 4
// the allocator moves eight VGPRs through AGPRs
#ifndef ACDSP_GS_BIG
#define ACDSP_GS_BIG 4
#endif
constexpr int kGroupSize = ACDSP_GS, kOccupancy = ACDSP_OCC;

// EPI 0: any OUT_TYPE / ACC width through requant64.
// EPI 1: OUT container int16, Q in {TRN, RND}, O = WRAP, no accumulator wrap possible, right shift 1..31:
//        32-bit epilogue (epi32).   EPI 2: the same with O = SAT (v_cvt_pk_i16_i32 clamps and packs).
// EPI 3: OUT container int64, signed, Q in {TRN, RND}, O = WRAP, no accumulator wrap possible (the OUT = ACC row
//        of config 2): 64-bit shift-and-wrap epilogue, stored straight from registers; pipelined body only.
// HS:    compile-time band of K-blocks whose high-byte Toeplitz plane is non-zero: HS = lo + 16 hi skips the first `lo` and the
//        last `hi` blocks (0: none; the band of a linear-phase set is centred on tap (N-1)/2, which is not a block centre, so the
//        two sides differ: config 2's set needs blocks 3 .. 6 of 9).
// WAVES: 8 = ping-pong workgroup (see header), 1 = single-wave workgroup.
// FAST:  the chunk is interior: all loads/stores are full vectors, so the loop has no divergent branch
//        around VMEM and the compiler counts outstanding operations exactly (vmcnt(k), not vmcnt(0)).
template <int NB, int EPI, int HS, int WAVES, bool FAST>
__device__ __forceinline__ void fir_mfma_body(const FirParams &p, const v4i *__restrict__ frag, const MfmaArgs &a,
                                              unsigned char *lds_all) {
  constexpr int HB = NB - 1;          // halo chunks
  constexpr int NC = 32 + HB;         // chunks staged per step
  constexpr int NP = 4 * NC;          // 16-byte raw pieces per step
  constexpr int JN = (NP + 63) / 64;  // raw loads per lane per step
  constexpr int ARR = staged_array_bytes(NC);   // bytes of one [plane][half] array
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = (WAVES == 8) ? (wave >> 2) : 0;  // waves w and w+4 share a SIMD
  const int n_col = lane & 31, h = lane >> 5;
  int ch = blockIdx.y * WAVES + wave;
  if (ch >= p.n_ch) { ch = p.n_ch - 1; }  // surplus waves redo the last channel (identical stores): barriers stay uniform
  ch = __builtin_amdgcn_readfirstlane(ch);  // wave-uniform: row bases live in SGPRs
  const int set = a.frag_per_channel ? ch : 0;
  unsigned char *lds = lds_all + wave * (2 * 4 * ARR + 2048);
  unsigned char *obuf = lds + 2 * 4 * ARR;  // 2 KB output tile (FAST path)

  v4i Ah[NB], Al[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    Ah[b] = frag[((int64_t)set * 2 * NB + 0 * NB + b) * 64 + lane];
    Al[b] = frag[((int64_t)set * 2 * NB + 1 * NB + b) * 64 + lane];
  }

  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;  // hrow[t], t < 0
  const int64_t s0 = (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;
  const int nsteps = (int)(s1 - s0);

  // raw 16-byte pieces of this lane: piece l + 64 j covers samples T0 - 32 HB + 8 (l + 64 j) ...
  v4i R[JN];
  auto issue_loads = [&](int64_t T0) {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      // every lane loads (surplus lanes repeat the last piece): no divergent branch around VMEM
      const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
      int64_t t = T0 - 32 * HB + 8 * pc;
      const int16_t *src = (t < 0) ? hrow + t : xrow + ((t < a.n8) ? t : 0);  // beyond n: any valid address
      R[j] = *(const v4i *)src;
    }
  };
  // FAST, every fetch but the chunk's first: no history and no end of row in reach (32 HB <= 1024), so the
  // address is a scalar row base plus a loop-invariant 32-bit lane offset -- no per-load VALU.  A fetch past
  // the chunk's last step is redirected to that step (valid, unused).
  auto issue_loads_in = [&](int64_t T0) {
    const int64_t tl = (s1 - 1) * 1024;
    const char *sb = (const char *)(xrow + ((T0 < tl ? T0 : tl) - 32 * HB));
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
      R[j] = *(const v4i *)(sb + (unsigned)(16 * pc));
    }
  };
  // split into byte planes and stage: arrays [plane][half][chunk] of 16 bytes
  auto stage = [&](unsigned char *buf) {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = lane + 64 * j;
      if (JN * 64 == NP || pc < NP) {
        const int c = pc >> 2, hh_ = (pc >> 1) & 1, sub = pc & 1;
        unsigned hi0 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x07050301u), a.hi_xor);
        unsigned hi1 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x07050301u), a.hi_xor);
        unsigned lo0 = __builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x06040200u) ^ 0x80808080u;
        unsigned lo1 = __builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x06040200u) ^ 0x80808080u;
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        *(v2u *)(buf + (0 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){hi0, hi1};
        *(v2u *)(buf + (1 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){lo0, lo1};
      }
    }
  };

  // epilogue constants
  const int rs = p.in.F + p.cf.F - p.out.F;
  const int64_t corr = a.corr[set];
  // EPI 1/2: C = 128*sum(c) + rounding constant rides in as the initial value of the low-plane accumulator;
  // epi32() then needs 3 VALU ops per output and v_cvt_pk_i16_i32 packs (and clamps, for AC_SAT).
  const int64_t corr_t = corr + (EPI != 0 ? q_preload(p.out.Q, rs) : 0);
  const int c_ll = (EPI != 0) ? (int)corr_t : 0;   // preloaded into the low-plane accumulator (int32-safe, host-checked)
  const v16i ll_init = {c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll};
  int16_t *yrow = (int16_t *)p.y + (int64_t)ch * p.out_stride + 32 * n_col + 4 * h;  // EPI 1/2

  // The Toeplitz fragments must have landed before the loop: otherwise the compiler keeps
  // "s_waitcnt vmcnt(k)" for them inside the loop body (simm16: vmcnt 0, expcnt/lgkmcnt untouched).
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);

  // Fragment reads (conflict-free ds_read_b128) run one group of GS K-blocks ahead of the MFMAs that
  // consume them; sched_barrier(0) pins the "reads of group g+1, then MFMAs of group g" order, which
  // the scheduler would otherwise re-serialise into read-wait-MFMA per block.  Group 0 of a step is
  // read at the end of the previous O phase, so its latency hides behind the barrier.
  constexpr int GS = kGroupSize, NG = (NB + GS - 1) / GS;
  v4i Bh[2][GS], Bl[2][GS];
  auto read_group = [&](const unsigned char *buf, int g, v4i (&dh)[GS], v4i (&dl)[GS]) {
    const unsigned char *fh = buf + (0 * 2 + h) * ARR + n_col * 16;
    const unsigned char *fl = buf + (1 * 2 + h) * ARR + n_col * 16;
#pragma unroll
    for (int i = 0; i < GS; i++) {
      const int b = g * GS + i;
      if (b < NB) {
        dh[i] = *(const v4i *)(fh + 16 * b);
        dl[i] = *(const v4i *)(fl + 16 * b);
      }
    }
  };

  // software pipeline: loads run two steps ahead of the MFMAs, staging one step ahead
  issue_loads(s0 * 1024);
  stage(lds);
  if (FAST) { issue_loads_in((s0 + 1) * 1024); }
  else if (nsteps > 1) { issue_loads((s0 + 1) * 1024); }
  read_group(lds, 0, Bh[0], Bl[0]);
  if (WAVES == 8 && grp == 1) { __builtin_amdgcn_s_barrier(); }  // second half starts one phase later

  for (int s = 0; s < nsteps; s++) {
    const int64_t T0 = (s0 + s) * 1024;
    const unsigned char *buf = lds + (s & 1) * (4 * ARR);

#ifdef ACDSP_X_PHASES
    const uint64_t tp0 = __builtin_readcyclecounter();
#endif
    // ---------------- phase M: MFMA run (four independent accumulators: every one is reused only
    // every fourth MFMA, so a single wave keeps the matrix pipe at its 32-cycle issue rate) ----------------
    v16i hh = {0}, mid = {0}, ll = ll_init;
#pragma unroll
    for (int g = 0; g < NG; g++) {
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < NG) { read_group(buf, g + 1, Bh[(g + 1) & 1], Bl[(g + 1) & 1]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GS; i++) {
        const int b = g * GS + i;
        if (b < NB) {
          // a Toeplitz block whose high-byte plane is all zero contributes nothing to hh / mid
          if (HS == 0 || (b >= (HS & 15) && b <= NB - 1 - (HS >> 4))) {
            hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bh[g & 1][i], hh, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bl[g & 1][i], mid, 0, 0, 0);
          }
          ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bl[g & 1][i], ll, 0, 0, 0);
          mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bh[g & 1][i], mid, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef ACDSP_X_PHASES
    const uint64_t tp1 = __builtin_readcyclecounter();
#endif
    if (WAVES == 8) { __builtin_amdgcn_s_barrier(); }
#ifdef ACDSP_X_PHASES
    const uint64_t tp2 = __builtin_readcyclecounter();
#endif

    // ---------------- phase O: epilogue, stores, staging of the next step, prefetch ----------------
    // D layout: lane (n_col, h), register r: sample T0 + 32 n_col + (r&3) + 8 (r>>2) + 4 h
    int o16[16];
    if (EPI != 0) { epi32(hh, mid, ll, rs - a.nar_d, a, o16); }
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int64_t t0 = T0 + 32 * n_col + 8 * g + 4 * h;
      if (EPI != 0) {
        const int *o = o16 + 4 * g;
        v4s pk;
        if (EPI == 2) {  // OUT_TYPE is a signed 16-bit AC_SAT type: clamp and pack in one instruction
          typedef short v2s __attribute__((ext_vector_type(2)));
          const v2s p0 = __builtin_amdgcn_cvt_pk_i16(o[0], o[1]), p1 = __builtin_amdgcn_cvt_pk_i16(o[2], o[3]);
          pk = (v4s){p0.x, p0.y, p1.x, p1.y};
        } else {
          pk = (v4s){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
        }
        int16_t *dst = yrow + T0 + 8 * g;
        if (FAST) {
          // stage the 8-byte piece for the row-contiguous write-out below: pair P = 4 n + g holds samples
          // 32 n + 8 g .. +7; its slot is rotated by P >> 4 so that both the ds_write_b64 here and the
          // ds_read_b128 there are bank-conflict free
          const int P = 4 * n_col + g;
          *(v4s *)(obuf + (((P & ~15) | ((P + (P >> 4)) & 15)) * 16 + 8 * h)) = pk;
        } else if (a.out_vec_ok && t0 + 4 <= p.n) {
          *(v4s *)dst = pk;
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            if (t0 + rr < p.n) { dst[rr] = pk[rr]; }   // pk: already clamped for AC_SAT
          }
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int r = 4 * g + rr;
          int64_t v = ((int64_t)hh[r] << 16) + ((int64_t)mid[r] << 8) + (int64_t)ll[r] + corr;
          int64_t acc = wrap64((int64_t)((uint64_t)v << p.lossless_shift), p.acc.W, p.acc.S);
          int64_t y = requant64(acc, p.acc.F, p.out);
          if (t0 + rr < p.n) { store_raw(p.y, (int64_t)ch * p.out_stride + t0 + rr, p.out_eb, y); }
        }
      }
    }
    if (FAST && EPI != 0) {
      // 1024 outputs = 2 KB contiguous: two fully coalesced 16-byte-per-lane stores (8 whole 128-byte
      // lines each) instead of four 8-byte scatters that L2 has to merge
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int P = 64 * half + lane;
        const v4i val = *(const v4i *)(obuf + ((P & ~15) | ((P + (P >> 4)) & 15)) * 16);
        *(v4i *)((int16_t *)p.y + (int64_t)ch * p.out_stride + T0 + 512 * half + 8 * lane) = val;
      }
    }
    if (s + 1 < nsteps) {
      unsigned char *nbuf = lds + ((s + 1) & 1) * (4 * ARR);
      stage(nbuf);                                                      // consumes the loads of step s+1
      if (FAST) { issue_loads_in(T0 + 2048); }                          // past the chunk: redirected, harmless
      else if (s + 2 < nsteps) { issue_loads(T0 + 2048); }
      read_group(nbuf, 0, Bh[0], Bl[0]);
    }
#ifdef ACDSP_X_PHASES
    const uint64_t tp3 = __builtin_readcyclecounter();
#endif
    if (WAVES == 8 && (grp == 0 || s + 1 < nsteps)) { __builtin_amdgcn_s_barrier(); }
#ifdef ACDSP_X_PHASES
    if (a.dbg && lane == 0) {
      const int64_t w = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave;
      int64_t *d = a.dbg + 2 * (int64_t)gridDim.x * gridDim.y * WAVES + 4 * w;
      d[0] += (int64_t)(tp1 - tp0); d[1] += (int64_t)(tp2 - tp1); d[2] += (int64_t)(tp3 - tp2);
      d[3] += (int64_t)(__builtin_readcyclecounter() - tp3);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------
// Software-pipelined body for interior chunks of single-wave workgroups (the steady state of every long
// run).  The plain body above alternates an MFMA run (M) with ~120 dependent VALU/LDS/VMEM instructions
// (O): measured 1290 + 1220 shader cycles per step, i.e. the matrix pipe idles half the time even with two
// waves per SIMD.  Here step s's MFMAs run with everything else of the neighbouring steps interleaved in
// the same instruction stream -- the epilogue and write-out of step s-1 (from a second accumulator set),
// the byte-plane staging of step s+1 and the global loads of step s+2 -- so the wave always has MFMAs to
// issue and the other work hides in their shadow (tools/mfma_probe.hip: 39 cycles/MFMA with the epilogue
// interleaved vs 36 bare).  The loop is unrolled by two so the accumulator sets and the B-fragment
// double buffer swap roles by renaming; it contains no branch.
template <int NB, int EPI, int HS, int NAR = 0, bool W4 = false, int GQF = 1>   // NAR: 1 = OUT_TYPEs of fewer than 16 bits, 2 = general rounding / overflow modes of a 16-bit OUT_TYPE (GQF: epi32_gq's FORM)
__device__ __forceinline__ void fir_mfma_pipe_body(const FirParams &p, const v4i *__restrict__ frag, const MfmaArgs &a,
                                                   unsigned char *lds) {
  static_assert(EPI >= 1 && EPI <= 3, "fast epilogue classes only");
  constexpr int HB = NB - 1, NC = 32 + HB, NP = 4 * NC, JN = (NP + 63) / 64;
  // The staged byte planes live in a RING of 128 chunk slots (four steps of 32) per [plane][half] array, plus HB mirror
  // slots [128, 128 + HB) that repeat slots [0, HB): the window of a step with s % 4 == PAR is the linear slot range
  // [32 PAR, 32 PAR + 32 + HB), so its first HB chunks are the tail the previous step staged -- a step loads and stages only
  // its own 2 KB of new samples (two 16-byte loads per lane) instead of the whole 2.5 KB window with its 25 % halo.
  // The wide-output class (EPI 3) keeps two separate windows per step pair: its loop, unrolled by four, ran 2 - 8 % slower
  // (same-box A/B, profiles/r2_ab_ring.txt) -- that row is bound by its 8-byte stores, not by the input side.
  constexpr bool RINGED = EPI != 3;
  constexpr int RING = 128 + HB, ARR = RINGED ? staged_array_bytes(RING) : staged_array_bytes(NC);
  // two accumulator sets + all Toeplitz fragments leave room for GS = 2 only when some high-byte blocks are skipped
  // (round 4: also NB = 7 dense and the wide-output class with at most two blocks skipped per side -- those spilled 1 - 9 VGPRs at GS = 2)
  constexpr int GS = (NB == 33 && HS == 14 + 16 * 14) ? ACDSP_GS_BIG : (NB > kMaxRegNB ? 2 : (((HS == 0 && NB >= 7) || (EPI == 3 && NB >= 9 && (HS == 0 || HS == 2 + 16 * 2))) ? 1 : 2)), NG = (NB + GS - 1) / GS;
  const int lane = threadIdx.x & 63;
  const int n_col = lane & 31, h = lane >> 5;
  int ch = blockIdx.y;
  if (ch >= p.n_ch) { ch = p.n_ch - 1; }
  ch = __builtin_amdgcn_readfirstlane(ch);
  const int set = a.frag_per_channel ? ch : 0;
  unsigned char *obuf = lds + (RINGED ? 4 : 2 * 4) * ARR;
  unsigned char *dummy = obuf + (EPI == 3 ? 8192 : 2048);   // 1 KB sink for the surplus lanes of stage()

  // (Round 3 also built a form whose Toeplitz rows were permuted so that a lane held 16 CONSECUTIVE outputs and stored 32 contiguous
  // bytes straight from registers, no LDS tile: 0.907 -> 1.23 ms on config 2, profiles/r3_ab_direct_nt.txt -- each store instruction then
  // writes every other 16-byte piece of a 2 KB run.  Rejected; the code was removed in round 4.)
  v4i Ah[NB], Al[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    Ah[b] = frag[((int64_t)set * 2 * NB + 0 * NB + b) * 64 + lane];
    Al[b] = frag[((int64_t)set * 2 * NB + 1 * NB + b) * 64 + lane];
  }
  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;
  int16_t *yout = (int16_t *)p.y + (int64_t)ch * p.out_stride;
  const int64_t s0 = (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;
  const int nsteps = (int)(s1 - s0);

  v4i R[JN];
  auto issue_loads_first = [&](int64_t T0) {   // may reach into the history rows
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
      int64_t t = T0 - 32 * HB + 8 * pc;
      const int16_t *src = (t < 0) ? hrow + t : xrow + ((t < a.n8) ? t : 0);
      R[j] = *(const v4i *)src;
    }
  };
  v4i Q[2];
  auto issue_loads_new = [&](int64_t T0) {     // the 1024 new samples of a later step; past the chunk: the last step's (never used)
    const int64_t tl = (s1 - 1) * 1024;
    if constexpr (RINGED) {
      const char *sb = (const char *)(xrow + (T0 < tl ? T0 : tl));
#pragma unroll
      for (int j = 0; j < 2; j++) {
#if ACDSP_FIR_NT & 1   // the ring reads every input byte once: non-temporal loads and stores -2.4 % same box (profiles/r2_ab_nt.txt)
        Q[j] = __builtin_nontemporal_load((const v4i *)(sb + (unsigned)(16 * (lane + 64 * j))));
#else
        Q[j] = *(const v4i *)(sb + (unsigned)(16 * (lane + 64 * j)));
#endif
      }
    } else {                                   // the whole window of that step (see fir_mfma_body)
      const char *sb = (const char *)(xrow + ((T0 < tl ? T0 : tl) - 32 * HB));
#pragma unroll
      for (int j = 0; j < JN; j++) {
        const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
        R[j] = *(const v4i *)(sb + (unsigned)(16 * pc));
      }
    }
  };
  typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
  auto stage_ring = [&](auto par_c) {          // Q -> slots [32 (PAR + 1) + HB, + 32) mod 128 of the step after a step of parity PAR
    constexpr int PAR = decltype(par_c)::value;
    constexpr int base = (PAR == 3) ? HB : 32 * (PAR + 1) + HB;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int pc = lane + 64 * j;
      const int c = pc >> 2, hh_ = (pc >> 1) & 1, sub = pc & 1;
      const unsigned hi0 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)Q[j].y, (unsigned)Q[j].x, 0x07050301u), a.hi_xor);
      const unsigned hi1 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)Q[j].w, (unsigned)Q[j].z, 0x07050301u), a.hi_xor);
      const unsigned lo0 = __builtin_amdgcn_perm((unsigned)Q[j].y, (unsigned)Q[j].x, 0x06040200u) ^ 0x80808080u;
      const unsigned lo1 = __builtin_amdgcn_perm((unsigned)Q[j].w, (unsigned)Q[j].z, 0x06040200u) ^ 0x80808080u;
      unsigned char *dh = lds + (0 * 2 + hh_) * ARR + (base + c) * 16 + sub * 8;
      unsigned char *dl = lds + (1 * 2 + hh_) * ARR + (base + c) * 16 + sub * 8;
      *(v2u_ *)dh = (v2u_){hi0, hi1};
      *(v2u_ *)dl = (v2u_){lo0, lo1};
      if (PAR == 2 && HB > 0 && 16 * (j + 1) > 32 - HB) {   // slots [128, 128 + HB) also go to [0, HB): the window of parity 0 starts there
        const bool m = c >= 32 - HB;                        // (chunks 32 - HB .. 31 of the step: both loads when HB > 16, i.e. NB = 33)
        *(v2u_ *)(m ? dh - 128 * 16 : dummy + lane * 8) = (v2u_){hi0, hi1};
        *(v2u_ *)(m ? dl - 128 * 16 : dummy + 512 + lane * 8) = (v2u_){lo0, lo1};
      }
    }
  };
  auto stage = [&](unsigned char *buf) {       // prologue: the whole window of the chunk's first step -> slots [0, NC)
#pragma unroll
    for (int j = 0; j < JN; j++) {
      // surplus lanes (last j only) store their copy of the last piece into a private dummy slot: no exec-mask branch
      // in the loop, and no 32 lanes hammering one address (same-address ds_writes serialise: measured as 27 % of
      // the LDS cycles in SQ_LDS_BANK_CONFLICT)
      const int pc = lane + 64 * j;
      const bool live = (64 * (j + 1) <= NP) || pc < NP;
      const int c = pc >> 2, hh_ = (pc >> 1) & 1, sub = pc & 1;
      unsigned hi0 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x07050301u), a.hi_xor);
      unsigned hi1 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x07050301u), a.hi_xor);
      unsigned lo0 = __builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x06040200u) ^ 0x80808080u;
      unsigned lo1 = __builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x06040200u) ^ 0x80808080u;
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      unsigned char *dh = live ? buf + (0 * 2 + hh_) * ARR + c * 16 + sub * 8 : dummy + lane * 8;
      unsigned char *dl = live ? buf + (1 * 2 + hh_) * ARR + c * 16 + sub * 8 : dummy + 512 + lane * 8;
      *(v2u *)dh = (v2u){hi0, hi1};
      *(v2u *)dl = (v2u){lo0, lo1};
    }
  };

  auto stage_new = [&](auto par_c) {
    if constexpr (RINGED) { stage_ring(par_c); }
    else { stage(lds + ((decltype(par_c)::value & 1) ^ 1) * (4 * ARR)); }
  };

  const int rs = p.in.F + p.cf.F - p.out.F;
  const int c_ll = (int)(a.corr[set] + q_preload(p.out.Q, rs));
  const v16i ll_init = {c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll};

  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // Toeplitz fragments landed (keeps their vmcnt out of the loop)
  __builtin_amdgcn_sched_barrier(0);

  // B fragments: groups of GS K-blocks, double buffered; the group sequence runs on across steps, so the
  // buffer of group g of a step with parity PAR is (PAR * NG + g) & 1.
  v4i Bh[2][GS], Bl[2][GS];
  auto read_group = [&](int woff, int g, v4i (&dh)[GS], v4i (&dl)[GS]) {   // woff: byte offset of the step's window
    const unsigned char *fh = lds + woff + (0 * 2 + h) * ARR + n_col * 16;
    const unsigned char *fl = lds + woff + (1 * 2 + h) * ARR + n_col * 16;
#pragma unroll
    for (int i = 0; i < GS; i++) {
      const int b = g * GS + i;
      if (b < NB) {
        dh[i] = *(const v4i *)(fh + 16 * b);
        dl[i] = *(const v4i *)(fl + 16 * b);
      }
    }
  };
  // epilogue of a finished step: 16 outputs per lane -> packed int16 -> swizzled 2 KB LDS tile (fir_mfma_body)
  int64_t *yout64 = (int64_t *)p.y + (int64_t)ch * p.out_stride;
  const int e3_sr = rs > 0 ? rs : 0, e3_wl = 64 - p.out.W, e3_sl = (rs < 0 ? -rs : 0) + e3_wl;
  auto emit = [&](auto wide_c, int64_t T0, const v16i &hh, const v16i &mid, const v16i &ll, int prsel = 2) {
    if constexpr (EPI == 3 && W4) {
      // 4-byte containers (OUT_TYPEs of 17 .. 32 bits; round 4): the same 64-bit value, wrapped (AC_WRAP) or clamped (AC_SAT: MfmaArgs::w4_*)
      // to W bits, as int32 -- a 4 KB tile of 16-byte slots, slot = 8 n + 2 g + h, XOR-swizzled with n >> 1 so that the ds_write_b128 here
      // (16 lanes = columns n .. n + 15 of one (g, h): two 128-byte rows per column pair, eight distinct 16-byte bank groups in each) and
      // the linear ds_read_b128 of flush() (16 consecutive slots = two whole column rows) are bank-conflict free.
#pragma unroll
      for (int g = 0; g < 4; g++) {
        int o[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int r = 4 * g + rr;
          if (a.w4_sat == 2) {
            // AC_WRAP needs the result mod 2^32 only: lo = 2^8 mid + ll is exact in int32 (host-checked), the rest may wrap
            const int lo = (int)(((unsigned)mid[r] << 8) + (unsigned)ll[r]);
            const unsigned q32 = rs <= 16 ? ((unsigned)hh[r] << (16 - rs)) + (unsigned)(lo >> rs) : (unsigned)((hh[r] + (lo >> 16)) >> (rs - 16));
            o[rr] = (int)(q32 << (32 - p.out.W)) >> (32 - p.out.W);
          } else {
            const int64_t V = ((int64_t)hh[r] << 16) + ((int64_t)mid[r] << 8) + (int64_t)ll[r];
            int64_t q = V >> e3_sr;
            if (a.w4_sat) { q = q < a.w4_lo ? a.w4_lo : (q > a.w4_hi ? a.w4_hi : q); }
            else { q = (int64_t)((uint64_t)q << e3_sl) >> e3_wl; }
            o[rr] = (int)q;
          }
        }
        const int slot = 8 * n_col + ((2 * g + h) ^ ((n_col >> 1) & 7));
        *(v4i *)(obuf + slot * 16) = (v4i){o[0], o[1], o[2], o[3]};
      }
      return;
    }
    if constexpr (EPI == 3) {
      // y = wrap_W((V + rnd) >> rs) (or V << -rs), V = 2^16 hh + 2^8 mid + ll.  The 1024 outputs of the step form an
      // 8 KB tile of 16-byte slots (slot = 16 n + 4 g + 2 h + half), XOR-swizzled with the column so that both the
      // ds_write_b128 here (8 consecutive columns per LDS cycle) and the row-contiguous ds_read_b128 of flush() are
      // bank-conflict free; the write-out is then whole 128-byte lines (32-byte pieces straight from registers ran
      // at 2.9 TB/s).
      typedef long v2l __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int g = 0; g < 4; g++) {
        int64_t v[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int r = 4 * g + rr;
          const int64_t V = ((int64_t)hh[r] << 16) + ((int64_t)mid[r] << 8) + (int64_t)ll[r];
          v[rr] = (int64_t)((uint64_t)(V >> e3_sr) << e3_sl) >> e3_wl;
        }
        const int slot = 16 * n_col + ((4 * g + 2 * h) ^ (n_col & 15));
        *(v2l *)(obuf + slot * 16) = (v2l){v[0], v[1]};
        *(v2l *)(obuf + (slot ^ 1) * 16) = (v2l){v[2], v[3]};
      }
      return;
    }
    int o[16];
    if constexpr (NAR == 2 && GQF < 0) { epi32_gq_rt<decltype(wide_c)::value>(hh, mid, ll, rs, a, o); }
    else if constexpr (NAR == 2) { epi32_gq<decltype(wide_c)::value, GQF>(hh, mid, ll, rs, a, o); }   // (general rounding modes: 16-bit OUT_TYPEs only)
    else { epi32_t<decltype(wide_c)::value>(hh, mid, ll, NAR == 1 ? rs - a.nar_d : rs, o); }
    // A lane holds rows 8 g + 4 h .. + 3 of column n: 8 bytes per g, and the 16 lanes of a ds_write_b64 group share h, so
    // they can reach only half of the 32 banks (2-way conflict on every store: the 16.6 % SQ_LDS_BANK_CONFLICT of round 1).
    // v_permlane32_swap trades g-odd of the h = 0 lanes for g-even of the h = 1 lanes: every lane then owns 16 contiguous
    // bytes (slot P = 4 n + 2 pr + h), written as two ds_write_b128; slot ^ ((n >> 1) & 3) spreads the 8 lanes of a store
    // group over the 8 slots of a 128-byte bank row and keeps the aligned 4-slot sets flush() reads conflict-free.
    unsigned d[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (EPI == 2) {
        d[g][0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(o[4 * g], o[4 * g + 1]));
        d[g][1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(o[4 * g + 2], o[4 * g + 3]));
      } else {
        d[g][0] = __builtin_amdgcn_perm((unsigned)o[4 * g + 1], (unsigned)o[4 * g], 0x05040100u);
        d[g][1] = __builtin_amdgcn_perm((unsigned)o[4 * g + 3], (unsigned)o[4 * g + 2], 0x05040100u);
      }
    }
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      if (prsel != 2 && prsel != pr) { continue; }
      const auto a0 = __builtin_amdgcn_permlane32_swap(d[2 * pr][0], d[2 * pr + 1][0], false, false);
      const auto a1 = __builtin_amdgcn_permlane32_swap(d[2 * pr][1], d[2 * pr + 1][1], false, false);
      const int P = 4 * n_col + 2 * pr + h;
      *(v4i *)(obuf + (P ^ ((n_col >> 1) & 3)) * 16) = (v4i){(int)a0[0], (int)a1[0], (int)a0[1], (int)a1[1]};
    }
  };
  // ... and its row-contiguous write-out: two coalesced 16-byte-per-lane stores
  auto flush = [&](int64_t T0) {
    if constexpr (EPI == 3 && W4) {
      char *yout32 = (char *)((int32_t *)p.y + (int64_t)ch * p.out_stride + T0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int P = 64 * k + lane;
        const v4i val = *(const v4i *)(obuf + ((P & ~7) | ((P & 7) ^ ((P >> 4) & 7))) * 16);
#if ACDSP_FIR_NT & 4
        __builtin_nontemporal_store(val, (v4i *)(yout32 + (unsigned)(16 * P)));
#else
        *(v4i *)(yout32 + (unsigned)(16 * P)) = val;
#endif
      }
      return;
    }
    if constexpr (EPI == 3) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int P = 64 * k + lane;
        const v4i val = *(const v4i *)(obuf + (P ^ ((P >> 4) & 15)) * 16);
#if ACDSP_FIR_NT & 4
        __builtin_nontemporal_store(val, (v4i *)((char *)(yout64 + T0) + (unsigned)(16 * P)));
#else
        *(v4i *)((char *)(yout64 + T0) + (unsigned)(16 * P)) = val;
#endif
      }
      return;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int P = 64 * half + lane;
      v4i val = *(const v4i *)(obuf + (P ^ ((P >> 3) & 3)) * 16);
      if constexpr (NAR == 1) { val = pk16_ashr(val, a.nar_d); }   // OUT_TYPEs of fewer than 16 bits (MfmaArgs::nar_*)
#if ACDSP_FIR_NT & 2
      __builtin_nontemporal_store(val, (v4i *)(yout + T0 + 512 * half + 8 * lane));
#else
      *(v4i *)(yout + T0 + 512 * half + 8 * lane) = val;
#endif
    }
  };

  // One step.  PAR: step parity (selects buffers by renaming); PREV: there is a finished step in (ph, pm, pl).
  auto run_step = [&](auto wide_c, auto par_c, auto prev_c, int s, v16i &hh, v16i &mid, v16i &ll, const v16i &ph, const v16i &pm,
                      const v16i &pl) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool PREV = decltype(prev_c)::value;
    const int64_t T0 = (s0 + s) * 1024;
    // byte offsets of this step's and the next step's window: ring slots 32 PAR, or one of the two separate windows
    constexpr int buf = RINGED ? 512 * PAR : (PAR & 1) * (4 * ARR), nbuf = RINGED ? 512 * ((PAR + 1) & 3) : ((PAR & 1) ^ 1) * (4 * ARR);
    hh = (v16i){0}; mid = (v16i){0}; ll = ll_init;
    // side work, spread over the first groups: S = stage step s+1, L = fetch step s+2, E1 = epilogue of step
    // s-1 into the LDS tile, E2 = its write-out
    // loads first (longest latency), the epilogue under the widest MFMA groups: same-box A/B of seven placements in
    // profiles/r2_ab_place.txt (round-2 start: L 1, E1 1, E2 2; this one -1.3 % on config 2, -0.6 % dense, -0.5 % wide).
    // With five or more groups the int16 epilogue is split over two of them (halves of the tile), write-out one group
    // later: another -0.9 % (profiles/r2_ab_place.txt, last section).
    constexpr bool kSplitEmit = NG > 4 && EPI != 3;
    constexpr int gS = 0, gL = 0, gE1 = (NG > 2) ? 2 : NG - 1, gE2 = kSplitEmit ? 4 : ((NG > 3) ? 3 : NG - 1);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      __builtin_amdgcn_sched_barrier(0);
      constexpr int dummy = 0; (void)dummy;
      const int cb = (PAR * NG + g) & 1, nb_ = cb ^ 1;
      // ACDSP_ABL_*: timing-only ablation builds (wrong results) behind the table in profiles/r2_fir255_clock.txt (d)
#ifndef ACDSP_ABL_STAGE
      if (NG == 1 && g == gS) { stage_new(par_c); }
#endif
#ifdef ACDSP_ABL_BREAD
      if (g == 0) {
#endif
      if (g + 1 < NG) { read_group(buf, g + 1, Bh[nb_], Bl[nb_]); }
      else { read_group(nbuf, 0, Bh[nb_], Bl[nb_]); }          // first group of the next step (staged in group 0)
#ifdef ACDSP_ABL_BREAD
      }
#endif
#ifndef ACDSP_ABL_STAGE
      if (NG > 1 && g == gS) { stage_new(par_c); }
#endif
#ifndef ACDSP_ABL_LOAD
      if (g == gL) { issue_loads_new(T0 + 2048); }
#endif
#ifndef ACDSP_ABL_EMIT
      if (PREV && kSplitEmit) {
        if (g == gE1) { emit(wide_c, T0 - 1024, ph, pm, pl, 0); }
        if (g == gE1 + 1) { emit(wide_c, T0 - 1024, ph, pm, pl, 1); }
      } else if (PREV && g == gE1) { emit(wide_c, T0 - 1024, ph, pm, pl); }
#else
      if (PREV && g == gE1) { asm volatile("" :: "v"(ph[0]), "v"(pm[0]), "v"(pl[0])); }   // keeps the MFMAs of the step alive
#endif
#ifndef ACDSP_ABL_FLUSH
      if (PREV && g == gE2) { flush(T0 - 1024); }
#endif
#if ACDSP_FIR_PRIO   // A/B knob: raise the wave's issue priority over its MFMA runs (round-2 review item 1c)
      __builtin_amdgcn_s_setprio(ACDSP_FIR_PRIO);
#endif
#pragma unroll
      for (int i = 0; i < GS; i++) {
        const int b = g * GS + i;
        if (b < NB) {
#if ACDSP_FIR_BORDER
          // (round 6 A/B) the sample operand stays for two consecutive products of a band block: Bh, Bh, Bl, Bl instead of Bh, Bl, Bl, Bh
          if (HS == 0 || (b >= (HS & 15) && b <= NB - 1 - (HS >> 4))) {
            hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bh[cb][i], hh, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bh[cb][i], mid, 0, 0, 0);
            ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bl[cb][i], ll, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bl[cb][i], mid, 0, 0, 0);
          } else {
            ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bl[cb][i], ll, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bh[cb][i], mid, 0, 0, 0);
          }
#else
          if (HS == 0 || (b >= (HS & 15) && b <= NB - 1 - (HS >> 4))) {
            hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bh[cb][i], hh, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah[b], Bl[cb][i], mid, 0, 0, 0);
          }
          ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bl[cb][i], ll, 0, 0, 0);
          mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al[b], Bh[cb][i], mid, 0, 0, 0);
#endif
        }
      }
#if ACDSP_FIR_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
  typedef integral_constant<int, 0> P0; typedef integral_constant<int, 1> P1;
  typedef integral_constant<int, 2> P2; typedef integral_constant<int, 3> P3;
  typedef integral_constant<bool, true> WithPrev; typedef integral_constant<bool, false> NoPrev;

  auto go = [&](auto wide_c) {   // the loop exists once per epilogue shift class: no branch inside it
    // prologue: step s0 staged, step s0+1 in flight, first B group read.  Inside the arm: hoisted above the branch on the shift class,
    // the first B fragments were live across the OTHER arm's loop and spilled around it (4 VGPRs of scratch in eight instantiations).
    issue_loads_first(s0 * 1024);
    stage(lds);
    issue_loads_new((s0 + 1) * 1024);
    read_group(0, 0, Bh[0], Bl[0]);
    v16i hA, mA, lA, hB, mB, lB;
    run_step(wide_c, P0(), NoPrev(), 0, hA, mA, lA, hA, mA, lA);
    int s = 1;
    auto last = [&](int sl, const v16i &h_, const v16i &m_, const v16i &l_) {
      emit(wide_c, (s0 + sl) * 1024, h_, m_, l_);
      flush((s0 + sl) * 1024);
    };
    if constexpr (!RINGED) {
      for (; s + 1 < nsteps; s += 2) {
        run_step(wide_c, P1(), WithPrev(), s, hB, mB, lB, hA, mA, lA);
        run_step(wide_c, P0(), WithPrev(), s + 1, hA, mA, lA, hB, mB, lB);
      }
      if (s < nsteps) {
        run_step(wide_c, P1(), WithPrev(), s, hB, mB, lB, hA, mA, lA);
        last(s, hB, mB, lB);
      } else {
        last(s - 1, hA, mA, lA);
      }
    } else {
      for (; s + 3 < nsteps; s += 4) {      // ring parity = s % 4, accumulator set = s % 2
        run_step(wide_c, P1(), WithPrev(), s, hB, mB, lB, hA, mA, lA);
        run_step(wide_c, P2(), WithPrev(), s + 1, hA, mA, lA, hB, mB, lB);
        run_step(wide_c, P3(), WithPrev(), s + 2, hB, mB, lB, hA, mA, lA);
        run_step(wide_c, P0(), WithPrev(), s + 3, hA, mA, lA, hB, mB, lB);
      }
      if (s < nsteps) {                     // up to three more steps; each arm ends with the write-out of its own last step
        run_step(wide_c, P1(), WithPrev(), s, hB, mB, lB, hA, mA, lA);
        if (s + 1 < nsteps) {
          run_step(wide_c, P2(), WithPrev(), s + 1, hA, mA, lA, hB, mB, lB);
          if (s + 2 < nsteps) {
            run_step(wide_c, P3(), WithPrev(), s + 2, hB, mB, lB, hA, mA, lA);
            last(s + 2, hB, mB, lB);
          } else {
            last(s + 1, hA, mA, lA);
          }
        } else {
          last(s, hB, mB, lB);
        }
      } else {
        last(s - 1, hA, mA, lA);
      }
    }
  };
  if (EPI == 3 || (NAR == 1 ? rs - a.nar_d : rs) <= 16) { go(integral_constant<bool, false>()); }
  else { go(integral_constant<bool, true>()); }
}

// NB > kMaxRegNB (the 1023-tap shape, NB = 33): one wave per SIMD with the whole 512-entry register file -- 2 * 33 Toeplitz
// fragments are 264 registers (fewer with a high-byte band), next to two accumulator sets and the B-fragment double buffer.
template <int NB, int EPI, int HS, int WAVES, int NAR = 0, bool W4 = false>
__global__ void __launch_bounds__(64 * WAVES, (NB > kMaxRegNB ? 1 : kOccupancy))
fir_mfma_kernel(FirParams p, const v4i *__restrict__ frag, MfmaArgs a) {
  // WAVES == 1: the pipelined body keeps a 4-step ring of staged planes (4 arrays of 128 + NB - 1 slots), the plain body two windows
  constexpr int kStaged = (WAVES == 1 && EPI != 0 && EPI != 3 && 4 * staged_array_bytes(128 + NB - 1) > 2 * 4 * staged_array_bytes(32 + NB - 1))
                              ? 4 * staged_array_bytes(128 + NB - 1) : 2 * 4 * staged_array_bytes(32 + NB - 1);
  __shared__ __attribute__((aligned(16))) unsigned char lds[WAVES * (kStaged + (EPI == 3 ? 8192 : 2048)) + 1024];
  const int64_t s0 = (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;
  // (a lone first step has no in-row window to park the unused prefetch on: see issue_loads_in)
  const bool interior = EPI != 0 && a.out_vec_ok && s1 * 1024 <= p.n && (s0 > 0 || s1 >= 2);
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (WAVES == 1 && EPI != 0) {
    if (interior) {
      // one copy of the loop per form of the increment (epi32_gq); the eight- and nine-block kernels with most of their fragments live
      // spill 6 - 18 VGPRs that way and keep one loop with the form behind uniform branches
      if constexpr (NAR == 2 && NB >= 8 && HS != 3 + 16 * 3) { fir_mfma_pipe_body<NB, EPI, HS, NAR, W4, -1>(p, frag, a, lds); }
      else if constexpr (NAR == 2) {
        if (a.gq_form == 0) { fir_mfma_pipe_body<NB, EPI, HS, NAR, W4, 0>(p, frag, a, lds); }
        else if (a.gq_form == 2) { fir_mfma_pipe_body<NB, EPI, HS, NAR, W4, 2>(p, frag, a, lds); }
        else { fir_mfma_pipe_body<NB, EPI, HS, NAR, W4, 1>(p, frag, a, lds); }
      } else { fir_mfma_pipe_body<NB, EPI, HS, NAR, W4>(p, frag, a, lds); }
    }
    else if constexpr (EPI == 3) { fir_mfma_body<NB, 0, 0, WAVES, false>(p, frag, a, lds); }   // edges: generic epilogue
    else { fir_mfma_body<NB, EPI, HS, WAVES, false>(p, frag, a, lds); }
  } else {
    if (interior) { fir_mfma_body<NB, EPI, HS, WAVES, true>(p, frag, a, lds); }
    else { fir_mfma_body<NB, EPI, HS, WAVES, false>(p, frag, a, lds); }
  }
  if constexpr (WAVES == 1) {
    // small host-side calls (FirParams::hist_next set): the first chunk's wave also writes the channel's next history, so that a
    // one-sample run() of the drop-in classes is ONE launch (cf. fir_hist_update_kernel)
    if (p.hist_next && blockIdx.x == 0 && (int)blockIdx.y < p.n_ch) {
      const int64_t row = (int64_t)blockIdx.y;
      for (int j = threadIdx.x; j < p.hl; j += 64) {
        const int64_t g = p.n - p.hl + j;
        const int64_t v = (g >= 0) ? load_raw(p.x, row * p.in_stride + g, p.in_eb, p.in.S) : load_raw(p.hist, row * p.hl + p.hl + g, p.in_eb, p.in.S);
        store_raw(p.hist_next, row * p.hl + j, p.in_eb, v);
      }
    }
  }
  if (a.dbg && (threadIdx.x & 63) == 0) {
    const int64_t w = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6);
    a.dbg[2 * w] = (int64_t)(__builtin_readcyclecounter() - c0);
    a.dbg[2 * w + 1] = (int64_t)(__builtin_amdgcn_s_memrealtime() - r0);
  }
}

// NAR (OUT_TYPEs of fewer than 16 bits; no band skip) and W4 (4-byte containers) instantiations of up to kMaxRegNB K-blocks live in a
// translation unit of their own (fir_mfma_alt.hip) for compile time
hipError_t launch_fir_mfma_alt(const FirParams &p, int nb, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);
hipError_t launch_fir_mfma_alt2(const FirParams &p, int nb, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);

template <int NB, int HS, int WAVES>
static hipError_t launch_nb_hs(const FirParams &p, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  const dim3 blk(64 * WAVES);
  if constexpr (WAVES == 1 && NB <= kMaxRegNB) {
    if ((HS == 0 && a.nar_on && (epi == 1 || epi == 2)) || (epi == 3 && p.out_eb == 4)) { return launch_fir_mfma_alt(p, NB, HS, d_frag, a, epi, grid, s); }
    if (HS != 0 && a.nar_on && (epi == 1 || epi == 2)) { return launch_fir_mfma_alt2(p, NB, HS, d_frag, a, epi, grid, s); }
  }
  if (epi == 1) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 1, HS, WAVES>), grid, blk, 0, s, p, (const v4i *)d_frag, a); }
  else if (epi == 2) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 2, HS, WAVES>), grid, blk, 0, s, p, (const v4i *)d_frag, a); }
  else if (epi == 3 && WAVES == 1 && NB <= kMaxRegNB) {
    if constexpr (NB <= kMaxRegNB) {
      hipLaunchKernelGGL((fir_mfma_kernel<NB, 3, HS, WAVES>), grid, blk, 0, s, p, (const v4i *)d_frag, a);
    }
  }
  else { hipLaunchKernelGGL((fir_mfma_kernel<NB, 0, 0, WAVES>), grid, blk, 0, s, p, (const v4i *)d_frag, a); }
  return hipGetLastError();
}

#ifndef ACDSP_FIR_TU_MID
// Waves per workgroup of the register-resident kernel.  8 = ping-pong (MFMA run of waves 0-3 against the
// epilogue / staging of waves 4-7, s_barrier between): measured 4 % SLOWER than independent single-wave
// workgroups on MI355X (1.215 vs 1.163 ms on config 2), because MFMA and VALU issue serialise per SIMD
// whatever the pairing; kept selectable for the record.
constexpr int kSmallWaves = 1;
// engine.hip's small-call path (FirParams::hist_next) relies on the single-wave kernel writing the next history itself
// (`if constexpr (WAVES == 1)` in fir_mfma_kernel): with the 8-wave form selected it would silently drop the state.
static_assert(kSmallWaves == 1, "the fused history update of small host-side calls exists in the single-wave kernel only");

// Band skip code (HS = lo + 16 hi) for the non-zero high-byte blocks: each side skips 2 or 3 blocks when the set allows it
// (instantiated: both sides >= 2), else nothing is skipped.
static int pick_hs(int nb, uint64_t hi_mask) {
  if (hi_mask == 0) { return nb >= 7 ? 3 + 16 * 3 : (nb >= 5 ? 2 + 16 * 2 : 0); }
  int lo = 0, hi = 0;
  while (lo < nb && !((hi_mask >> lo) & 1)) { lo++; }
  while (hi < nb && !((hi_mask >> (nb - 1 - hi)) & 1)) { hi++; }
  lo = lo > 3 ? 3 : lo; hi = hi > 3 ? 3 : hi;
  if (nb < 7) { lo = lo > 2 ? 2 : lo; hi = hi > 2 ? 2 : hi; }
  if (nb < 5 || lo < 2 || hi < 2) { return 0; }
  return lo + 16 * hi;
}

// the NAR instantiations (narrow OUT_TYPEs, general rounding modes) exist for the widest and the narrowest skip only: 3 + 3 where the set
// allows it, else 2 + 2 (fir_mfma_alt2.hip)
static int pick_hs_nar(int nb, uint64_t hi_mask) {
  const int hs = pick_hs(nb, hi_mask);
  return hs == 3 + 16 * 3 ? hs : (hs ? 2 + 16 * 2 : 0);
}

template <int NB>
static hipError_t launch_nb(const FirParams &p, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  static const bool no_nar_hs = getenv("ACDSP_NO_NAR_SKIP") != nullptr;   // A/B knob: the round-4 NAR kernels (every high-plane product issued)
  const int hs = !epi ? 0 : (a.nar_on ? (no_nar_hs ? 0 : pick_hs_nar(NB, a.hi_mask)) : pick_hs(NB, a.hi_mask));
  if (NB >= 7) {
    constexpr int H33 = NB >= 7 ? 3 + 16 * 3 : 0, H32 = NB >= 7 ? 3 + 16 * 2 : 0, H23 = NB >= 7 ? 2 + 16 * 3 : 0;
    if (hs == 3 + 16 * 3) { return launch_nb_hs<NB, H33, kSmallWaves>(p, d_frag, a, epi, grid, s); }
    if (hs == 3 + 16 * 2) { return launch_nb_hs<NB, H32, kSmallWaves>(p, d_frag, a, epi, grid, s); }
    if (hs == 2 + 16 * 3) { return launch_nb_hs<NB, H23, kSmallWaves>(p, d_frag, a, epi, grid, s); }
  }
  if (NB >= 5 && hs == 2 + 16 * 2) { return launch_nb_hs<NB, (NB >= 5 ? 2 + 16 * 2 : 0), kSmallWaves>(p, d_frag, a, epi, grid, s); }
  return launch_nb_hs<NB, 0, kSmallWaves>(p, d_frag, a, epi, grid, s);
}


// NB = 33 (993 .. 1025 taps): the register-resident kernel at one wave per SIMD, for coefficient sets whose high-byte band leaves
// 12 or 14 K-blocks free on either side (any unit-gain low-pass in <16,2>: |c| >= 128 only within ~40 taps of the centre, a
// 5-block band -- BASELINE config 4 issues 76 instead of 132 MFMAs per step).  Fragments that would all be live (dense sets: 264
// registers + two accumulator sets) spill; those sets stay on the LDS-resident kernels below.
// Same-box A/B on config 4 (profiles/r3_fir1023_reg33.txt): 2.32 -> 2.10 ms, 0.45 -> 0.50 of the int8 peak in MFMAs issued.
static int reg33_band_code(uint64_t hi_mask, int epi) {
  if (epi != 1 && epi != 2) { return 0; }
  int lo = 0, hi = 0;
  if (hi_mask == 0) { lo = hi = 16; }
  else {
    while (lo < 33 && !((hi_mask >> lo) & 1)) { lo++; }
    while (hi < 33 && !((hi_mask >> (32 - hi)) & 1)) { hi++; }
  }
  if (lo >= 14 && hi >= 14) { return 14 + 16 * 14; }
  if (lo >= 12 && hi >= 12) { return 12 + 16 * 12; }
  return 0;
}
static hipError_t launch_nb33(const FirParams &p, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  if (reg33_band_code(a.hi_mask, epi) == 14 + 16 * 14) { return launch_nb_hs<33, 14 + 16 * 14, 1>(p, d_frag, a, epi, grid, s); }
  return launch_nb_hs<33, 12 + 16 * 12, 1>(p, d_frag, a, epi, grid, s);
}

#endif  // !ACDSP_FIR_TU_MID

// =============================================================================================
// Large tap counts (NB = 10 .. 33, e.g. the 1023-tap configuration): the 2*NB Toeplitz fragments no
// longer fit the register file, so they live in LDS (2 KB per K-block, one shared coefficient set per
// launch) and are read next to the data fragments: four ds_read_b128 per four MFMAs, ~50 % of the LDS
// bandwidth.  Same one-channel-per-wave mapping, eight free-running waves per workgroup (they share only the A
// fragments; ping-pong barriers measured 20 % slower on the double-wide variant below), run-time K loop.
// =============================================================================================
template <int EPI, bool FAST>
__device__ __forceinline__ void fir_mfma_big_body(const FirParams &p, const v4i *__restrict__ frag, const MfmaArgs &a,
                                                  unsigned char *lds_all) {
  const int NB = a.nb;
  const int HB = NB - 1, NC = 32 + HB, NP = 4 * NC, ARR = staged_array_bytes(NC);
  constexpr int JN = 4;   // NP <= 256
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_col = lane & 31, h = lane >> 5;
  int ch = blockIdx.y * 8 + wave;
  if (ch >= p.n_ch) { ch = p.n_ch - 1; }
  // LDS: [A fragments: 2 planes x NB x 1 KB][per wave: 2 x 4 arrays x ARR staging + 2 KB output tile]
  v4i *ldsA = (v4i *)lds_all;
  const int wave_bytes = 2 * 4 * ARR + 2048;
  unsigned char *lds = lds_all + 2 * NB * 1024 + wave * wave_bytes;
  unsigned char *obuf = lds + 2 * 4 * ARR;
  for (int i = threadIdx.x; i < 2 * NB * 64; i += 512) { ldsA[i] = frag[i]; }

  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;
  const int64_t s0 = a.step0 + (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;
  const int nsteps = (int)(s1 - s0);

  v4i R[JN];
  auto issue_loads = [&](int64_t T0) {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
      int64_t t = T0 - 32 * HB + 8 * pc;
      const int16_t *src = (t < 0) ? hrow + t : xrow + ((t < a.n8) ? t : 0);
      R[j] = *(const v4i *)src;
    }
  };
  auto stage = [&](unsigned char *buf) {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = lane + 64 * j;
      if (pc < NP) {
        const int c = pc >> 2, hh_ = (pc >> 1) & 1, sub = pc & 1;
        unsigned hi0 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x07050301u), a.hi_xor);
        unsigned hi1 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x07050301u), a.hi_xor);
        unsigned lo0 = __builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x06040200u) ^ 0x80808080u;
        unsigned lo1 = __builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x06040200u) ^ 0x80808080u;
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        *(v2u *)(buf + (0 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){hi0, hi1};
        *(v2u *)(buf + (1 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){lo0, lo1};
      }
    }
  };

  const int rs = p.in.F + p.cf.F - p.out.F;
  const int64_t corr = a.corr[0];
  const int64_t corr_t = corr + (EPI != 0 ? q_preload(p.out.Q, rs) : 0);
  const int c_ll = (EPI == 1 || EPI == 2) ? (int)corr_t : 0;   // EPI 4 adds C in 64 bits: 128 sum(c) need not fit int32
  const Epi64 e64 = make_epi64(p, corr);
  const v16i ll_init = {c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll};
  int16_t *yrow = (int16_t *)p.y + (int64_t)ch * p.out_stride + 32 * n_col + 4 * h;

  issue_loads(s0 * 1024);
  __syncthreads();          // A fragments visible to every wave (also drains the first loads: once per chunk)
  stage(lds);
  if (FAST || nsteps > 1) { issue_loads((s0 + 1) * 1024); }

  for (int s = 0; s < nsteps; s++) {
    const int64_t T0 = (s0 + s) * 1024;
    const unsigned char *buf = lds + (s & 1) * (4 * ARR);
    const unsigned char *fh = buf + (0 * 2 + h) * ARR + n_col * 16;
    const unsigned char *fl = buf + (1 * 2 + h) * ARR + n_col * 16;
    const v4i *ah = ldsA + lane, *al = ldsA + NB * 64 + lane;

    // ---------------- phase M ----------------
    v16i hh = {0}, mid = {0}, ll = ll_init;
    v4i Ahc = ah[0], Alc = al[0], Bhc = *(const v4i *)fh, Blc = *(const v4i *)fl;
    for (int b = 0; b < NB; b++) {
      v4i Ahn = Ahc, Aln = Alc, Bhn = Bhc, Bln = Blc;
      if (b + 1 < NB) {   // fragments of the next K-block are in flight while this one multiplies
        Ahn = ah[(b + 1) * 64]; Aln = al[(b + 1) * 64];
        Bhn = *(const v4i *)(fh + 16 * (b + 1)); Bln = *(const v4i *)(fl + 16 * (b + 1));
      }
      if (b >= a.hb0 && b <= a.hb1) {
        hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahc, Bhc, hh, 0, 0, 0);
        mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ahc, Blc, mid, 0, 0, 0);
      }
      ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alc, Blc, ll, 0, 0, 0);
      mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Alc, Bhc, mid, 0, 0, 0);
      Ahc = Ahn; Alc = Aln; Bhc = Bhn; Blc = Bln;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase O ----------------
    int o16[16];
    if (EPI == 4) { epi64(hh, mid, ll, e64, o16); }
    else if (EPI != 0) { epi32<false>(hh, mid, ll, rs - a.nar_d, a, o16); }
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int64_t t0 = T0 + 32 * n_col + 8 * g + 4 * h;
      if (EPI != 0) {
        const int *o = o16 + 4 * g;
        v4s pk;
        if (EPI == 2) {
          typedef short v2s __attribute__((ext_vector_type(2)));
          const v2s p0 = __builtin_amdgcn_cvt_pk_i16(o[0], o[1]), p1 = __builtin_amdgcn_cvt_pk_i16(o[2], o[3]);
          pk = (v4s){p0.x, p0.y, p1.x, p1.y};
        } else {
          pk = (v4s){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
        }
        int16_t *dst = yrow + T0 + 8 * g;
        if (FAST) {
          const int P = 4 * n_col + g;
          *(v4s *)(obuf + (((P & ~15) | ((P + (P >> 4)) & 15)) * 16 + 8 * h)) = pk;
        } else if (a.out_vec_ok && t0 + 4 <= p.n) {
          *(v4s *)dst = pk;
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            if (t0 + rr < p.n) { dst[rr] = pk[rr]; }   // pk: already clamped for AC_SAT
          }
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int r = 4 * g + rr;
          int64_t v = ((int64_t)hh[r] << 16) + ((int64_t)mid[r] << 8) + (int64_t)ll[r] + corr;
          int64_t acc = wrap64((int64_t)((uint64_t)v << p.lossless_shift), p.acc.W, p.acc.S);
          int64_t y = requant64(acc, p.acc.F, p.out);
          if (t0 + rr < p.n) { store_raw(p.y, (int64_t)ch * p.out_stride + t0 + rr, p.out_eb, y); }
        }
      }
    }
    if (FAST && EPI != 0) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int P = 64 * half + lane;
        const v4i val = *(const v4i *)(obuf + ((P & ~15) | ((P + (P >> 4)) & 15)) * 16);
        *(v4i *)((int16_t *)p.y + (int64_t)ch * p.out_stride + T0 + 512 * half + 8 * lane) = val;
      }
    }
    if (s + 1 < nsteps) {
      stage(lds + ((s + 1) & 1) * (4 * ARR));
      if (FAST || s + 2 < nsteps) { issue_loads(T0 + 2048); }
    }
  }
}

template <int EPI>
__global__ void __launch_bounds__(512, 2)
fir_mfma_big_kernel(FirParams p, const v4i *__restrict__ frag, MfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  const int64_t s0 = a.step0 + (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;
  const bool interior = EPI != 0 && a.out_vec_ok && s1 * 1024 <= p.n;
  if (interior) { fir_mfma_big_body<EPI, true>(p, frag, a, lds_dyn); }
  else { fir_mfma_big_body<EPI, false>(p, frag, a, lds_dyn); }
}

// ---------------------------------------------------------------------------------------------------
// Double-wide variant for complete chunks: a step is 2048 outputs (two column sets of 32 blocks), so every
// A fragment fetched from LDS feeds eight MFMAs instead of four.  With NB = 33 the single-wide kernel moves
// 4 KB of LDS per four MFMAs per wave -- ~220 of the 256 B/clk the LDS delivers, i.e. it is LDS-bound as much as
// MFMA-bound; here it is 6 KB per eight.  The two sets share their halo (one staged array of 64 + NB - 1 chunks,
// single-buffered: a wave stages step s+1 in its own O phase, after its own M phase is done reading).  The eight
// waves of a workgroup share only the A fragments and run free (the ping-pong barriers of the single-wide kernel
// cost 20 % here: 2.86 -> 2.30 ms on config 4).
// Launched over chunks of complete double steps only (EPI 1 / 2); the ragged rest goes to fir_mfma_big_kernel.
template <int EPI>
__global__ void __launch_bounds__(512, 1)
fir_mfma_big2_kernel(FirParams p, const v4i *__restrict__ frag, MfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
  const int NB = a.nb;
  const int HB = NB - 1, NC = 64 + HB, NP = 4 * NC, ARR = staged_array_bytes(NC);
  constexpr int JN = 6;   // NP <= 4 * 96
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_col = lane & 31, h = lane >> 5;
  int ch = blockIdx.y * 8 + wave;
  if (ch >= p.n_ch) { ch = p.n_ch - 1; }
  v4i *ldsA = (v4i *)lds_all;
  const int wave_bytes = 4 * ARR + 4096;
  unsigned char *lds = lds_all + 2 * NB * 1024 + wave * wave_bytes;
  unsigned char *obuf = lds + 4 * ARR;
  for (int i = threadIdx.x; i < 2 * NB * 64; i += 512) { ldsA[i] = frag[i]; }

  const int16_t *xrow = (const int16_t *)p.x + (int64_t)ch * p.in_stride;
  const int16_t *hrow = (const int16_t *)p.hist + (int64_t)ch * p.hl + p.hl;
  const int64_t d0 = (int64_t)blockIdx.x * a.steps_per_wave;   // in double steps; the launch covers complete chunks
  const int nsteps = (int)a.steps_per_wave;
  const int64_t t_last = (d0 + nsteps - 1) * 2048;

  v4i R[JN];
  auto issue_loads = [&](int64_t T0) {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = (lane + 64 * j < NP) ? lane + 64 * j : NP - 1;
      int64_t t = T0 - 32 * HB + 8 * pc;
      const int16_t *src = (t < 0) ? hrow + t : xrow + ((t < a.n8) ? t : 0);
      R[j] = *(const v4i *)src;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < JN; j++) {
      const int pc = lane + 64 * j;
      if (pc < NP) {
        const int c = pc >> 2, hh_ = (pc >> 1) & 1, sub = pc & 1;
        unsigned hi0 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x07050301u), a.hi_xor);
        unsigned hi1 = hi_flip<EPI == 3>(__builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x07050301u), a.hi_xor);
        unsigned lo0 = __builtin_amdgcn_perm((unsigned)R[j].y, (unsigned)R[j].x, 0x06040200u) ^ 0x80808080u;
        unsigned lo1 = __builtin_amdgcn_perm((unsigned)R[j].w, (unsigned)R[j].z, 0x06040200u) ^ 0x80808080u;
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        *(v2u *)(lds + (0 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){hi0, hi1};
        *(v2u *)(lds + (1 * 2 + hh_) * ARR + c * 16 + sub * 8) = (v2u){lo0, lo1};
      }
    }
  };

  const int rs = p.in.F + p.cf.F - p.out.F;
  const int c_ll = EPI == 4 ? 0 : (int)(a.corr[0] + q_preload(p.out.Q, rs));
  const Epi64 e64 = make_epi64(p, a.corr[0]);
  const v16i ll_init = {c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll, c_ll};
  int16_t *yout = (int16_t *)p.y + (int64_t)ch * p.out_stride;
  const unsigned char *fh = lds + (0 * 2 + h) * ARR + n_col * 16;
  const unsigned char *fl = lds + (1 * 2 + h) * ARR + n_col * 16;
  const v4i *ah = ldsA + lane, *al = ldsA + NB * 64 + lane;

  issue_loads(d0 * 2048);
  __syncthreads();          // A fragments visible to every wave
  stage();
  issue_loads(nsteps > 1 ? (d0 + 1) * 2048 : t_last);

  for (int s = 0; s < nsteps; s++) {
    const int64_t T0 = (d0 + s) * 2048;
    // ---------------- phase M: column set 0 = chunks n + b, set 1 = chunks 32 + n + b ----------------
    v16i h0 = {0}, m0 = {0}, l0 = ll_init, h1 = {0}, m1 = {0}, l1 = ll_init;
    // fragments of K-block b live in slot b % 3 and are fetched two blocks ahead (one block = 4..8 MFMAs = 128..256
    // cycles, less than an LDS round trip when all eight waves of the workgroup are reading)
    struct Frag { v4i Ah, Al, Bh0, Bl0, Bh1, Bl1; };
    auto fetch = [&](Frag &f, int bb) {
      f.Ah = ah[bb * 64]; f.Al = al[bb * 64];
      f.Bh0 = *(const v4i *)(fh + 16 * bb); f.Bl0 = *(const v4i *)(fl + 16 * bb);
      f.Bh1 = *(const v4i *)(fh + 512 + 16 * bb); f.Bl1 = *(const v4i *)(fl + 512 + 16 * bb);
    };
    auto mac = [&](const Frag &f, int bb) {
      if (bb >= a.hb0 && bb <= a.hb1) {
        h0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Ah, f.Bh0, h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Ah, f.Bh1, h1, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Ah, f.Bl0, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Ah, f.Bl1, m1, 0, 0, 0);
      }
      l0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Al, f.Bl0, l0, 0, 0, 0);
      l1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Al, f.Bl1, l1, 0, 0, 0);
      m0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Al, f.Bh0, m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.Al, f.Bh1, m1, 0, 0, 0);
    };
    Frag F0, F1, F2;
    fetch(F0, 0);
    if (NB > 1) { fetch(F1, 1); } else { F1 = F0; }
    F2 = F0;
    for (int b = 0; b < NB; b += 3) {
      if (b + 2 < NB) { fetch(F2, b + 2); }
      mac(F0, b);
      if (b + 1 < NB) {
        if (b + 3 < NB) { fetch(F0, b + 3); }
        mac(F1, b + 1);
      }
      if (b + 2 < NB) {
        if (b + 4 < NB) { fetch(F1, b + 4); }
        mac(F2, b + 2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase O ----------------
#pragma unroll
    for (int set = 0; set < 2; set++) {
      int o[16];
      if (EPI == 4) { if (set == 0) { epi64(h0, m0, l0, e64, o); } else { epi64(h1, m1, l1, e64, o); } }
      else if (set == 0) { epi32<false>(h0, m0, l0, rs - a.nar_d, a, o); } else { epi32<false>(h1, m1, l1, rs - a.nar_d, a, o); }
#pragma unroll
      // (the permlane32-swap / ds_write_b128 tile of fir_mfma_pipe_body was tried here: conflicts 2.5e7 -> 0 but +0.8 % time,
      // the phase is not in the shadow of MFMAs)
      for (int g = 0; g < 4; g++) {
        v4s pk;
        if (EPI == 2) {
          typedef short v2s __attribute__((ext_vector_type(2)));
          const v2s p0 = __builtin_amdgcn_cvt_pk_i16(o[4 * g], o[4 * g + 1]), p1 = __builtin_amdgcn_cvt_pk_i16(o[4 * g + 2], o[4 * g + 3]);
          pk = (v4s){p0.x, p0.y, p1.x, p1.y};
        } else {
          pk = (v4s){(short)o[4 * g], (short)o[4 * g + 1], (short)o[4 * g + 2], (short)o[4 * g + 3]};
        }
        const int P = 4 * n_col + g;
        *(v4s *)(obuf + 2048 * set + (((P & ~15) | ((P + (P >> 4)) & 15)) * 16 + 8 * h)) = pk;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {   // 4 KB contiguous: four coalesced 16-byte-per-lane stores
      const int P = 64 * (q & 1) + lane;
      const v4i val = *(const v4i *)(obuf + 2048 * (q >> 1) + ((P & ~15) | ((P + (P >> 4)) & 15)) * 16);
#if ACDSP_FIR_NT & 8   // -0.4 % same box (profiles/r2_ab_nt.txt)
      __builtin_nontemporal_store(val, (v4i *)(yout + T0 + 512 * q + 8 * lane));
#else
      *(v4i *)(yout + T0 + 512 * q + 8 * lane) = val;
#endif
    }
    if (s + 1 < nsteps) {
      stage();
      const int64_t tn = T0 + 4096;
      issue_loads(tn < t_last ? tn : t_last);
    }
  }
}

#ifdef ACDSP_FIR_TU_MID
// ---- further translation units (fir_mfma_mid*.hip): the register-resident shapes for 11 .. 31 K-blocks, split three ways for compile time ----
#if ACDSP_FIR_TU_MID == 1
hipError_t launch_fir_mfma_mid(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  switch (nb) {
    case 11: return launch_nb_hs<11, 3 + 16 * 3, 1>(p, d_frag, a, epi, grid, s);
    case 13: return launch_nb_hs<13, 4 + 16 * 4, 1>(p, d_frag, a, epi, grid, s);
    case 15: return launch_nb_hs<15, 5 + 16 * 5, 1>(p, d_frag, a, epi, grid, s);
    case 17: return launch_nb_hs<17, 6 + 16 * 6, 1>(p, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#elif ACDSP_FIR_TU_MID == 2
hipError_t launch_fir_mfma_mid2(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  switch (nb) {
    case 19: return launch_nb_hs<19, 7 + 16 * 7, 1>(p, d_frag, a, epi, grid, s);
    case 21: return launch_nb_hs<21, 8 + 16 * 8, 1>(p, d_frag, a, epi, grid, s);
    case 23: return launch_nb_hs<23, 9 + 16 * 9, 1>(p, d_frag, a, epi, grid, s);
    case 25: return launch_nb_hs<25, 10 + 16 * 10, 1>(p, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#elif ACDSP_FIR_TU_MID == 4
template <int NB, int HS>
static hipError_t launch_alt_hs(const FirParams &p, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  const v4i *f = (const v4i *)d_frag;
  if (epi == 3) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 3, HS, 1, 0, true>), grid, dim3(64), 0, s, p, f, a); }
  else if constexpr (HS == 0) {
    // (the general-rounding epilogue has instantiations of its own: inside the narrow-type kernels it spilled 5 - 13 registers at 6 and 9 K-blocks)
    if (a.gq_on) {
      if (epi == 1) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 1, 0, 1, 2>), grid, dim3(64), 0, s, p, f, a); }
      else { hipLaunchKernelGGL((fir_mfma_kernel<NB, 2, 0, 1, 2>), grid, dim3(64), 0, s, p, f, a); }
    }
    else if (epi == 1) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 1, 0, 1, 1>), grid, dim3(64), 0, s, p, f, a); }
    else { hipLaunchKernelGGL((fir_mfma_kernel<NB, 2, 0, 1, 1>), grid, dim3(64), 0, s, p, f, a); }
  } else { return hipErrorInvalidValue; }
  return hipGetLastError();
}
template <int NB>
static hipError_t launch_alt_nb(const FirParams &p, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  if constexpr (NB >= 7) {   // the band-skip codes launch_nb picks
    if (hs == 3 + 16 * 3) { return launch_alt_hs<NB, 3 + 16 * 3>(p, d_frag, a, epi, grid, s); }
    if (hs == 3 + 16 * 2) { return launch_alt_hs<NB, 3 + 16 * 2>(p, d_frag, a, epi, grid, s); }
    if (hs == 2 + 16 * 3) { return launch_alt_hs<NB, 2 + 16 * 3>(p, d_frag, a, epi, grid, s); }
  }
  if constexpr (NB >= 5) { if (hs == 2 + 16 * 2) { return launch_alt_hs<NB, 2 + 16 * 2>(p, d_frag, a, epi, grid, s); } }
  return hs == 0 ? launch_alt_hs<NB, 0>(p, d_frag, a, epi, grid, s) : hipErrorInvalidValue;
}
hipError_t launch_fir_mfma_alt(const FirParams &p, int nb, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  switch (nb) {
    case 1: return launch_alt_nb<1>(p, hs, d_frag, a, epi, grid, s);
    case 2: return launch_alt_nb<2>(p, hs, d_frag, a, epi, grid, s);
    case 3: return launch_alt_nb<3>(p, hs, d_frag, a, epi, grid, s);
    case 4: return launch_alt_nb<4>(p, hs, d_frag, a, epi, grid, s);
    case 5: return launch_alt_nb<5>(p, hs, d_frag, a, epi, grid, s);
    case 6: return launch_alt_nb<6>(p, hs, d_frag, a, epi, grid, s);
    case 7: return launch_alt_nb<7>(p, hs, d_frag, a, epi, grid, s);
    case 8: return launch_alt_nb<8>(p, hs, d_frag, a, epi, grid, s);
    case 9: return launch_alt_nb<9>(p, hs, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#elif ACDSP_FIR_TU_MID == 5
// NAR instantiations with a band skip (round 5): band-limited sets into narrow OUT_TYPEs or through the general rounding modes ran every
// high-plane product (36 instead of 24 MFMAs per step at nine blocks) -- that, not the epilogue, was most of their distance to the plain classes
template <int NB, int HS>
static hipError_t launch_alt2_hs(const FirParams &p, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  const v4i *f = (const v4i *)d_frag;
  if (a.gq_on) {
    if (epi == 1) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 1, HS, 1, 2>), grid, dim3(64), 0, s, p, f, a); }
    else { hipLaunchKernelGGL((fir_mfma_kernel<NB, 2, HS, 1, 2>), grid, dim3(64), 0, s, p, f, a); }
  }
  else if (epi == 1) { hipLaunchKernelGGL((fir_mfma_kernel<NB, 1, HS, 1, 1>), grid, dim3(64), 0, s, p, f, a); }
  else { hipLaunchKernelGGL((fir_mfma_kernel<NB, 2, HS, 1, 1>), grid, dim3(64), 0, s, p, f, a); }
  return hipGetLastError();
}
template <int NB>
static hipError_t launch_alt2_nb(const FirParams &p, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  if constexpr (NB >= 7) { if (hs == 3 + 16 * 3) { return launch_alt2_hs<NB, 3 + 16 * 3>(p, d_frag, a, epi, grid, s); } }
  return hs == 2 + 16 * 2 ? launch_alt2_hs<NB, 2 + 16 * 2>(p, d_frag, a, epi, grid, s) : hipErrorInvalidValue;
}
hipError_t launch_fir_mfma_alt2(const FirParams &p, int nb, int hs, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  switch (nb) {
    case 5: return launch_alt2_nb<5>(p, hs, d_frag, a, epi, grid, s);
    case 6: return launch_alt2_nb<6>(p, hs, d_frag, a, epi, grid, s);
    case 7: return launch_alt2_nb<7>(p, hs, d_frag, a, epi, grid, s);
    case 8: return launch_alt2_nb<8>(p, hs, d_frag, a, epi, grid, s);
    case 9: return launch_alt2_nb<9>(p, hs, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#else
hipError_t launch_fir_mfma_mid3(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  switch (nb) {
    case 27: return launch_nb_hs<27, 11 + 16 * 11, 1>(p, d_frag, a, epi, grid, s);
    case 29: return launch_nb_hs<29, 12 + 16 * 12, 1>(p, d_frag, a, epi, grid, s);
    case 31: return launch_nb_hs<31, 13 + 16 * 13, 1>(p, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}
#endif
#else
hipError_t launch_fir_mfma_mid(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);
hipError_t launch_fir_mfma_mid2(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);
hipError_t launch_fir_mfma_mid3(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);

static hipError_t launch_big(const FirParams &p, const uint32_t *d_frag, MfmaArgs a, int epi, dim3 grid, hipStream_t s) {
  const int nb = a.nb;
  // contiguous range of non-zero high-byte blocks
  a.hb0 = 0; a.hb1 = nb - 1;
  if (epi) {
    if (a.hi_mask == 0) { a.hb0 = 1; a.hb1 = 0; }
    else {
      while (!((a.hi_mask >> a.hb0) & 1)) { a.hb0++; }
      while (!((a.hi_mask >> a.hb1) & 1)) { a.hb1--; }
    }
  }
  hipError_t e = hipSuccess;
  a.step0 = 0;
  // complete chunks of double steps on the double-wide kernel ...
  const int64_t spw2 = a.steps_per_wave >= 8 ? a.steps_per_wave / 2 : 4;
  const int64_t fast_chunks = (epi == 1 || epi == 2 || epi == 4) && a.out_vec_ok ? (p.n / 2048) / spw2 : 0;
  if (fast_chunks > 0) {
    MfmaArgs a2 = a;
    a2.steps_per_wave = spw2;
    const size_t lds2 = (size_t)2 * nb * 1024 + 8 * ((size_t)4 * staged_array_bytes(64 + nb - 1) + 4096);
    dim3 g2((unsigned)fast_chunks, grid.y);
    if (epi == 1) {
      e = hipFuncSetAttribute((const void *)fir_mfma_big2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big2_kernel<1>), g2, dim3(512), lds2, s, p, (const v4i *)d_frag, a2); }
    } else if (epi == 4) {
      e = hipFuncSetAttribute((const void *)fir_mfma_big2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big2_kernel<4>), g2, dim3(512), lds2, s, p, (const v4i *)d_frag, a2); }
    } else {
      e = hipFuncSetAttribute((const void *)fir_mfma_big2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big2_kernel<2>), g2, dim3(512), lds2, s, p, (const v4i *)d_frag, a2); }
    }
    if (e != hipSuccess) { return e; }
    a.step0 = fast_chunks * spw2 * 2;
    if (a.step0 >= a.n_steps) { return hipGetLastError(); }
    grid.x = (unsigned)((a.n_steps - a.step0 + a.steps_per_wave - 1) / a.steps_per_wave);
  }
  // ... the ragged rest (and the generic epilogue class) on the single-wide kernel
  const size_t lds_bytes = (size_t)2 * nb * 1024 + 8 * ((size_t)2 * 4 * staged_array_bytes(32 + nb - 1) + 2048);
  if (epi == 1) {
    e = hipFuncSetAttribute((const void *)fir_mfma_big_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big_kernel<1>), grid, dim3(512), lds_bytes, s, p, (const v4i *)d_frag, a); }
  } else if (epi == 2) {
    e = hipFuncSetAttribute((const void *)fir_mfma_big_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big_kernel<2>), grid, dim3(512), lds_bytes, s, p, (const v4i *)d_frag, a); }
  } else if (epi == 4) {
    e = hipFuncSetAttribute((const void *)fir_mfma_big_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big_kernel<4>), grid, dim3(512), lds_bytes, s, p, (const v4i *)d_frag, a); }
  } else {
    e = hipFuncSetAttribute((const void *)fir_mfma_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) { hipLaunchKernelGGL((fir_mfma_big_kernel<0>), grid, dim3(512), lds_bytes, s, p, (const v4i *)d_frag, a); }
  }
  return e != hipSuccess ? e : hipGetLastError();
}

// Can the all-32-bit epilogue be used for this plan / type combination?  0: no, 1: WRAP, 2: SAT
// a rounding / overflow mode beyond the AC_TRN / AC_RND x AC_WRAP / AC_SAT of the plain fast classes
static bool fir_mfma_general_q(const FirParams &p) {
  return !(q_const_mode(p.out.Q) && (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT));
}
int fir_mfma_epilogue_class(const FirParams &p, const FirMfmaPlan &plan) {
  // an unsigned AC_WRAP accumulator turns every negative sum into 2^W - |v| before OUT_TYPE sees it (the reference's
  // `acc += ...` assignment): only the generic epilogue wraps to ACC_TYPE, the fast ones convert the signed sum
  if (!p.acc.S) { return 0; }
  const int rs = p.in.F + p.cf.F - p.out.F;
  // |y| <= 32768 * sum|c| must fit the accumulator (no AC_WRAP event possible) ...
  const int acc_bits = p.acc.W - (p.acc.S ? 1 : 0) - p.lossless_shift;
  const int64_t x_max = int64_t(1) << (p.in.W > 1 && p.in.W <= 16 ? p.in.W - (p.in.S ? 1 : 0) : 15);   // |x| <= 2^(W_in - 1) (unsigned: < 2^W_in): narrow samples need a narrower accumulator
  const bool acc_wide = acc_bits >= 63 || plan.sum_abs * x_max < (int64_t(1) << (acc_bits > 0 ? acc_bits : 0));
  // ... hh*256 + mid + carry must fit int32, and so must the low plane with corr + rounding constant preloaded
  const int64_t hh_max = 128 * plan.sum_abs_hi, mid_max = 128 * (plan.sum_abs_hi + plan.sum_abs_lo), ll_max = 128 * plan.sum_abs_lo;
  // (the sign- / parity-dependent modes add up to K + |C| <= 2^rs to lo before the shift: epi32_gq)
  const int64_t rnd = q_const_mode(p.out.Q) ? (rs <= 38 ? q_preload(p.out.Q, rs) : 0) : (rs >= 1 && rs <= 31 ? (int64_t(1) << rs) : 0);
  const int64_t corr_abs = (plan.corr < 0 ? -plan.corr : plan.corr) + rnd;
  // lo = 2^8 mid + ll (with the preloaded constant) and the shifted sum must stay inside int32
  // (OUT_TYPEs of W < 16 bits shift by rse = rs - (16 - W) and finish on the packed words: MfmaArgs::nar_*)
  const int nar_d = (p.out_eb == 2 && p.out.W >= 2 && p.out.W < 16) ? 16 - p.out.W : 0, rse = rs - nar_d;
  const bool small = mid_max * 256 + ll_max + corr_abs + 2 < (int64_t(1) << 31) &&
                     (rse <= 16 ? (hh_max << (16 - (rse < 16 ? (rse > 0 ? rse : 0) : 16))) + ((mid_max * 256 + ll_max + corr_abs) >> (rse > 0 ? rse : 0)) + 2
                                : hh_max + ((mid_max * 256 + ll_max + corr_abs) >> 16) + 2) < (int64_t(1) << 31);
  if (p.out_eb == 2 && p.out.S && q_const_mode(p.out.Q) && (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) &&
      rse >= 1 && rs <= 31 && acc_wide && small && (p.out.W == 16 || (nar_d > 0 && plan.nb <= kMaxRegNB))) {
    return p.out.O == ACDSP_SAT ? 2 : 1;
  }
  // the other rounding modes and AC_SAT_SYM / AC_SAT_ZERO on full 16-bit OUT_TYPEs: the same classes with the increment of the dropped
  // bits (MfmaArgs::gq_*; NAR instantiations: register-resident shapes of up to kMaxRegNB K-blocks).  ACDSP_NO_GQ: generic class (A/B knob)
  static const bool no_gq = getenv("ACDSP_NO_GQ") != nullptr;
  if (!no_gq && fir_mfma_general_q(p) && p.out_eb == 2 && p.out.S && p.out.W == 16 && rs >= 1 && rs <= 31 && acc_wide && small && plan.nb <= kMaxRegNB) {
    return p.out.O == ACDSP_WRAP ? 1 : 2;
  }
  // EPI 4: int16 containers past the 32-bit bounds, on the LDS-resident kernels (more than kMaxRegNB K-blocks): exact 64-bit recombination,
  // ACC_TYPE wrap included, branch-free (epi64).  ACDSP_NO_EPI4: the generic class instead (A/B knob).
  static const bool no_epi4 = getenv("ACDSP_NO_EPI4") != nullptr;
  if (!no_epi4 && plan.nb > kMaxRegNB && p.out_eb == 2 && p.out.S && q_const_mode(p.out.Q) &&
      (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) && p.acc.W >= 2 && p.acc.W <= 62 && p.lossless_shift >= 0 && p.lossless_shift <= 32 &&
      p.acc.F - p.out.F >= 1 && p.acc.F - p.out.F <= 62 && p.out.W >= 2 && p.out.W <= 16) {
    return 4;
  }
  // 4-byte containers (W_out 17 .. 32): the wide class with an int32 tile; AC_WRAP or AC_SAT
  if (p.out_eb == 4 && p.out.S && q_const_mode(p.out.Q) && (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) && acc_wide &&
      rs >= 0 && rs <= 38 && p.out.W >= 2 && p.out.W <= 32 && ll_max + corr_abs + 2 < (int64_t(1) << 31) && plan.nb <= kMaxRegNB) {
    return 3;
  }
  // wide rows: 64-bit shift-and-wrap epilogue of the pipelined body (the low plane still carries C in 32 bits)
  if (p.out_eb == 8 && p.out.S && q_const_mode(p.out.Q) && p.out.O == ACDSP_WRAP && acc_wide &&
      rs >= -16 && rs <= 38 && p.out.W >= 2 && p.out.W <= 64 && (rs < 0 ? -rs : 0) + (64 - p.out.W) <= 63 && ll_max + corr_abs + 2 < (int64_t(1) << 31) && plan.nb <= kMaxRegNB) {
    return 3;
  }
  return 0;
}

static hipError_t launch_switch(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s);

// 32x32x32 int8 MFMAs the selected kernel issues per 1024 samples of one channel: two low-plane products per K-block plus two
// high-plane products per K-block of the instantiated band (the honest numerator of an MFMA-utilisation figure; the dense
// formulation would be 4 * nb).
int fir_mfma_issued_per_step(const FirParams &p, const FirMfmaPlan &plan) {
  const int epi = fir_mfma_epilogue_class(p, plan), nb = plan.nb;
  int band = nb;
  if (nb <= kMaxRegNB) {
    const bool nar = (epi == 1 || epi == 2) && p.out_eb == 2 && (p.out.W < 16 || fir_mfma_general_q(p));   // NAR instantiations
    static const bool no_nar_hs = getenv("ACDSP_NO_NAR_SKIP") != nullptr;
    const int hs = !epi ? 0 : (nar ? (no_nar_hs ? 0 : pick_hs_nar(nb, plan.hi_mask)) : pick_hs(nb, plan.hi_mask));
    band = nb - (hs & 15) - (hs >> 4);
  } else if (epi) {
    int b0 = 0, b1 = nb - 1;
    if (plan.hi_mask == 0) { band = 0; b0 = nb; b1 = -1; }
    else {
      while (!((plan.hi_mask >> b0) & 1)) { b0++; }
      while (!((plan.hi_mask >> b1) & 1)) { b1--; }
      band = b1 - b0 + 1;
    }
    if (use_reg33(nb, plan.hi_mask, epi)) { const int code = reg33_band_code(plan.hi_mask, epi); band = nb - (code & 15) - (code >> 4); }
    if (use_mid(nb, plan.hi_mask, epi)) { band = 5; }
  }
  return 2 * nb + 2 * band;
}

// true if the plan runs on a kernel that keeps the Toeplitz fragments in registers (the only ones that can take a coefficient set per
// channel): up to 9 K-blocks always, more when the set's high-byte band fits a compiled shape and the fast int16 epilogue applies
bool fir_mfma_register_resident(const FirParams &p, const FirMfmaPlan &plan) {
  if (plan.nb <= kMaxRegNB) { return true; }
  const int epi = fir_mfma_epilogue_class(p, plan);
  return use_mid(plan.nb, plan.hi_mask, epi) || use_reg33(plan.nb, plan.hi_mask, epi);
}

hipError_t launch_fir_mfma(const FirParams &p, const FirMfmaPlan &plan, int frag_per_channel, const uint32_t *d_frag,
                           const int64_t *d_corr, hipStream_t s) {
  if (p.n <= 0) { return hipSuccess; }
  const int epi = fir_mfma_epilogue_class(p, plan);
  MfmaArgs a;
  a.nar_on = 0; a.nar_d = 0; a.nar_lo = INT32_MIN; a.nar_hi = INT32_MAX; a.nar_sh = 0;
  a.gq_on = 0; a.gq_off = 0; a.gq_c = 0; a.gq_k = 0; a.gq_form = 1; a.gq_lo = INT32_MIN; a.gq_hi = INT32_MAX;
  a.hi_xor = p.in_flip ? 0x80808080u : 0u;
  if ((epi == 1 || epi == 2) && p.out_eb == 2 && fir_mfma_general_q(p)) {   // W_out = 16 (fir_mfma_epilogue_class)
    const int rsq = p.in.F + p.cf.F - p.out.F;                               // 1 .. 31 (fir_mfma_epilogue_class)
    const int32_t half = (int32_t)(uint32_t(1) << (rsq - 1)), all = (int32_t)((uint32_t(1) << rsq) - 1u);
    a.nar_on = 1; a.gq_on = 1;
    if (p.out.O == ACDSP_SAT_SYM) { a.gq_lo = -32767; a.gq_hi = 32767; }
    switch (p.out.Q) {   // (the constant modes keep their constant in ll like the plain classes: no increment on top of it)
      case ACDSP_TRN_ZERO:     a.gq_off = 31; a.gq_c = all; a.gq_k = 0; break;
      case ACDSP_RND_ZERO:     a.gq_off = 31; a.gq_c = 1;   a.gq_k = half - 1; break;
      case ACDSP_RND_INF:      a.gq_off = 31; a.gq_c = -1;  a.gq_k = half; break;
      case ACDSP_RND_CONV:     a.gq_off = 0;  a.gq_c = 1;   a.gq_k = half - 1; break;
      case ACDSP_RND_CONV_ODD: a.gq_off = 0;  a.gq_c = -1;  a.gq_k = half; break;
      default: break;
    }
    a.gq_form = p.out.O == ACDSP_SAT_ZERO ? 2 : ((a.gq_off == 0 && rsq < 16) ? 0 : 1);
  }
  a.w4_sat = (epi == 3 && p.out_eb == 4 && p.out.O == ACDSP_SAT) ? 1 : 0; a.w4_lo = p.out.lo; a.w4_hi = p.out.hi;
  if (epi == 3 && p.out_eb == 4 && p.out.O == ACDSP_WRAP) {
    const int rs4 = p.in.F + p.cf.F - p.out.F;
    const int64_t rnd4 = q_preload(p.out.Q, rs4);
    const int64_t lo_max = 128 * (plan.sum_abs_hi + plan.sum_abs_lo) * 256 + 128 * plan.sum_abs_lo + (plan.corr < 0 ? -plan.corr : plan.corr) + rnd4 + 2;
    if (rs4 >= 1 && rs4 <= 31 && lo_max < (int64_t(1) << 31) && 128 * plan.sum_abs_hi + (lo_max >> 16) + 2 < (int64_t(1) << 31)) { a.w4_sat = 2; }
  }
  if ((epi == 1 || epi == 2) && p.out_eb == 2 && p.out.W < 16) {   // (class 1 / 2 with fewer than 16 bits: at most kMaxRegNB K-blocks)
    a.nar_on = 1; a.nar_d = 16 - p.out.W;
    if (epi == 2) { a.nar_lo = (int32_t)p.out.lo; a.nar_hi = (int32_t)p.out.hi; }
    else { a.nar_sh = 32 - p.out.W; }
  }
  a.n_steps = (p.n + 1023) / 1024;
  a.n8 = (p.n + 7) / 8 * 8;
  // >= 16384 waves when the problem allows it; a chunk re-reads NB-1 halo blocks per step anyway
  int64_t spw = (a.n_steps * p.n_ch + 16383) / 16384;
  if (spw < 8) { spw = 8; }
  // Short filters (<= 4 K-blocks, up to 97 taps) are bound by the memory system, not by the MFMAs or the power cap: 32 KB spans per wave,
  // the best span of a bare copy (tools/copy_probe2.hip), run them 13 % faster than 128 KB spans (15 - 95 taps: 0.733 against 0.84 ms per
  // 1024 ch x 2^20 samples, same box; 12 and 24 steps: 0.76 / 0.78).  From 127 taps on 16 .. 64 steps measure alike (profiles/r3_taps_sweep.txt).
  if (plan.nb <= 4 && spw > 16) { spw = 16; }
  // 8-byte outputs (OUT = ACC rows: 2 KB read and 8 KB written per step): 8 steps per wave, 1.96 against 2.07 ms at 64 on the config-2
  // wide row (2 / 4 / 6 / 12 / 16 steps: 2.02 / 1.98 / 1.98 / 2.02 / 2.02; same box, two passes)
  if (epi == 3 && spw > 8) { spw = 8; }
  ACDSP_TUNE_ENV(spw_env, "ACDSP_FIR_SPW");   // tuning knob: 1024-sample steps per wave
  if (spw_env && atoi(spw_env) > 1) { spw = atoi(spw_env); }
  a.steps_per_wave = spw;
  const int oeb = p.out_eb;
  // vector stores at any element-aligned address (gfx950 takes them: 652 parity tests with unaligned rows forced through the vector
  // paths, profiles/r3_unaligned.txt); ACDSP_ALIGNED_ONLY=1 restores the round-2 rule (guarded element-wise stores for such rows)
  static const bool aligned_only = getenv("ACDSP_ALIGNED_ONLY") != nullptr;
  a.out_vec_ok = !aligned_only || (((uintptr_t)p.y % (4 * oeb) == 0) && ((p.out_stride * oeb) % (4 * oeb) == 0));
  a.frag_per_channel = frag_per_channel;
  a.hi_mask = plan.hi_mask;
  a.lo_mask = plan.lo_mask;
  a.nb = plan.nb; a.hb0 = 0; a.hb1 = plan.nb - 1; a.step0 = 0;
  a.corr = d_corr;
  const int wpb = (plan.nb > kMaxRegNB && !use_reg33(plan.nb, plan.hi_mask, epi) && !use_mid(plan.nb, plan.hi_mask, epi)) ? 8 : kSmallWaves;   // channels (waves) per workgroup
  dim3 grid((unsigned)((a.n_steps + spw - 1) / spw), (unsigned)((p.n_ch + wpb - 1) / wpb));
  a.dbg = nullptr;
  static const bool dbg_clock = getenv("ACDSP_DEBUG_CLOCK") != nullptr;
  const size_t n_waves = (size_t)grid.x * grid.y * wpb;
  if (dbg_clock) { if (hipMalloc((void **)&a.dbg, n_waves * 48) != hipSuccess) { a.dbg = nullptr; } else { (void)hipMemsetAsync(a.dbg, 0, n_waves * 48, s); } }
  hipError_t rc = launch_switch(p, plan.nb, d_frag, a, epi, grid, s);
  if (a.dbg) {
    std::vector<int64_t> hd(2 * n_waves);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(hd.data(), a.dbg, n_waves * 16, hipMemcpyDeviceToHost);
    double sc = 0, sr = 0;
    for (size_t i = 0; i < n_waves; i++) { sc += (double)hd[2 * i]; sr += (double)hd[2 * i + 1]; }
    fprintf(stderr, "[acdsp] fir_mfma: %zu waves, mean wave life %.1f us, shader clock %.3f GHz (spw %lld)\n", n_waves,
            sr / n_waves / 100.0, sc / sr * 0.1, (long long)spw);
#ifdef ACDSP_X_PHASES
    {
      std::vector<int64_t> ph(4 * n_waves);
      (void)hipMemcpy(ph.data(), a.dbg + 2 * n_waves, n_waves * 32, hipMemcpyDeviceToHost);
      double t[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < n_waves; i++) { for (int q = 0; q < 4; q++) { t[q] += (double)ph[4 * i + q]; } }
      const double steps = (double)n_waves * (double)spw;
      fprintf(stderr, "[acdsp] cycles per step: M %.0f | barrier %.0f | O %.0f | barrier %.0f\n", t[0] / steps, t[1] / steps, t[2] / steps, t[3] / steps);
    }
#endif
    (void)hipFree(a.dbg);
  }
  return rc;
}

static bool use_reg33(int nb, uint64_t hi_mask, int epi) {
  static const bool off = getenv("ACDSP_NO_REG33") != nullptr;   // A/B knob: LDS-resident fragments (fir_mfma_big2_kernel) for NB = 33 too
  return nb == 33 && !off && reg33_band_code(hi_mask, epi) != 0;
}

// NB = 10 .. 31 (258 - 961 taps): the register-resident kernel at one wave per SIMD, like NB = 33, for sets whose high-byte band fits the
// five central K-blocks of the instantiated shape.  Shapes exist for odd NB (11 .. 31: fir_mfma_mid.hip, _mid2, _mid3, translation
// units of their own for the compile time); fir_mfma_plan_blocks pads an even plan by one leading zero block (2 MFMAs per step).  Same-box A/B at 319
// taps: 1.27 -> 1.12 ms (profiles/r3_taps_sweep.txt); dense sets and wider bands stay on the LDS-resident kernels.
static int mid_band_code(int nb, uint64_t hi_mask, int epi) {
  if ((epi != 1 && epi != 2) || nb < 11 || nb > 31 || (nb & 1) == 0) { return 0; }
  const int sk = (nb - 5) / 2;
  int lo = 0, hi = 0;
  if (hi_mask == 0) { lo = hi = nb; }
  else {
    while (lo < nb && !((hi_mask >> lo) & 1)) { lo++; }
    while (hi < nb && !((hi_mask >> (nb - 1 - hi)) & 1)) { hi++; }
  }
  return (lo >= sk && hi >= sk) ? sk + 16 * sk : 0;
}
static bool use_mid(int nb, uint64_t hi_mask, int epi) {
  static const bool off = getenv("ACDSP_NO_MID") != nullptr;   // A/B knob: LDS-resident fragments (fir_mfma_big2_kernel)
  return !off && mid_band_code(nb, hi_mask, epi) != 0;
}

static hipError_t launch_switch(const FirParams &p, int nb, const uint32_t *d_frag, const MfmaArgs &a, int epi, dim3 grid, hipStream_t s) {
  if (use_reg33(nb, a.hi_mask, epi)) { return launch_nb33(p, d_frag, a, epi, grid, s); }
  if (use_mid(nb, a.hi_mask, epi)) {
    return nb <= 17 ? launch_fir_mfma_mid(p, nb, d_frag, a, epi, grid, s)
                    : (nb <= 25 ? launch_fir_mfma_mid2(p, nb, d_frag, a, epi, grid, s) : launch_fir_mfma_mid3(p, nb, d_frag, a, epi, grid, s));
  }
  if (nb > kMaxRegNB) { return launch_big(p, d_frag, a, epi, grid, s); }
  switch (nb) {
#ifndef ACDSP_FIR_DEV_NB9   // development builds: only the 255-tap shape (compile time 3 min -> 35 s)
    case 1: return launch_nb<1>(p, d_frag, a, epi, grid, s);
    case 2: return launch_nb<2>(p, d_frag, a, epi, grid, s);
    case 3: return launch_nb<3>(p, d_frag, a, epi, grid, s);
    case 4: return launch_nb<4>(p, d_frag, a, epi, grid, s);
    case 5: return launch_nb<5>(p, d_frag, a, epi, grid, s);
    case 6: return launch_nb<6>(p, d_frag, a, epi, grid, s);
    case 7: return launch_nb<7>(p, d_frag, a, epi, grid, s);
    case 8: return launch_nb<8>(p, d_frag, a, epi, grid, s);
#endif
    case 9: return launch_nb<9>(p, d_frag, a, epi, grid, s);
    default: return hipErrorInvalidValue;
  }
}

#endif  // !ACDSP_FIR_TU_MID

}  // namespace acdsp
