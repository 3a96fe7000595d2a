// cic2.hip -- ac_cic_dec_full at the rates it is deployed at (R = 32 ... 256): two stages in one launch.
//
// What it replaces: ac_cic_full_core_intg::intStage / decIntgCore (reference include/ac_dsp/ac_cic_full_core.h:80-87,110-135) and
// ac_cic_full_core_diff::comb / diffStage (:228-255), driven by ac_cic_dec_full::run (ac_cic_dec_full.h:187-222).
//
// The decimator is, end to end (cic.hip, tests/test_oracle.py::test_cic_closed_form_fir_identity),
//     y[j] = (H * x)[first + j R]  mod 2^W_int,   H(z) = z^-(N-1) (1 + z^-1 + ... + z^-(R M' - 1))^N,   M' = min(M, 2).
// fir_gen.hip evaluates that FIR directly on the matrix cores; its Toeplitz tile spans 15 R + N R M' taps, which stops fitting at
// R ~ 20.  With R = R1 R2 the boxcar factors exactly,
//     (1 - z^-(R M')) / (1 - z^-1)  =  (1 - z^-R1) / (1 - z^-1)  *  (1 - w^-(R2 M')) / (1 - w^-1),   w = z^R1,
// i.e.  H(z) = [z^-(N-1) boxcar(R1)^N](z) * [boxcar(R2 M')^N](z^R1): a CIC of rate R1 (M = 1) followed by a CIC of rate R2 (M = M')
// running on the first one's outputs (noble identity).  So:
//   stage 1  u[m] = (h1 * x)[first + m R1],  h1 = z^-(N-1) boxcar(R1)^N  -- the ring kernel's strided-Toeplitz form on the matrix cores
//            (fir_gen.hip: byte planes in an LDS ring of two steps, every input byte loaded once, 256 u per step);
//   stage 2  y[j] = sum_i (-1)^i C(N, i) s_N[(j - i M') R2],  s_N = the N-fold running sum of u  -- N integrators at the u rate, the combs
//            at the output rate, all mod 2^64.  The N integrators of a step are N cascaded prefix sums over the wave: the step's 256 u
//            pass through a 2 KB LDS tile into time order (four consecutive u per lane), each level is three in-lane adds, one
//            64-lane DPP scan of the lane totals (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) and four adds of the lane's prefix; the
//            running totals are wave-uniform.  Against cic_kernel's 2 N wide adds per INPUT sample and channel that is ~30 N VALU
//            instructions per 256 R1 input samples.
// Only the decimated values s_N[j R2] are kept (an LDS ring of one chunk); the chunk ends with the comb differences, the OUT_TYPE
// conversion and fully coalesced stores (consecutive lanes = consecutive outputs).
//
// Chunks carry no state: a chunk starts `wu` steps early from ZERO integrators and an empty comb line.  The difference between the true
// integrator state and zero is a polynomial of degree < N in the u index, which the N combs annihilate -- equivalently y is an FIR of
// N (R2 M' - 1) + 1 taps on u, so 256 wu >= N (R2 M' - 1) warm-up values reproduce every output of the chunk exactly (the same argument
// that replaces the reference's registers by an input history between run() calls).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "cic_kernels.hpp"
#include "fir_kernels.hpp"

namespace acdsp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef long v2l __attribute__((ext_vector_type(2)));

constexpr int kCic2MaxN = 6;       // (R M)^N < 2^31 with R >= 32 leaves N <= 6
constexpr int kCic2Zero = 16;      // zero entries in front of the decimated ring: the comb line of a chunk starts empty (N M' <= 12)

struct Cic2Args {
  FirGenPlan pl;                   // stage 1: taps z^-(N-1) boxcar(R1)^N, decimation R1, window offset for first % 16
  int32_t n_ch, N, me, R2, w_int;
  int32_t in_F;
  DFmt out;
  int32_t out_eb, out_simple;      // OUT_TYPE conversion as in cic_kernel (2: OUT holds INT_TYPE, 1: same fraction + AC_WRAP, 0: general)
  int32_t hl, ring_delta;
  int32_t nst, wu;                 // steps per chunk (warm-up included), warm-up steps
  int32_t full_chunks, nst_last;   // chunks [0, full_chunks) run nst steps; one more chunk of nst_last steps may follow (the end of the call)
  int32_t pw, xcd_map;
  int32_t dbg;                     // ACDSP_CIC2_DBG (timing ablations only, results wrong; profiles/r6_cic2_ablation.txt): 1 no stage 2, 8 no comb / conversion / stores, 16 outputs into LDS
  uint32_t rcp2;                   // ceil(2^32 / R2)
  int64_t corr;                    // re-bias correction of the unsigned input planes (fir_gen.hip)
  int64_t first, n_out, n16;
  int64_t in_stride, out_stride;
  const void *x; void *y; const void *hist;
};

// x[lane] += x[lane - d] inside rows of 16 lanes (zero beyond the row), then across the rows: a 64-lane inclusive prefix sum of 64-bit words
template <int CTRL, int RM>
__device__ __forceinline__ uint64_t dpp_add(uint64_t x) {
  const v2u v = __builtin_bit_cast(v2u, x);
  v2u s;
  s.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, RM, 0xf, true);
  s.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, RM, 0xf, true);
  return x + __builtin_bit_cast(uint64_t, s);
}
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t x) {
  x = dpp_add<0x111, 0xf>(x);   // row_shr:1
  x = dpp_add<0x112, 0xf>(x);   // row_shr:2
  x = dpp_add<0x114, 0xf>(x);   // row_shr:4
  x = dpp_add<0x118, 0xf>(x);   // row_shr:8
  x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
  const v2u v = __builtin_bit_cast(v2u, x);
  v2u s;
  s.x = (unsigned)__builtin_amdgcn_readlane((int)v.x, l);
  s.y = (unsigned)__builtin_amdgcn_readlane((int)v.y, l);
  return __builtin_bit_cast(uint64_t, s);
}

__device__ __forceinline__ unsigned c2_gather4(unsigned d0, unsigned d1, unsigned d2, unsigned d3, int p) {
  const unsigned sel = 0x0c0c0400u + 0x0101u * (unsigned)p;
  const unsigned lo = __builtin_amdgcn_perm(d1, d0, sel);
  const unsigned hi = __builtin_amdgcn_perm(d3, d2, sel);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// One wave = one channel x a chunk of `nst` steps (the first `wu` of them warm-up); a step = 256 stage-1 outputs = 256 R1 inputs.
// Ring geometry, slot map and load / staging order are fir_gen_ring_kernel's (fir_gen.hip), rolled into a loop over batches of G steps
// (load groups alternate between two register sets, each fetched two groups ahead; a group is one step, or two where a step advances
// half a 1 KB load: M = 2).  Stage 2 runs once per batch: the G x 256 u of a batch wait in an LDS tile and come back in time order, 4 G
// consecutive u per lane, so a prefix-sum level costs 8 G - 1 adds and ONE wave scan per G steps.
//   tile: u index t of the batch at byte 8 t + 16 (t >> 5) -- 16 bytes of padding per 256 keep the 32 G-byte lane rows of the reads and the
//         32-byte pieces of the writes on distinct bank groups
//   ring: [16 history][decimated values of the batch] as 64-bit words; the last 16 move to the front after every batch
template <typename TIN, int PCT, int NBT, int R1, int G, int NN>
__global__ void __launch_bounds__(64, 2) cic2_kernel(Cic2Args a, const v4i *__restrict__ frag) {
  constexpr int S = (int)sizeof(TIN), PX = S;
  constexpr int LS = 8 / S;
  constexpr int ADV = 16 * R1, NSL = 15 * R1 + 4 * NBT;
  constexpr int H = (LS - 1 + NSL - ADV + LS - 1) / LS * LS;
  constexpr int M = (R1 * S) % 4 == 0 ? 1 : 2;
  constexpr int NLD = R1 * S * M / 4;
  constexpr int PPB = 16 / S;
  constexpr int SPK = 64 / S;
  static_assert(S == 2 || S == 4, "2- and 4-byte samples");
  static_assert(G % (2 * M) == 0 && G <= 4, "a batch holds whole pairs of load groups");
  constexpr bool HOLES = (R1 % 2 == 0 && SPK % R1 == 0);
  static_assert(H <= ADV && S * H <= 64 && NLD * SPK == M * ADV && NLD >= 1, "ring geometry");
  constexpr int KSTEP = HOLES ? (SPK + 2 * (SPK / R1)) * 16 : SPK * 16;
  constexpr int PADV = HOLES ? (ADV + 2 * (ADV / R1)) * 16 : ADV * 16;
  constexpr int RING = M * ADV + H;                       // ONE load group (fir_gen_ring_kernel keeps two steps): a wave stages, multiplies and only then moves the group's tail into the halo
  constexpr int PH = HOLES ? (R1 - H % R1) % R1 : 0;
  constexpr int PLANE = HOLES ? (RING + 2 * ((RING + PH) / R1) + 2) * 16 : (RING + 2) * 16;
  constexpr int DUMP = PLANE - 16;
  constexpr int EPL = 4 * G;                              // u per lane and batch
  constexpr int TILE = G * (2048 + 128);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [PX][PLANE] byte planes, the batch tile, the decimated ring
  unsigned char *const tile = lds + PX * PLANE;
  uint64_t *const wr = (uint64_t *)(tile + TILE);
  const int lane = threadIdx.x;
  const int n_col = lane & 15, kg = lane >> 4;
  int bx, ch;
  xcd_remap(a.xcd_map, bx, ch);
  const int NB = a.pl.nb, PC = a.pl.pc;
  auto phys = [](int s) { return HOLES ? s + 2 * ((s + PH) / R1) : s; };
  auto taddr = [](int t) { return 8 * t + 16 * (t >> 5); };

  v4i A[NBT][PCT];
#pragma unroll
  for (int b = 0; b < NBT; b++) {
#pragma unroll
    for (int q = 0; q < PCT; q++) { A[b][q] = (b < NB && q < PC) ? frag[((size_t)q * NB + b) * 64 + lane] : (v4i){0, 0, 0, 0}; }
  }
  const TIN *xrow = (const TIN *)a.x + (int64_t)ch * a.in_stride;
  const TIN *hrow = (const TIN *)a.hist + (int64_t)ch * a.hl + a.hl;
  const int nst = bx < a.full_chunks ? a.nst : a.nst_last, wu = a.wu, R2 = a.R2;
  const int64_t s0 = (int64_t)bx * (a.nst - wu) - wu;                  // first step of the chunk, warm-up included (chunk 0: negative -> history)
  const int64_t c0 = a.first - a.pl.off - 16 * (int64_t)a.ring_delta + s0 * (256 * R1);
  const bool interior = c0 >= 0 && c0 + 16 * (int64_t)(H + nst * ADV) <= a.n16;

  const int pl_lane = lane < S * H ? lane : S * H - 1;
  const int pr_off = phys(pl_lane / S) * 16 + (pl_lane % S) * PPB;
  const int st_base = phys(H + lane / S) * 16 + (lane % S) * PPB;
  const int mir_off = lane >= 64 - S * H ? st_base - KSTEP : DUMP;
  int xs[NBT];
#pragma unroll
  for (int b = 0; b < NBT; b++) { xs[b] = phys(a.ring_delta + R1 * n_col + 4 * b + kg) * 16; }
  const int tw_off = taddr(16 * n_col + 4 * kg);                       // this lane's four u of a step (32 bytes inside one padding block)
  const int tr_off = taddr(EPL * lane);                                // ... and its 4 G consecutive u of a batch

  // ---- stage 2 bookkeeping (u index m = 256 step + ...; output j sits at m = j R2) ----
  // The chunk's first u index is 256 s0; 512 R2 is added so that the division below is one of non-negative numbers (s0 >= -wu >= -2).
  const unsigned mo = (unsigned)(s0 * 256 + 512 * (int64_t)R2);
  const unsigned Q0 = mo / (unsigned)R2, mb0 = mo - Q0 * (unsigned)R2;        // once per chunk
  const unsigned tA = mb0 + 256u * (unsigned)wu + (unsigned)R2 - 1, tB = mb0 + 256u * (unsigned)nst + (unsigned)R2 - 1;
  const int64_t jA = (int64_t)Q0 - 512 + __umulhi(tA, a.rcp2);          // outputs the chunk owns: ceil(u index of its first main step / R2) ...
  int64_t jB = (int64_t)Q0 - 512 + __umulhi(tB, a.rcp2);                //   ... up to the same of the next chunk's
  if (jB > a.n_out) { jB = a.n_out; }
  if (lane < kCic2Zero) { wr[lane] = 0; }                               // the comb line of a chunk starts empty
  uint64_t cy[NN];
#pragma unroll
  for (int l = 0; l < NN; l++) { cy[l] = 0; }

  // piece at sample offset tp (wave-uniform) + lp (this lane's) from C0; interior chunks: uniform base pointer + 32-bit lane offset
  const char *const xb = (const char *)(xrow + c0);
  auto piece = [&](int64_t tp, int lp, auto fast_c) __attribute__((always_inline)) -> v4i {
    if constexpr (decltype(fast_c)::value) {
      return __builtin_nontemporal_load((const v4i *)(xb + tp * S + (size_t)(unsigned)(lp * S)));
    } else {
      const int64_t t = c0 + tp + lp;
      const int64_t th = t < -(int64_t)a.hl ? -(int64_t)a.hl : t;      // before the history: slots below the first window, never read
      const TIN *src = (t < 0) ? hrow + th : xrow + ((t < a.n16) ? t : 0);
      return __builtin_nontemporal_load((const v4i *)src);
    }
  };
  auto stage_piece = [&](const v4i &v, int off) __attribute__((always_inline)) {
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      if constexpr (S == 2) {
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        unsigned lo = __builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, sel), hi = __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, sel);
        if (pp < PX - 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
        *(v2u *)(lds + pp * PLANE + off) = (v2u){lo, hi};
      } else {
        unsigned w = c2_gather4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w, pp);
        if (pp < PX - 1) { w ^= 0x80808080u; }
        *(unsigned *)(lds + pp * PLANE + off) = w;
      }
    }
  };

  // stage 1 of one step: 256 outputs on the matrix cores into slot s of the batch tile (par: the ring half the step's window starts in)
  auto stage1 = [&](int par, int s) __attribute__((always_inline)) {
    v4i acc[PX + PCT - 1];
#pragma unroll
    for (int w = 0; w < PX + PCT - 1; w++) { acc[w] = (v4i){0, 0, 0, 0}; }
#pragma unroll
    for (int b = 0; b < NBT; b++) {
      v4i X[PX];
#pragma unroll
      for (int pp = 0; pp < PX; pp++) { X[pp] = *(const v4i *)(lds + pp * PLANE + xs[b] + par); }
#pragma unroll
      for (int q = 0; q < PCT; q++) {
#pragma unroll
        for (int pp = 0; pp < PX; pp++) { acc[pp + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[b][q], X[pp], acc[pp + q], 0, 0, 0); }
      }
    }
    // lane (n_col, kg) holds u[16 n_col + 4 kg + r]: recombine the plane accumulators mod 2^64 (fir_gen_ring_kernel's two forms)
    uint64_t u[4];
    constexpr int NACC = PX + PCT - 1, NPR = (NACC + 1) / 2;
    if (a.pw) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int pr[NPR];
#pragma unroll
        for (int m = 0; m < NPR; m++) { pr[m] = (2 * m + 1 < NACC) ? (int)(((unsigned)acc[2 * m + 1][r] << 8) + (unsigned)acc[2 * m][r]) : acc[2 * m][r]; }
        unsigned lo = (unsigned)a.corr, hi = (unsigned)((uint64_t)a.corr >> 32);
        { const unsigned t0 = lo + (unsigned)pr[0]; hi += (unsigned)(pr[0] >> 31) + (t0 < lo); lo = t0; }
        if constexpr (NPR > 1) { const unsigned t1 = (unsigned)pr[1] << 16, t2 = lo + t1; hi += (unsigned)(pr[1] >> 16) + (t2 < lo); lo = t2; }
        if constexpr (NPR > 2) { hi += (unsigned)pr[2]; }
        if constexpr (NPR > 3) { hi += (unsigned)pr[3] << 16; }
        u[r] = ((uint64_t)hi << 32) | lo;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint64_t y = (uint64_t)a.corr;
#pragma unroll
        for (int w = 0; w < NACC; w++) { y += (uint64_t)(int64_t)acc[w][r] << (8 * w); }
        u[r] = y;
      }
    }
    unsigned char *tp = tile + s * (2048 + 128) + tw_off;
    *(v2l *)tp = (v2l){(long)u[0], (long)u[1]};
    *(v2l *)(tp + 16) = (v2l){(long)u[2], (long)u[3]};
  };

  // stage 2 of one batch (steps k0 .. k0 + G - 1 of the chunk): integrators, decimation, combs, outputs
  auto stage2 = [&](int k0) __attribute__((always_inline)) {
    // into time order: 4 G consecutive u per lane (single-wave workgroup: LDS operations execute in order, no barrier)
    uint64_t e[EPL];
#pragma unroll
    for (int i = 0; i < EPL / 2; i++) {
      const v2l v = *(const v2l *)(tile + tr_off + 16 * i);
      e[2 * i] = (uint64_t)v.x; e[2 * i + 1] = (uint64_t)v.y;
    }
    // intStage x N at the u rate (ac_cic_full_core.h:80-87 after the R1 factor has been taken out): running sums mod 2^64
#pragma unroll
    for (int l = 0; l < NN; l++) {
#pragma unroll
      for (int i = 1; i < EPL; i++) { e[i] += e[i - 1]; }
      const uint64_t t = wave_incl_scan(e[EPL - 1]);
      const uint64_t ex = t - e[EPL - 1] + cy[l];
#pragma unroll
      for (int i = 0; i < EPL; i++) { e[i] += ex; }
      cy[l] += readlane64(t, 63);
    }
#pragma unroll
    for (int i = 0; i < EPL / 2; i++) { *(v2l *)(tile + tr_off + 16 * i) = (v2l){(long)e[2 * i], (long)e[2 * i + 1]}; }
    // valid = (rate_cnt == 0) (:116-133) at the u rate: the batch's u indices that are 0 mod R2
    const unsigned tb = mb0 + 256u * (unsigned)k0;
    const unsigned qb = __umulhi(tb, a.rcp2), rb = tb - qb * (unsigned)R2;
    const int t0 = rb == 0 ? 0 : R2 - (int)rb;                           // first of them inside the batch
    const int nd = (int)__umulhi((unsigned)(256 * G - t0 + R2 - 1), a.rcp2);
    const int64_t j0 = (int64_t)Q0 - 512 + qb + (rb != 0);               // its output index
    for (int idx = lane; idx < nd; idx += 64) {
      wr[kCic2Zero + idx] = *(const uint64_t *)(tile + taddr(t0 + idx * R2));
    }
    // comb x N at the output rate (ac_cic_full_core.h:228-255, differential delay M' = min(M, 2)), OUT_TYPE conversion, coalesced stores:
    // outputs [i_lo, i_hi) of the batch belong to the chunk (everything but the warm-up), consecutive lanes = consecutive outputs
    const int64_t dlo = jA - j0, dhi = jB - j0;
    const int i_lo = dlo > 0 ? (dlo < nd ? (int)dlo : nd) : 0, i_hi = dhi < nd ? (dhi > 0 ? (int)dhi : 0) : nd;
    auto comb = [&](int idx) __attribute__((always_inline)) -> uint64_t {
      // (1 - z^-M')^N: binomial coefficients of a compile-time N
      uint64_t v = 0;
      unsigned c = 1;
#pragma unroll
      for (int i = 0; i <= NN; i++) {
        const uint64_t t = (uint64_t)c * wr[kCic2Zero + idx - i * a.me];
        v = (i & 1) ? v - t : v + t;
        c = c * (unsigned)(NN - i) / (unsigned)(i + 1);
      }
      return v;
    };
    // (straight-line code, NIT predicated passes of 64 outputs: a store inside a loop of unknown trip count makes the compiler drain
    //  vmcnt -- and with it the prefetched loads -- at the head of the batch loop: 2.15 instead of 1.5 ms per 8.6 GB in the first form)
    constexpr int NIT = (256 * G / 2 + 63) / 64;
    if (i_lo < i_hi && !(a.dbg & 8)) {
      // OUT_TYPE with INT_TYPE's fraction and AC_WRAP (out_simple): bit-field wraps, i.e. shift pairs.  Any other OUT_TYPE: the general
      // conversion runs in a rolled loop into the (now free) tile, and the stores below pick the finished words up from there.
      const int sh1 = 64 - a.w_int, sh2 = (a.out_simple == 2 || a.out.W >= 64) ? 0 : 64 - a.out.W;
      int64_t *const cv = (int64_t *)tile;
      if (!a.out_simple) {
        for (int idx = i_lo + lane; idx < i_hi; idx += 64) { cv[idx] = requant64(wrap64((int64_t)comb(idx), a.w_int, 1), a.in_F, a.out); }
      }
      auto emit = [&](auto *yp) __attribute__((always_inline)) {
        typedef typename std::remove_pointer<decltype(yp)>::type OT;
        OT *yrow = yp + (int64_t)ch * a.out_stride + j0;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int idx = i_lo + lane + 64 * it;
          if (idx < i_hi) {
            int64_t val;
            if (a.out_simple) {
              val = (int64_t)(comb(idx) << sh1) >> sh1;
              if (sh2) { val = a.out.S ? (int64_t)((uint64_t)val << sh2) >> sh2 : (int64_t)(((uint64_t)val << sh2) >> sh2); }
            } else {
              val = cv[idx];
            }
            if (a.dbg & 16) { cv[idx] = val; } else { __builtin_nontemporal_store((OT)val, yrow + idx); }
          }
        }
      };
      if (a.out_eb == 8) { emit((int64_t *)a.y); } else if (a.out_eb == 4) { emit((int32_t *)a.y); } else { emit((int16_t *)a.y); }
    }
    // the newest 16 decimated values are the next batch's history
    if (lane < kCic2Zero) {
      const uint64_t hst = wr[nd + lane];
      wr[lane] = hst;
    }
  };

  auto chunk = [&](auto fast_c) __attribute__((always_inline)) {
    v4i pre[2][NLD];
    auto fetch = [&](int g, int set) __attribute__((always_inline)) {      // the 16 R1 M new slots of load group g
#pragma unroll
      for (int k = 0; k < NLD; k++) { pre[set][k] = piece(16 * (int64_t)(H + (int64_t)g * M * ADV) + 64 * k * PPB, lane * PPB, fast_c); }
    };
    const int ng = nst / M;                 // load groups of the chunk (host: nst is a multiple of G)
    const v4i prm = piece(0, pl_lane * PPB, fast_c);
    fetch(0, 0);
    fetch(1, 1);
    stage_piece(prm, pr_off);
    for (int k0 = 0; k0 < nst; k0 += G) {
#pragma unroll
      for (int s = 0; s < G; s += M) {
        const int gg = (k0 + s) / M, set = (s / M) & 1;
        const int gn = gg + 2 < ng ? gg + 2 : gg;   // the last groups re-fetch themselves: no branch in the loop
        if constexpr (M == 1) {
#pragma unroll
          for (int q = 0; q < NLD; q++) { stage_piece(pre[set][q], st_base + q * KSTEP); }
          const v4i tail = pre[set][NLD - 1];
          fetch(gn, set);
          asm volatile("" ::: "memory");
          stage1(0, s);
          stage_piece(tail, mir_off);                                   // the step's last H slots are the next step's halo (behind this step's fragment reads)
        } else {
#pragma unroll
          for (int q = 0; q < NLD; q++) { stage_piece(pre[set][q], st_base + q * KSTEP); }
          asm volatile("" ::: "memory");
          stage1(0, s);
          stage_piece(pre[set][NLD - 1], mir_off);
          fetch(gn, set);
          asm volatile("" ::: "memory");
          stage1(PADV, s + 1);
        }
      }
      if (!(a.dbg & 1)) { stage2(k0); }
    }
  };
  if (interior) { chunk(std::integral_constant<bool, true>()); } else { chunk(std::integral_constant<bool, false>()); }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
namespace {

template <typename TIN, int PCT, int NBT, int R1>
struct Cic2Geom {
  static constexpr int S = (int)sizeof(TIN), LS = 8 / S, ADV = 16 * R1, NSL = 15 * R1 + 4 * NBT;
  static constexpr int H = (LS - 1 + NSL - ADV + LS - 1) / LS * LS;
  static constexpr int SPK = 64 / S;
  static constexpr bool HOLES = (R1 % 2 == 0 && SPK % R1 == 0);
  static constexpr int M = (R1 * S) % 4 == 0 ? 1 : 2;
  static constexpr int RING = M * ADV + H;
  static constexpr int PH = HOLES ? (R1 - H % R1) % R1 : 0;
  static constexpr int PLANE = HOLES ? (RING + 2 * ((RING + PH) / R1) + 2) * 16 : (RING + 2) * 16;
};

template <typename TIN, int PCT, int NBT, int R1, int G, int NN>
hipError_t launch_shape(dim3 grid, hipStream_t s, const Cic2Args &a, const v4i *frag) {
  typedef Cic2Geom<TIN, PCT, NBT, R1> Ge;
  const size_t ring_bytes = (size_t)(kCic2Zero + (256 * G + a.R2 - 1) / a.R2 + 2) * 8;   // history + the decimated values of one batch
  const size_t lds = (size_t)Ge::S * Ge::PLANE + (size_t)G * (2048 + 128) + ring_bytes;
  if (lds > 65536) { return hipErrorInvalidValue; }
  hipLaunchKernelGGL((cic2_kernel<TIN, PCT, NBT, R1, G, NN>), grid, dim3(64), lds, s, a, frag);
  return hipGetLastError();
}
// N = integrator / comb stages (compile-time: the prefix-sum levels, their wave-uniform running totals and the comb's binomials)
template <typename TIN, int PCT, int NBT, int R1, int G>
hipError_t launch_n(int n, dim3 grid, hipStream_t s, const Cic2Args &a, const v4i *frag) {
  switch (n) {
    case 1: return launch_shape<TIN, PCT, NBT, R1, G, 1>(grid, s, a, frag);
    case 2: return launch_shape<TIN, PCT, NBT, R1, G, 2>(grid, s, a, frag);
    case 3: return launch_shape<TIN, PCT, NBT, R1, G, 3>(grid, s, a, frag);
    case 4: return launch_shape<TIN, PCT, NBT, R1, G, 4>(grid, s, a, frag);
    case 5: return launch_shape<TIN, PCT, NBT, R1, G, 5>(grid, s, a, frag);
    case 6: if constexpr (R1 != 15) { return launch_shape<TIN, PCT, NBT, R1, G, 6>(grid, s, a, frag); } else { return hipErrorInvalidValue; }   // (R1 = 15: its taps need a third digit plane at N = 6)
    default: return hipErrorInvalidValue;
  }
}
// G = steps per stage-2 batch: two where a load group is one step (G = 4 measured equal -- 2.144 / 2.125 ms, profiles/r6_cic2_ablation.txt -- and its
// tile is LDS the occupancy pays for), four where a group is two steps
template <typename TIN, int PCT, int NBT, int R1>
hipError_t launch_g(int g, int n, dim3 grid, hipStream_t s, const Cic2Args &a, const v4i *frag) {
  constexpr int G = Cic2Geom<TIN, PCT, NBT, R1>::M == 1 ? 2 : 4;
  if (g != G) { return hipErrorInvalidValue; }
  return launch_n<TIN, PCT, NBT, R1, G>(n, grid, s, a, frag);
}

}  // namespace

// Compiled stage-1 shapes: (translation unit, sample type, tag, container bytes, R1, digit planes, K blocks) -- planes / blocks cover
// z^-(N-1) boxcar(R1)^N for every N <= 6 (R1 = 15: N <= 5 -- fifteen 1 KB loads per register set leave no room for a third digit plane;
// launch_cic2 declines a plan that does not fit and the call stays on the recurrence kernel).  The order is the order of preference where several rates divide R: the most input bytes per
// step first (a step's fixed work -- fragment reads, recombination, the prefix-sum levels -- is per 256 u, whatever R1).
// Translation units (Makefile): cic2.hip = unit 0 + the host side; cic2_b / _c / _d.hip re-include this file with ACDSP_CIC2_PART set.
#define ACDSP_CIC2_SHAPES(X)                                                                                       \
  X(0, int16_t, s16_r16, 2, 16, 3, 6) X(4, int16_t, s16_r12, 2, 12, 3, 4) X(1, int16_t, s16_r10, 2, 10, 3, 4)        \
  X(1, int16_t, s16_r8, 2, 8, 2, 3) X(4, int16_t, s16_r6, 2, 6, 2, 2) X(5, int16_t, s16_r4, 2, 4, 2, 2)              \
  X(3, int16_t, s16_r15, 2, 15, 2, 5) X(4, int16_t, s16_r7, 2, 7, 2, 3) X(3, int16_t, s16_r5, 2, 5, 2, 2)            \
  X(5, int16_t, s16_r3, 2, 3, 2, 2)                                                                                  \
  X(2, int32_t, s32_r10, 4, 10, 3, 4) X(2, int32_t, s32_r8, 4, 8, 2, 3) X(5, int32_t, s32_r7, 4, 7, 2, 3)            \
  X(5, int32_t, s32_r6, 4, 6, 2, 2) X(3, int32_t, s32_r5, 4, 5, 2, 2) X(3, int32_t, s32_r4, 4, 4, 2, 2)              \
  X(3, int32_t, s32_r3, 4, 3, 2, 2)
#ifndef ACDSP_CIC2_PART
#define ACDSP_CIC2_PART 0
#endif
#define ACDSP_CIC2_DECL(PART, TIN, TAG, EB, R1V, PCTV, NBTV) \
  hipError_t cic2_launch_##TAG(int g, int n, dim3 grid, hipStream_t s, const Cic2Args &a, const v4i *frag);
ACDSP_CIC2_SHAPES(ACDSP_CIC2_DECL)
#undef ACDSP_CIC2_DECL
#define ACDSP_CIC2_DEF(PART, TIN, TAG, EB, R1V, PCTV, NBTV)                                                         \
  ACDSP_CIC2_IF_##PART(hipError_t cic2_launch_##TAG(int g, int n, dim3 grid, hipStream_t s, const Cic2Args &a, const v4i *frag) { \
    return launch_g<TIN, PCTV, NBTV, R1V>(g, n, grid, s, a, frag);                                                  \
  })
#define ACDSP_CIC2_KEEP(...) __VA_ARGS__
#define ACDSP_CIC2_DROP(...)
#if ACDSP_CIC2_PART == 0
#define ACDSP_CIC2_IF_0 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_0 ACDSP_CIC2_DROP
#endif
#if ACDSP_CIC2_PART == 1
#define ACDSP_CIC2_IF_1 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_1 ACDSP_CIC2_DROP
#endif
#if ACDSP_CIC2_PART == 2
#define ACDSP_CIC2_IF_2 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_2 ACDSP_CIC2_DROP
#endif
#if ACDSP_CIC2_PART == 3
#define ACDSP_CIC2_IF_3 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_3 ACDSP_CIC2_DROP
#endif
#if ACDSP_CIC2_PART == 4
#define ACDSP_CIC2_IF_4 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_4 ACDSP_CIC2_DROP
#endif
#if ACDSP_CIC2_PART == 5
#define ACDSP_CIC2_IF_5 ACDSP_CIC2_KEEP
#else
#define ACDSP_CIC2_IF_5 ACDSP_CIC2_DROP
#endif
ACDSP_CIC2_SHAPES(ACDSP_CIC2_DEF)
#undef ACDSP_CIC2_DEF

#if ACDSP_CIC2_PART == 0
namespace {

// compiled shapes: (container bytes, R1) -> digit planes / K blocks compiled in, steps per load group
struct Shape { int in_eb, R1, pct, nbt, m; hipError_t (*launch)(int, int, dim3, hipStream_t, const Cic2Args &, const v4i *); };
#define ACDSP_CIC2_ROW(PART, TIN, TAG, EB, R1V, PCTV, NBTV) {EB, R1V, PCTV, NBTV, ((R1V) * (EB)) % 4 == 0 ? 1 : 2, cic2_launch_##TAG},
const Shape kShapes[] = {ACDSP_CIC2_SHAPES(ACDSP_CIC2_ROW)};
#undef ACDSP_CIC2_ROW

const Shape *find_shape(int in_eb, int R1) {
  for (const Shape &s : kShapes) { if (s.in_eb == in_eb && s.R1 == R1) { return &s; } }
  return nullptr;
}

}  // namespace

// Which factorisation R = R1 R2 (R1 a compiled stage-1 rate of this container width) serves the parameter set; false: none.
bool cic2_factor(int in_eb, int R, int me, int N, int *R1_out, int *R2_out, int *wu_out) {
  if (N < 1 || N > kCic2MaxN || N * me > kCic2Zero) { return false; }
  static const char *force = getenv("ACDSP_CIC2_R1");   // A/B knob: the stage-1 rate to use where it divides R
  for (const Shape &s : kShapes) {
    if (s.in_eb != in_eb || R % s.R1 != 0 || R / s.R1 < 2) { continue; }
    if (force && atoi(force) != s.R1) { continue; }
    const int R2 = R / s.R1;
    const int wu = (N * (R2 * me - 1) + 255) / 256;
    if (wu < 1 || wu > 2) { continue; }
    // the stage-1 taps of this rate must fit the digit planes / K blocks the shape was compiled with, at both extreme phases of a call start
    // (R1 = 15 at N = 6 needs a third plane: R = 45 then goes to R1 = 5, not to the recurrence kernel)
    std::vector<int64_t> taps;
    cic2_stage1_taps(s.R1, N, &taps);
    bool fits = true;
    for (int ph : {0, 15}) {
      FirGenPlan probe;
      std::vector<uint32_t> fr;
      if (!fir_gen_plan(taps.data(), (int)taps.size(), s.R1, ph, &probe, &fr) || probe.pc > s.pct || probe.nb > s.nbt) { fits = false; }
    }
    if (!fits) { continue; }
    *R1_out = s.R1; *R2_out = R2; *wu_out = wu;
    return true;
  }
  return false;
}

// taps of stage 1: z^-(N-1) boxcar(R1)^N
void cic2_stage1_taps(int R1, int N, std::vector<int64_t> *h) {
  std::vector<int64_t> c(1, 1);
  for (int st = 0; st < N; st++) {
    std::vector<int64_t> nx(c.size() + R1 - 1, 0);
    for (size_t i = 0; i < c.size(); i++) { for (int j = 0; j < R1; j++) { nx[i + j] += c[i]; } }
    c.swap(nx);
  }
  h->assign((size_t)N - 1, 0);
  h->insert(h->end(), c.begin(), c.end());
}

// history samples a handle must keep in front of a call so that chunk 0's warm-up steps and first window are readable
int cic2_hist_len(int in_eb, int R1, int N, int wu) {
  const int ls = 8 / in_eb;
  return wu * 256 * R1 + (N * R1 + 15) + 16 * ls + 64;
}

static int cic2_batch(const Shape *sh) { return (sh && sh->m == 2) ? 4 : 2; }

// Steps per chunk, warm-up included.  A chunk pays `wu` steps of re-read and re-computed input and one pipeline start (fragments, halo,
// two load groups before the first product): 48 steps where a row holds at least two such chunks, 24 where it holds two of those, else 12
// (same-process sweep in profiles/r6_cic2_ablation.txt: 2.31 / 2.14 / 1.99 ms at 12 / 24 / 48 steps on 2^20-sample rows; 64 and 96 lose again
// to the uneven tail of rows that hold four or three chunks).
static int cic2_steps_per_chunk(const void *shape, int wu, int64_t steps_per_row) {
  const Shape *sh = (const Shape *)shape;
  const int g = cic2_batch(sh);
  ACDSP_TUNE_ENV(env, "ACDSP_CIC2_NST");   // tuning knob
  int nst = env && atoi(env) > 0 ? atoi(env) : (steps_per_row >= 2 * 47 ? 48 : (steps_per_row >= 2 * 23 ? 24 : 12));
  if (nst < wu + 1) { nst = wu + 1; }
  nst = (nst + g - 1) / g * g;
  if (nst > 96) { nst = 96; }
  return nst;
}

// Outputs [0, *covered) of the call are written; the caller runs the recurrence kernel on the rest (the ragged end of the call).
hipError_t launch_cic2(const CicParams &p, const FirGenPlan &pl, const uint32_t *d_frag, int R1, int R2, int wu, int64_t n_out,
                       hipStream_t s, int64_t *covered) {
  *covered = 0;
  const Shape *sh = find_shape(p.in_eb, R1);
  if (!sh || pl.pc > sh->pct || pl.nb > sh->nbt || pl.R != R1 || n_out <= 0) { return hipSuccess; }
  Cic2Args a;
  memset(&a, 0, sizeof a);
  a.pl = pl;
  a.n_ch = p.n_ch; a.N = p.N; a.me = p.me; a.R2 = R2; a.w_int = p.w_int;
  a.in_F = p.in.F; a.out = p.out; a.out_eb = p.out_eb; a.out_simple = p.out_simple;
  a.hl = p.hl;
  const int ls = 8 / p.in_eb;
  const int64_t w0slot = (p.first - pl.off) / 16;                    // exact: the plan aligns the window start to a slot
  a.ring_delta = (int32_t)(((w0slot % ls) + ls) % ls);
  a.wu = wu;
  a.rcp2 = (uint32_t)((0x100000000ull + R2 - 1) / R2);
  // pairwise recombination of the plane accumulators (fir_gen.hip: set_pairwise)
  {
    const int px = p.in_eb;
    int64_t bw[16] = {0};
    for (int w = 0; w < px + pl.pc - 1; w++) {
      for (int q = 0; q < pl.pc; q++) { if (w - q >= 0 && w - q < px) { bw[w] += 128 * pl.dig_abs[q]; } }
    }
    bool ok = true;
    for (int w = 0; w + 1 < 16; w += 2) { ok = ok && (bw[w + 1] * 256 + bw[w] < (int64_t(1) << 31)); }
    static const bool no_pw = getenv("ACDSP_GEN_NO_PW") != nullptr;
    a.pw = (ok && !no_pw) ? 1 : 0;
    unsigned __int128 bias = 0;
    for (int q = 0; q < px - 1; q++) { bias += (unsigned __int128)1 << (8 * q); }
    a.corr = (int64_t)(unsigned long long)((unsigned __int128)128 * bias * (unsigned long long)pl.sum_h);
  }
  { ACDSP_TUNE_ENV(dbg_env, "ACDSP_CIC2_DBG"); a.dbg = dbg_env ? atoi(dbg_env) : 0; }
  a.first = p.first; a.n_out = n_out; a.n16 = (p.n_in + 15) / 16 * 16;
  a.in_stride = p.in_stride; a.out_stride = p.out_stride;
  a.x = p.x; a.y = p.y; a.hist = p.hist;
  // complete chunks only: every u of the chunk has its whole window inside the call's samples
  const int64_t u_valid = p.n_in > p.first ? (p.n_in - p.first + R1 - 1) / R1 : 0;     // u[m] needs x[first + m R1]
  a.nst = cic2_steps_per_chunk(sh, wu, u_valid / 256);
  const int nmain = a.nst - wu;
  const int64_t chunks = u_valid / (256 * (int64_t)nmain);
  // ... and one shorter chunk behind them for the end of the call (a whole number of load-group pairs)
  const int g = cic2_batch(sh);
  const int64_t rem_steps = u_valid / 256 - chunks * nmain;
  a.full_chunks = (int32_t)chunks;
  a.nst_last = (int32_t)((rem_steps + wu) / g * g);
  const int last_main = a.nst_last > wu ? a.nst_last - wu : 0;
  if (last_main == 0) { a.nst_last = 0; }
  if (chunks + (last_main > 0) < 1 || u_valid >= (int64_t(1) << 30) || chunks >= (int64_t(1) << 30)) { return hipSuccess; }
  // chunk 0 reads back to c0 = first - off - 16 delta - wu 256 R1
  if (-(p.first - pl.off - 16 * (int64_t)a.ring_delta - (int64_t)wu * 256 * R1) > p.hl) { return hipSuccess; }
  const int64_t fast_out = ((chunks * nmain + last_main) * 256 + R2 - 1) / R2;         // j R2 < u indices covered
  dim3 grid((unsigned)(chunks + (last_main > 0)), (unsigned)p.n_ch);
  // XCD-affine chunk order: -2 .. -4 % in the same-process sweeps (profiles/r6_cic2_ablation.txt); ACDSP_XCD_MAP=0 is the A/B knob
  a.xcd_map = (xcd_map_wanted(true) && ((int64_t)grid.x * grid.y) % 8 == 0) ? 1 : 0;
  const v4i *fr = (const v4i *)d_frag;
  hipError_t e = hipErrorInvalidValue;
  e = sh->launch(g, p.N, grid, s, a, fr);
  if (e != hipSuccess) { return e; }
  *covered = fast_out < n_out ? fast_out : n_out;
  return hipSuccess;
}

#endif   // ACDSP_CIC2_PART == 0

}  // namespace acdsp
