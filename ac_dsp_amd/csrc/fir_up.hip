// fir_up.hip -- exact interpolating (polyphase) FIR on the matrix cores.
//
//     z[n*L + j] = sum_{k < NT} E_j[k] * x[n - k]     (mod 2^64),   j = 0 .. L-1
//
// L outputs per input sample, one NT-tap sub-filter per phase.  It serves the two interpolators of SURVEY 8:
//   * ac_cic_intr_full through its FIR identity: out[q] = sum_n h[q - R n] x[n] with h = z^-(N-1) boxcar(R M')^N
//     (reference ac_cic_full_core.h:143-160,198-255, ac_cic_intr_full.h:195-215; identity in cic.hip), i.e. L = R,
//     E_r[k] = h[r + R k] -- the N wide adds per OUTPUT sample of intStage become 8-bit MFMAs;
//   * ac_poly_intr's exact-accumulation class (reference ac_poly_intr.h:126-257): the folded sub-filter sums, their
//     one-sample delay and the symmetric-pair combination (t1 -/+ t2) >> 1 are linear in the input, so they are folded
//     into per-phase taps on the host (engine.hip) and only the >> 1 and the OUT_TYPE conversion stay in the epilogue.
//
// Mapping (v_mfma_i32_32x32x32_i8).  The write-out decides the speed (outputs are L x the input volume), so the MFMA
// tile is shaped after the OUTPUT stream: the 32 rows of a tile are the L phases of SPC = 32 / L consecutive input samples
// and column c is the c-th such sample group, i.e. D[i][c] = z[32 c + i] -- one MFMA group produces 1024 CONSECUTIVE
// outputs of one channel.  With d = i / L, j = i % L:
//     D[i][c] = sum_kappa A[i][kappa] * X_c[kappa],   A[i][kappa] = E_j[32 NB - SPC + d - kappa],
//     X_c[kappa] = x[n0 + SPC (c + 1) - 32 NB + kappa],   kappa in [0, 32 NB)
// (NB = 1 or 2 K blocks; only SPC + NT - 1 of the 32 NB window positions carry taps -- the matrix pipe has slack to burn
// here, HBM write bandwidth does not).  The Toeplitz fragments A are the same for every column group and stay in
// registers.  Operands are split into byte planes exactly as in fir_gen.hip (x: 2 or 4 planes, lower ones re-biased to
// signed; taps: balanced base-256 digits), products of equal weight share an int32 accumulator, the 64-bit recombination
// runs once per output.  The re-bias correction depends on the phase: 128 * sum_k E_j[k] * sum_{p < PX-1} 256^p, a small
// per-lane table (the rows of a lane repeat with period L <= 32).
//
// Data movement.  One wave = one channel x a short chunk of 1 .. 4 steps (about 32 KB of outputs); a step is 512 input samples =
// L / 2 MFMA groups.  The samples of every step of the chunk (+ 32 NB of history in front of each) are loaded up front, one
// register set per step; a step splits its set into byte planes (v_perm_b32) and stages them in LDS as plain byte arrays; X_c is a 16-byte read at byte offset SPC (32 g + c + 1) + 32 b + 16 h of the plane
// (unaligned for SPC < 16: the LDS takes it).  Every group is converted into a padded LDS tile (conflict-free 8..32-byte
// writes per lane) and leaves as full-wave contiguous 8-byte-per-lane stores: 8 KB runs per group for 8-byte outputs
// (the first version of this kernel wrote 256-byte runs from 32 places per wave and reached 2.6 TB/s on the CIC row; the
// one-thread-per-output VALU kernel, 512-byte runs, 4.0 TB/s).  Rounds 2 - 4 ran chunks of 2 - 8 steps with the loads of a
// step two steps ahead of their use; the one-shot chunks measure 4 - 7 % faster on both bench rows and 13 % at L = 4
// (profiles/r4_up_oneshot.txt) -- as with the decimating kernels (fir_gen_ring_kernel), what the memory system rewards is many short
// waves in memory order whose loads are all in flight before their first store.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "fir_kernels.hpp"

// Load policy of the 512 new samples of a step (read exactly once).  Round 4 A/B (profiles/r4_ab_up_nt.txt, pipelined form: non-temporal
// loads +2.3 % on the ac_cic_intr_full row, +-0 on the ac_poly_intr row; profiles/r4_up_oneshot.txt, one-shot form: +2 - 3 %) -- the
// bare 1:16 stream's gain from that policy (tools/fill_probe.hip: 5.4 -> 6.1 TB/s) does not carry over to a kernel whose time is its
// stores.  Plain loads; -DACDSP_UP_LD_NT builds the other form.
#ifdef ACDSP_UP_LD_NT
#define ACDSP_UP_LD(ptr) __builtin_nontemporal_load(ptr)
#else
#define ACDSP_UP_LD(ptr) (*(ptr))
#endif

namespace acdsp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

constexpr int kUpMaxPC = 3;

__device__ inline unsigned up_gather4(unsigned d0, unsigned d1, unsigned d2, unsigned d3, int p) {
  const unsigned sel = 0x0c0c0400u + 0x0101u * (unsigned)p;
  const unsigned lo = __builtin_amdgcn_perm(d1, d0, sel);
  const unsigned hi = __builtin_amdgcn_perm(d3, d2, sel);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// index into the per-lane phase table of accumulator register r (row i = (r & 3) + 8 (r >> 2) + 4 h, phase i % L); factors that do not
// divide 32 have no period inside the 16 registers: one entry per register
template <int L> __device__ constexpr int up_tab_idx(int r) { return (L >= 32 || 32 % L != 0) ? r : (L == 16 ? (r & 3) + 4 * ((r >> 2) & 1) : (r & 3)); }
template <int L> constexpr int up_tab_size() { return (L >= 32 || 32 % L != 0) ? 16 : (L == 16 ? 8 : 4); }
// input samples per MFMA column: the largest power of two with SPC * L <= 32 rows (L = 7: 4 samples x 7 phases = 28 of the 32 rows; the
// Toeplitz fragments of the idle rows are zero and their outputs are never written)
constexpr int up_spc(int L) { return L <= 2 ? 16 : (L <= 4 ? 8 : (L <= 8 ? 4 : 2)); }

}  // namespace

struct UpArgs {
  FirParams p;                // x, in_stride, y, out_stride, formats; p.lossless_shift / p.acc used by mode 0
  int64_t slot0;              // first input slot (16 samples) of the launch; >= 2 NB
  int64_t n_steps;            // steps of 32 slots
  int64_t steps_per_wave;
  int64_t out_off;            // output element index = n * L + j + out_off
  int32_t mode;               // 0: poly_intr ((V << shift) >> sh_j, ACC -> OUT)   1: CIC (wrap to w_int, IN fraction -> OUT)
  int32_t w_int, out_simple;
  uint32_t sh_mask;           // mode 0: bit j set = phase j is a symmetric pair: halve
  const int64_t *corr;        // [L] re-bias correction per phase (mod 2^64)
  // EPI 1 (mode 0, every intermediate inside int32, AC_TRN / AC_RND into AC_WRAP / AC_SAT):
  //   q = (V + (rnd << sh_j)) >> (rs + sh_j);  q = clamp(q, lo, hi);  q = ((q << w) >> w) & mask       (all branch-free)
  // EPI 2 (mode 1, INT_TYPE and OUT_TYPE wider than 32 bits, same fraction, AC_WRAP): bit-field wraps of the high word
  int32_t e_rs, e_rnd, e_lo, e_hi, e_w;
  uint64_t e_mask;
  // EPI 4 (mode 0 into 4- / 8-byte containers, AC_TRN / AC_RND into AC_WRAP / AC_SAT, signed OUT_TYPE): q = ((V + c_rnd) >> c_rs) << c_ls2,
  // clamp to [c_lo, c_hi], sign-extend the low 64 - c_ko bits -- the generic conversion's result without its branches
  int64_t c_rnd, c_lo, c_hi;
  int32_t c_rs, c_ls2, c_ko;
  int32_t xcd_map;   // XCD-affine chunk order (acdsp_dev.hpp: xcd_remap)
};

// EPI 0: 64-bit recombination + the generic conversions (any Q / O mode; uniform branches per output).
// EPI 1: poly_intr with every intermediate inside int32 and a shift / clamp / wrap conversion (host-checked).
// EPI 2: CIC with a bit-field wrap conversion.  1, 2 and 3 are branch-free.
// EPI 3: CIC whose INT_TYPE (and OUT_TYPE container) fit 32 bits: EPI 1's recombination mod 2^32, one sign-extending wrap, a mask.
// EPI 4: poly_intr into 4- / 8-byte containers with a branch-free shift / clamp / wrap conversion in 64 bits (round 4).
// PCT: coefficient digit planes compiled in (2 or 3; the fragment array always has 3 per K block).
// NST: steps per wave (1 .. 4).  A wave is a short one-shot chunk: the samples of all its NST steps are loaded up front into NST register
// sets, then the steps run back to back with nothing but their stores on the memory pipeline; no prefetch state is carried.  The host
// picks NST so that a wave writes about 32 KB, dispatched in memory order (round 4; the software-pipelined form it replaces -- chunks
// of 2 - 8 steps, loads two steps ahead of their use -- measured 4 - 7 % slower on both interpolator rows: profiles/r4_up_oneshot.txt).
template <typename TIN, int PX, int PCT, int NBT, int L, int OEB, int EPI, int NST>
// (two waves per SIMD where the fragments + the register sets + accumulators need more than 168 registers: spills inside a step are
// VMEM operations that every store-counting wait would have to drain)
__global__ void __launch_bounds__(64, ((PX * PCT * NBT >= 12 || 32 % L != 0) ? 2 : 3)) fir_up_kernel(UpArgs a, const v4i *__restrict__ frag) {   // (factors that do not divide 32: sixteen-entry phase tables)
  constexpr int SPC = up_spc(L);                              // input samples per MFMA column
  constexpr int ROWS = SPC * L;                               // live rows of a tile (32 when L divides 32)
  constexpr int SS = 512;                                     // samples per step
  constexpr int G = 16 / SPC;                                 // MFMA groups (32 ROWS outputs each) per step
  constexpr int NLD = (int)sizeof(TIN) / 2;                   // 1 KB loads per step
  static_assert(NST >= 1 && NST <= 4, "up to four steps loaded up front");
  constexpr int HP = 32 * NBT;                                // history samples staged in front of a step
  constexpr int SPL = 16 / (int)sizeof(TIN);                  // samples per 16-byte load
  constexpr int NHL = HP / SPL;                               // lanes that load history
  constexpr int PLB = HP + SS + 16;                           // bytes of one plane array
  constexpr int FU = OEB == 8 ? 1 : ((OEB == 4 ? 2 : 4) < G ? (OEB == 4 ? 2 : 4) : G);   // groups per write-out
  constexpr int RUN = ROWS * OEB;                             // output bytes of one column
  constexpr int UNIT = OEB == 2 ? 8 : 16;                     // bytes a lane writes per tile store
  constexpr int RUNP = RUN + ((RUN / UNIT) % 2 == 0 ? UNIT : 0);   // column pitch of the tile: an odd number of store units (conflict-free writes)
  constexpr int NACC = PX + PCT - 1;
  constexpr int TS = up_tab_size<L>();
  static_assert(L >= 2 && L <= 16 && ROWS % 4 == 0 && ROWS <= 32, "live rows come in the accumulator groups of four");
  static_assert(G % FU == 0, "groups per write-out must divide the groups of a step");
  static_assert((FU * 32 * RUN) % (OEB == 2 ? 1024 : 512) == 0, "whole store instructions per write-out");
  __shared__ __attribute__((aligned(16))) unsigned char lds[PX * PLB + 64 * 16 + FU * 32 * RUNP];
  unsigned char *sink = lds + PX * PLB;                       // private dump of the lanes without a history load
  unsigned char *tile = sink + 64 * 16;
  const FirParams &p = a.p;
  const int lane = threadIdx.x;
  const int c = lane & 31, h = lane >> 5;
  int bx, ch;
  xcd_remap(a.xcd_map, bx, ch);

  v4i A[NBT][PCT];
#pragma unroll
  for (int b = 0; b < NBT; b++) {
#pragma unroll
    for (int q = 0; q < PCT; q++) { A[b][q] = frag[((size_t)b * kUpMaxPC + q) * 64 + lane]; }
  }
  // phase-dependent constants of this lane's accumulator registers
  int64_t corr_t[TS];
  int corr32_t[TS], shift_t[TS];   // EPI 1: correction + rounding constant, total right shift
  unsigned sh_t = 0;
#pragma unroll
  for (int t = 0; t < TS; t++) {
    // representative register of table entry t: r with up_tab_idx(r) == t
    const int r = L >= 32 ? t : (L == 16 ? (t & 3) + 4 * (t >> 2) : t);
    const int j = ((r & 3) + 8 * (r >> 2) + 4 * h) % L;
    const unsigned sh = (a.sh_mask >> j) & 1u;
    corr_t[t] = a.corr[j] - (EPI == 2 ? (int64_t)((uint64_t(1) << 31) + (uint64_t(1) << 47)) : 0);
    sh_t |= sh << t;
    corr32_t[t] = (int)a.corr[j] + (a.e_rnd << sh);   // ((V >> sh) + rnd) >> rs == (V + (rnd << sh)) >> (rs + sh)
    shift_t[t] = a.e_rs + (int)sh;
  }

  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  char *yrow = (char *)p.y + ((int64_t)ch * p.out_stride + a.out_off) * OEB;
  const int64_t st0 = (int64_t)bx * a.steps_per_wave;
  const int64_t st1 = (st0 + a.steps_per_wave < a.n_steps) ? st0 + a.steps_per_wave : a.n_steps;

  const int hl = lane < NHL ? lane : NHL - 1;                 // lanes past the history repeat its last load ...
  // one register set per step of the chunk
  v4i pre[NST][NLD], preh[NST];
  auto fetch = [&](int64_t st, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    if (st > st1 - 1) { st = st1 - 1; }                       // past the chunk: the last step again (never consumed)
    const TIN *src = xrow + 16 * a.slot0 + SS * st;
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      pre[S][q] = ACDSP_UP_LD((const v4i *)src + 64 * q + lane);
    }
    preh[S] = ((const v4i *)(src - HP))[hl];
  };
  // byte plane pp of the SPL samples in one 16-byte register set -> SPL bytes at `dst`
  auto put = [&](const v4i &v, int pp, unsigned char *dst) {
    if constexpr (sizeof(TIN) == 2) {
      const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
      unsigned lo = __builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, sel), hi = __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, sel);
      if (pp < PX - 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      *(v2u *)dst = (v2u){lo, hi};
    } else {
      unsigned w = up_gather4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w, pp);
      if (pp < PX - 1) { w ^= 0x80808080u; }
      *(unsigned *)dst = w;
    }
  };
  auto stage = [&](auto set_c) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      unsigned char *pl = lds + pp * PLB;
#pragma unroll
      for (int q = 0; q < NLD; q++) { put(pre[S][q], pp, pl + HP + (64 * q + lane) * SPL); }
      put(preh[S], pp, lane < NHL ? pl + lane * SPL : sink + lane * 16);   // ... and dump it into a private sink (branch-free)
    }
  };

  // write-out of FU finished groups: FU x 8 KB (OEB 8) / FU x 2 KB (OEB 2) contiguous, 8 bytes per lane and instruction.
  // e_unit = output element (before out_off) of column 0, row 0 of the first group.
  auto flush = [&](int64_t e_unit) {
    // Store shape and policy by output container, same-box A/B (alternating processes; profiles/r4_ab_up_store.txt, r3_ab_store_width.txt):
    //   2-byte outputs (ac_poly_intr row): 16 bytes per lane (1 KB per instruction), non-temporal -- 0.959 -> 0.911 ms against the
    //   8-byte-per-lane form of round 3 (which had measured the same as 16 bytes under the round-3 plain-store policy);
    //   8-byte outputs (ac_cic_intr_full row): 8 bytes per lane, plain -- 16 bytes per lane 3.64 against 3.63 ms, and the non-temporal
    //   policy costs 13 % at either width (4.10 ms).
    if constexpr (OEB == 2) {
#pragma unroll
      for (int k = 0; k < FU * 32 * RUN / 1024; k++) {
        // two aligned 8-byte reads: the column pitch (72 bytes at 32 rows) is a multiple of 8, not of 16, and a DS access off its natural
        // alignment is replayed lane by lane (~85 instead of ~8 cycles per wave read: profiles/r5_lds_align.txt; round 4 read 16 bytes here)
        const int lin = (k * 64 + lane) * 16;
        const int c0 = lin / RUN, w0 = lin % RUN, c1 = (lin + 8) / RUN, w1 = (lin + 8) % RUN;
        typedef long v2l_ __attribute__((ext_vector_type(2)));
        const v2l_ val = (v2l_){*(const long *)(tile + c0 * RUNP + w0), *(const long *)(tile + c1 * RUNP + w1)};
        __builtin_nontemporal_store(val, (v2l_ *)(yrow + e_unit * OEB + lin));
      }
    } else {
#pragma unroll
      for (int k = 0; k < FU * 32 * RUN / 512; k++) {
        const int lin = (k * 64 + lane) * 8;
        const int cc = lin / RUN, w = lin % RUN;
        const long val = *(const long *)(tile + cc * RUNP + w);
        *(long *)(yrow + e_unit * OEB + lin) = val;
      }
    }
  };
  // One step: stage its register set, then per group read the fragments, run the MFMAs, convert into the tile, write the tile out.
  auto body = [&](int64_t st, auto set_c) __attribute__((always_inline)) {
    // (single-wave workgroup: the LDS operations of a wave execute in order, no barrier needed)
    stage(set_c);
    const int64_t e_step = (16 * a.slot0 + SS * st) * (int64_t)L;   // output element (before out_off) of the step's first sample, phase 0
#pragma unroll
    for (int g = 0; g < G; g++) {
      v4i X[NBT][PX];
#pragma unroll
      for (int b = 0; b < NBT; b++) {
#pragma unroll
        for (int pp = 0; pp < PX; pp++) {
          // 16 bytes at byte offset SPC (32 g + c + 1) + ...: only SPC-aligned, and a DS read off its natural alignment is replayed lane by
          // lane (round 4 issued one 16-byte read here: 26 % SQ_LDS_BANK_CONFLICT on the L = 8 rows was this).  Aligned pieces instead.
          const unsigned char *src = lds + pp * PLB + SPC * (32 * g + c + 1) + 32 * b + 16 * h;
          if constexpr (SPC >= 16) { X[b][pp] = *(const v4i *)src; }
          else if constexpr (SPC == 8) {
            typedef int v2i_ __attribute__((ext_vector_type(2)));
            const v2i_ lo = *(const v2i_ *)src, hi = *(const v2i_ *)(src + 8);
            X[b][pp] = (v4i){lo.x, lo.y, hi.x, hi.y};
          } else if constexpr (SPC == 4) {
            X[b][pp] = (v4i){*(const int *)src, *(const int *)(src + 4), *(const int *)(src + 8), *(const int *)(src + 12)};
          } else {
            // 2-byte steps: five aligned dwords, realigned per lane (odd columns start two bytes into a dword)
            const unsigned sh = 2u * ((unsigned)(c + 1) & 1u);
            const unsigned char *al = src - sh;
            const unsigned d0 = *(const unsigned *)al, d1 = *(const unsigned *)(al + 4), d2 = *(const unsigned *)(al + 8), d3 = *(const unsigned *)(al + 12),
                           d4 = *(const unsigned *)(al + 16);
            X[b][pp] = (v4i){(int)__builtin_amdgcn_alignbyte(d1, d0, sh), (int)__builtin_amdgcn_alignbyte(d2, d1, sh),
                             (int)__builtin_amdgcn_alignbyte(d3, d2, sh), (int)__builtin_amdgcn_alignbyte(d4, d3, sh)};
          }
        }
      }
      v16i acc[NACC];
#pragma unroll
      for (int w = 0; w < NACC; w++) { acc[w] = (v16i){0}; }
      if constexpr (EPI == 2) {
        static_assert(EPI != 2 || NACC >= 3, "biased pairs");
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = (int)0x80000000u; acc[2][r] = (int)0x80000000u; }
      }
      if constexpr (EPI == 1 || EPI == 3) {   // the correction + rounding constant rides in as the initial value of the lowest accumulator
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = corr32_t[up_tab_idx<L>(r)]; }
      }
#pragma unroll
      for (int b = 0; b < NBT; b++) {
#pragma unroll
        for (int q = 0; q < PCT; q++) {
#pragma unroll
          for (int pp = 0; pp < PX; pp++) {
            acc[pp + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[b][q], X[b][pp], acc[pp + q], 0, 0, 0);
          }
        }
      }
      // epilogue: lane (c, h), register r: row i = (r & 3) + 8 (r >> 2) + 4 h = output 32 c + i of the group.  The plane
      // accumulators are recombined pairwise in 32 bits first (|acc| < 2^22, so a + (b << 8) is exact), then in 64 bits.
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        int64_t o[4];
        int o32[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int r = 4 * g4 + rr;
          const int t = up_tab_idx<L>(r);
          int pr[(NACC + 1) / 2];
#pragma unroll
          for (int m = 0; m < (NACC + 1) / 2; m++) {
            pr[m] = (2 * m + 1 < NACC) ? (int)(((unsigned)acc[2 * m + 1][r] << 8) + (unsigned)acc[2 * m][r]) : acc[2 * m][r];
          }
          if constexpr (EPI == 3) {
            // everything mod 2^32: planes of weight 2^32 and above drop out; wrap to min(W_int, W_out) bits (sign-extending), mask for
            // unsigned OUT_TYPEs
            unsigned y32 = 0;
#pragma unroll
            for (int w = (NACC < 4 ? NACC : 4) - 1; w >= 0; w--) { y32 = (y32 << 8) + (unsigned)acc[w][r]; }
            o32[rr] = ((int)(y32 << a.e_rs) >> a.e_rs) & (int)a.e_mask;
          } else if constexpr (EPI == 1) {
            static_assert(EPI != 1 || NACC <= 4, "32-bit epilogue: four accumulators");
            // V + corr = sum_w acc[w] << 8 w by Horner's rule mod 2^32 (|V + corr| < 2^31, host-checked), one shift, one clamp
            // (v_med3_i32; the bounds are the int32 range when OUT_TYPE wraps at its container width)
            unsigned y32 = (unsigned)acc[NACC - 1][r];
#pragma unroll
            for (int w = NACC - 2; w >= 0; w--) { y32 = (y32 << 8) + (unsigned)acc[w][r]; }
            int q = (int)y32 >> shift_t[t];
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(q) : "v"(q), "s"(a.e_lo), "v"(a.e_hi));   // one SGPR per VALU instruction on gfx9
            o32[rr] = q;
          } else {
            // y = corr + sum_m sext(pr[m]) << 16 m, in 32-bit halves (carry chains instead of 64-bit shifts).  EPI 2 starts
            // acc[0] and acc[2] at 2^31, so pr[0] and pr[1] are biased to unsigned and need no sign extension (the bias is
            // taken out of corr_t)
            uint64_t y;
            {
              unsigned lo = (unsigned)corr_t[t], hi = (unsigned)((uint64_t)corr_t[t] >> 32);
#pragma unroll
              for (int m = 0; m < (NACC + 1) / 2; m++) {
                if (m == 0) { const unsigned s0 = lo + (unsigned)pr[0]; hi += (EPI == 2 ? 0u : (unsigned)(pr[0] >> 31)) + (s0 < lo); lo = s0; }
                else if (m == 1) {
                  const unsigned t1 = (unsigned)pr[1] << 16, s1 = lo + t1;
                  hi += (EPI == 2 ? (unsigned)pr[1] >> 16 : (unsigned)(pr[1] >> 16)) + (s1 < lo); lo = s1;
                }
                else if (m == 2) { hi += (unsigned)pr[2]; }
                else { hi += (unsigned)pr[3] << 16; }
              }
              y = ((uint64_t)hi << 32) | lo;
            }
            if constexpr (EPI == 2) {
              // CIC: wrap to INT_TYPE, then to OUT_TYPE (same fraction, AC_WRAP, both wider than 32 bits and OUT_TYPE signed or
              // no wider than INT_TYPE: host-checked) = one bit-field extract of the high word + a mask for unsigned OUT_TYPEs
              const int hi = (int)__builtin_amdgcn_sbfe((int)(y >> 32), 0, (unsigned)a.e_w) & (int)a.e_mask;
              o[rr] = (int64_t)(((uint64_t)(unsigned)hi << 32) | (uint32_t)y);
            } else if constexpr (EPI == 4) {
              int64_t v = (int64_t)(y << p.lossless_shift) >> ((sh_t >> t) & 1u);
              v = (int64_t)((uint64_t)((v + a.c_rnd) >> a.c_rs) << a.c_ls2);
              v = v < a.c_lo ? a.c_lo : (v > a.c_hi ? a.c_hi : v);
              o[rr] = (int64_t)((uint64_t)v << a.c_ko) >> a.c_ko;
            } else if (a.mode == 1) {
              o[rr] = requant64(wrap64((int64_t)y, a.w_int, 1), p.in.F, p.out);
            } else {
              const int64_t v = (int64_t)(y << p.lossless_shift) >> ((sh_t >> t) & 1u);
              o[rr] = requant64(v, p.acc.F, p.out);
            }
          }
        }
        unsigned char *dst = tile + ((g % FU) * 32 + c) * RUNP + (8 * g4 + 4 * h) * OEB;
        if (ROWS < 32 && 8 * g4 + 4 * h >= ROWS) { continue; }   // idle rows of a factor that does not divide 32
        if (EPI == 1 || EPI == 3) {
          if (OEB == 4) { *(v4i *)dst = (v4i){o32[0], o32[1], o32[2], o32[3]}; }
          else {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            *(v2u *)dst = (v2u){__builtin_amdgcn_perm((unsigned)o32[1], (unsigned)o32[0], 0x05040100u),
                                __builtin_amdgcn_perm((unsigned)o32[3], (unsigned)o32[2], 0x05040100u)};
          }
        } else if (OEB == 8) {
          typedef long v2l __attribute__((ext_vector_type(2)));
          *(v2l *)dst = (v2l){o[0], o[1]};
          *(v2l *)(dst + 16) = (v2l){o[2], o[3]};
        } else if (OEB == 4) {
          *(v4i *)dst = (v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
        } else {
          typedef short v4s __attribute__((ext_vector_type(4)));
          *(v4s *)dst = (v4s){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
        }
      }
      if ((g + 1) % FU == 0) { flush(e_step + 32 * ROWS * (int64_t)(g + 1 - FU)); }
      // keep the groups apart: interleaved, their accumulators and temporaries exceed the register budget
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (st0 >= st1) { return; }
  typedef std::integral_constant<int, 0> C0;
  typedef std::integral_constant<int, 1> C1;
  typedef std::integral_constant<int, 2> C2;
  typedef std::integral_constant<int, 3> C3;
  fetch(st0, C0());
  if constexpr (NST >= 2) { fetch(st0 + 1, C1()); }
  if constexpr (NST >= 3) { fetch(st0 + 2, C2()); }
  if constexpr (NST >= 4) { fetch(st0 + 3, C3()); }
  body(st0, C0());
  if constexpr (NST >= 2) { if (st0 + 1 < st1) { body(st0 + 1, C1()); } }
  if constexpr (NST >= 3) { if (st0 + 2 < st1) { body(st0 + 2, C2()); } }
  if constexpr (NST >= 4) { if (st0 + 3 < st1) { body(st0 + 3, C3()); } }
}

#ifndef ACDSP_UP_TU
// ---------------------------------------------------------------------------------------------
// host: digit planes, Toeplitz fragments, correction table
// ---------------------------------------------------------------------------------------------
bool fir_up_plan(const int64_t *E, int L, int nt, int px, FirUpPlan *pl, std::vector<uint32_t> *frag, std::vector<int64_t> *corr) {
  if (L < 2 || L > 16 || (up_spc(L) * L) % 4 != 0 || nt < 1 || px < 1 || px > 4) { return false; }
  // K blocks: the window of a column spans its SPC samples and nt - 1 earlier ones
  const int SPC = up_spc(L);
  int nb = 1;
  while (32 * nb - SPC < nt - 1) { nb++; }
  if (nb > 2) { return false; }
  std::vector<std::vector<int8_t>> dig(kUpMaxPC, std::vector<int8_t>((size_t)L * nt, 0));
  int pc = 1;
  for (int i = 0; i < L * nt; i++) {
    __int128 v = E[i];
    for (int q = 0; q < kUpMaxPC; q++) {
      int lo = (int)(((v % 256) + 256) % 256);
      if (lo >= 128) { lo -= 256; }
      dig[q][(size_t)i] = (int8_t)lo;
      v = (v - lo) / 256;
      if (lo != 0 && q + 1 > pc) { pc = q + 1; }
    }
    if (v != 0) { return false; }
  }
  pl->L = L; pl->nt = nt; pl->pc = pc; pl->nb = nb; pl->hs = 2 * nb;
  frag->assign((size_t)nb * kUpMaxPC * 64 * 4, 0u);
  for (int b = 0; b < nb; b++) {
    for (int q = 0; q < kUpMaxPC; q++) {
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 31, kg = lane >> 5;
        const int d = i / L, j = i % L;
        for (int dw = 0; dw < 4; dw++) {
          uint32_t word = 0;
          for (int bj = 0; bj < 4; bj++) {
            const int kappa = 32 * b + 16 * kg + 4 * dw + bj;
            const int tap = 32 * nb - SPC + d - kappa;
            const int8_t val = (d < SPC && tap >= 0 && tap < nt) ? dig[q][(size_t)j * nt + tap] : (int8_t)0;   // rows past SPC * L: idle
            word |= (uint32_t)(uint8_t)val << (8 * bj);
          }
          (*frag)[((((size_t)b * kUpMaxPC) + q) * 64 + lane) * 4 + dw] = word;
        }
      }
    }
  }
  // re-bias of the px - 1 unsigned planes: x = signed planes + 128 * sum_{p < px-1} 256^p
  unsigned __int128 bias = 0;
  for (int pp = 0; pp < px - 1; pp++) { bias += ((unsigned __int128)128) << (8 * pp); }
  corr->assign((size_t)L, 0);
  for (int j = 0; j < L; j++) {
    unsigned __int128 s = 0;
    for (int k = 0; k < nt; k++) { s += (unsigned __int128)(__int128)E[(size_t)j * nt + k]; }
    (*corr)[(size_t)j] = (int64_t)(uint64_t)(s * bias);
  }
  return true;
}

bool fir_up_shape_ok(int in_eb, int px, int nb, int L, int out_eb) {
  if (nb < 1 || nb > 2 || (L != 2 && L != 4 && L != 8 && L != 16 && L != 7 && L != 3 && L != 5 && L != 6)) { return false; }
  if (in_eb == 2 && px == 2) { return out_eb == 2 || (out_eb == 4 && nb == 1) || out_eb == 8; }
  if (in_eb == 4 && px == 4) { return nb == 1 && out_eb == 8; }
  return false;
}

// compiled shapes: poly_intr = int16 samples, 3 digit planes (the pair taps E_j - E_cj have 17 bits) or 2 when the set allows it, 2- or 8-byte outputs;
// CIC = int16 / int32 samples, 2 digit planes (boxcar^N taps of the BASELINE shapes fit 16 bits), 8-byte outputs (2-byte ones
// for int16 samples)
#endif   // ACDSP_UP_TU

// steps per wave: about 32 KB of outputs (a step writes 512 L OEB bytes); one for the branchy generic epilogue (two steps measured 1.8 x
// SLOWER than one on ac_poly_intr IF = 4 into 8-byte outputs: profiles/r4_poly_shapes.txt)
constexpr int up_nst(int L, int oeb, int epi) {
  const int n = 32768 / (512 * L * oeb), cap = epi == 0 ? 1 : 4;
  return n < 1 ? 1 : (n > cap ? cap : n);
}

template <typename TIN, int PX, int PCT, int NBT, int L>
static hipError_t launch_up_oeb(const UpArgs &a, const uint32_t *d_frag, int out_eb, int epi, dim3 grid, hipStream_t s) {
  const v4i *f = (const v4i *)d_frag;
  if (out_eb == 8) {
    if (epi == 4) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 8, 4, up_nst(L, 8, 4)>), grid, dim3(64), 0, s, a, f); }
    else if (epi == 2) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 8, 2, up_nst(L, 8, 2)>), grid, dim3(64), 0, s, a, f); }
    else { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 8, 0, up_nst(L, 8, 0)>), grid, dim3(64), 0, s, a, f); }
  } else if (out_eb == 4) {
    if constexpr (sizeof(TIN) == 2 && NBT == 1) {   // CIC on 16-bit inputs: INT_TYPE of up to 32 bits
      if (epi == 4) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 4, 4, up_nst(L, 4, 4)>), grid, dim3(64), 0, s, a, f); }
      else if (epi == 3) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 4, 3, up_nst(L, 4, 3)>), grid, dim3(64), 0, s, a, f); }
      else if (epi == 0) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 4, 0, up_nst(L, 4, 0)>), grid, dim3(64), 0, s, a, f); }
      else { return hipErrorNotSupported; }
    } else { return hipErrorNotSupported; }
  } else if (out_eb == 2) {
    if constexpr (sizeof(TIN) == 2) {
      if (epi == 1) {
        if constexpr (PCT == 3 || NBT == 1) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 1, up_nst(L, 2, 1)>), grid, dim3(64), 0, s, a, f); }
        else { return hipErrorNotSupported; }
      } else if (epi == 2) {
        if constexpr (PCT == 2) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 2, up_nst(L, 2, 2)>), grid, dim3(64), 0, s, a, f); }
        else { return hipErrorNotSupported; }
      } else { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 0, up_nst(L, 2, 0)>), grid, dim3(64), 0, s, a, f); }
    } else { return hipErrorNotSupported; }
  } else {
    return hipErrorNotSupported;
  }
  return hipGetLastError();
}

template <typename TIN, int PX, int PCT, int NBT>
static hipError_t launch_up_l(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s) {
  switch (L) {
#ifndef ACDSP_UP_ONLY_L8   // A/B builds compile the bench shapes only (the full set takes five minutes)
    case 2: return launch_up_oeb<TIN, PX, PCT, NBT, 2>(a, d_frag, out_eb, epi, grid, s);
    case 4: return launch_up_oeb<TIN, PX, PCT, NBT, 4>(a, d_frag, out_eb, epi, grid, s);
    case 16: return launch_up_oeb<TIN, PX, PCT, NBT, 16>(a, d_frag, out_eb, epi, grid, s);
    // factors that do not divide 32 (the reference's own CIC testbench: R = 7, ac_cic_intr_full_param.h:33-47): SPC * L live rows of 32
    case 3: return launch_up_oeb<TIN, PX, PCT, NBT, 3>(a, d_frag, out_eb, epi, grid, s);
    case 5: return launch_up_oeb<TIN, PX, PCT, NBT, 5>(a, d_frag, out_eb, epi, grid, s);
    case 6: return launch_up_oeb<TIN, PX, PCT, NBT, 6>(a, d_frag, out_eb, epi, grid, s);
    case 7: return launch_up_oeb<TIN, PX, PCT, NBT, 7>(a, d_frag, out_eb, epi, grid, s);
#endif
    case 8: return launch_up_oeb<TIN, PX, PCT, NBT, 8>(a, d_frag, out_eb, epi, grid, s);
    default: return hipErrorNotSupported;
  }
}

// ---- translation units: the shapes are split three ways for compile time (fir_up_b.hip, fir_up_c.hip re-include this file) ----
#define ACDSP_UP_SHAPE(NAME, TIN, PX, PCT, NBT)                                                                                   \
  hipError_t NAME(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s) {               \
    return launch_up_l<TIN, PX, PCT, NBT>(a, d_frag, L, out_eb, epi, grid, s);                                                   \
  }
#if !defined(ACDSP_UP_TU)
ACDSP_UP_SHAPE(launch_up_s221, int16_t, 2, 2, 1)
#elif ACDSP_UP_TU == 1
ACDSP_UP_SHAPE(launch_up_s231, int16_t, 2, 3, 1)
ACDSP_UP_SHAPE(launch_up_s232, int16_t, 2, 3, 2)
#else
ACDSP_UP_SHAPE(launch_up_i421, int32_t, 4, 2, 1)
ACDSP_UP_SHAPE(launch_up_i431, int32_t, 4, 3, 1)
#endif
#undef ACDSP_UP_SHAPE

#ifndef ACDSP_UP_TU
hipError_t launch_up_s231(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s);
hipError_t launch_up_s232(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s);
hipError_t launch_up_i421(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s);
hipError_t launch_up_i431(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s);

// Input slots [slot0, slot0 + 32 n_steps) of every channel; the caller covers everything else with the VALU kernels.
hipError_t launch_fir_up(const FirParams &p, const FirUpPlan &pl, int px, const uint32_t *d_frag, const int64_t *d_corr, int mode, int w_int,
                         int out_simple, uint32_t sh_mask, int64_t max_abs_v, int64_t slot0, int64_t n_steps, int64_t out_off, hipStream_t s) {
  if (n_steps <= 0) { return hipSuccess; }
  if (!fir_up_shape_ok(p.in_eb, px, pl.nb, pl.L, p.out_eb) || slot0 < pl.hs) { return hipErrorNotSupported; }
  UpArgs a;
  a.p = p; a.slot0 = slot0; a.n_steps = n_steps; a.out_off = out_off; a.mode = mode; a.w_int = w_int; a.out_simple = out_simple;
  a.sh_mask = sh_mask; a.corr = d_corr;
  a.e_rs = a.e_rnd = a.e_w = 0; a.e_lo = INT32_MIN; a.e_hi = INT32_MAX; a.e_mask = ~uint64_t(0);
  a.c_rnd = 0; a.c_lo = INT64_MIN; a.c_hi = INT64_MAX; a.c_rs = a.c_ls2 = a.c_ko = 0;
  int epi = 0;
  const int rs = p.acc.F - p.out.F;
  // bits (sign included) the accumulator value can reach: the host's bound on |V| where it has one, else ACC_TYPE's width
  int acc_bits = p.acc.W;
  if (mode == 0 && max_abs_v >= 0 && p.lossless_shift >= 0 && p.lossless_shift < 32) {
    int vb = 0;
    while (vb < 62 && (int64_t(1) << vb) <= max_abs_v) { vb++; }
    if (vb + p.lossless_shift + 1 < acc_bits) { acc_bits = vb + p.lossless_shift + 1; }
  }
  if (mode == 0 && px == 2 && p.lossless_shift == 0 && rs >= 0 && rs <= 28 && p.out_eb == 2 &&
      (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) && (p.out.O == ACDSP_SAT || (p.out.O == ACDSP_WRAP && p.out.W == 16)) &&
      max_abs_v >= 0 && max_abs_v < (int64_t(1) << 30)) {
    // 32-bit epilogue: poly_intr, no left shift into ACC_TYPE, |V| (+ rounding constant) inside int32; AC_SAT clamps, AC_WRAP at
    // the container width is the truncation of the 16-bit store
    epi = 1;
    a.e_rs = rs;
    a.e_rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (1 << (rs - 1)) : 0;
    if (p.out.O == ACDSP_SAT) { a.e_lo = (int32_t)p.out.lo; a.e_hi = (int32_t)p.out.hi; }
  } else if (mode == 0 && (px == 2 || (px == 4 && p.out_eb == 8)) && (p.out_eb == 8 || (p.out_eb == 4 && pl.nb == 1)) && p.out.S && (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) &&
             (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) && p.lossless_shift >= 0 && p.lossless_shift < 32 && rs >= -16 && rs <= 62 &&
             acc_bits + (rs < 0 ? -rs : 0) <= 63 && p.out.W >= 2 && p.out.W <= 8 * p.out_eb) {
    // exact-accumulation class into wider containers: the conversion without the generic epilogue's branches
    epi = 4;
    a.c_rs = rs > 0 ? rs : 0; a.c_ls2 = rs < 0 ? -rs : 0;
    a.c_rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (int64_t(1) << (rs - 1)) : 0;
    if (p.out.O == ACDSP_SAT) { a.c_lo = p.out.lo; a.c_hi = p.out.hi; a.c_ko = 0; }
    else { a.c_lo = INT64_MIN; a.c_hi = INT64_MAX; a.c_ko = 64 - p.out.W; }
  } else if (mode == 1 && out_simple >= 1) {
    // bit-field wrap of the high word: to INT_TYPE, then (out_simple 1) to an OUT_TYPE of the same fraction with AC_WRAP
    const int wo = out_simple == 2 ? w_int : p.out.W, so = out_simple == 2 ? 1 : p.out.S;
    const int wmin = wo < w_int ? wo : w_int;
    if (w_int > 32 && wmin > 32 && wmin < 64 && (so || wo <= w_int)) {
      epi = 2;
      a.e_w = wmin - 32;
      if (!so) { a.e_mask = (uint64_t)(~uint32_t(0) >> (64 - wo)); }
    } else if (w_int <= 32 && p.out_eb == 4 && wmin >= 1 && (so || wo <= w_int)) {
      epi = 3;
      a.e_rs = 32 - wmin;   // (y << rs) >> rs: sign-extending wrap to wmin bits
      if (!so) { a.e_mask = (uint64_t)(~uint32_t(0) >> (32 - wo)); }
    }
  }
  // one-shot waves of about 32 KB of outputs, dispatched in memory order (fir_up_kernel's NST)
  const int64_t spw = up_nst(pl.L, p.out_eb, epi);
  a.steps_per_wave = spw;
  dim3 grid((unsigned)((n_steps + spw - 1) / spw), (unsigned)p.n_ch);
  // XCD-affine chunk order: loses 2 - 5 % on both interpolator rows in the one-shot form (profiles/r4_up_oneshot.txt); ACDSP_XCD_MAP=1 forces it
  a.xcd_map = (xcd_map_wanted(false) && ((int64_t)grid.x * grid.y) % 8 == 0) ? 1 : 0;
  if (p.in_eb == 2) {
    if (mode == 0) {
      // poly_intr: the pair taps E_j -+ E_cj can have 17 bits = 3 digit planes; sets whose folded taps stay inside two planes
      // (|tap| < 2^15: e.g. any low-pass with sum |c| < 2) skip the third plane's MFMAs and its accumulator
      if (pl.nb == 1 && pl.pc <= 2) { return launch_up_s221(a, d_frag, pl.L, p.out_eb, epi, grid, s); }
      return pl.nb == 1 ? launch_up_s231(a, d_frag, pl.L, p.out_eb, epi, grid, s)
                        : launch_up_s232(a, d_frag, pl.L, p.out_eb, epi, grid, s);
    }
    // CIC: boxcar^N taps in two digit planes (the BASELINE shapes) or three (R^(N-1) past 2^15: R = 16 at N = 5)
    if (pl.nb != 1) { return hipErrorNotSupported; }
    return pl.pc <= 2 ? launch_up_s221(a, d_frag, pl.L, p.out_eb, epi, grid, s) : launch_up_s231(a, d_frag, pl.L, p.out_eb, epi, grid, s);
  }
  if (pl.nb != 1) { return hipErrorNotSupported; }
  return pl.pc <= 2 ? launch_up_i421(a, d_frag, pl.L, p.out_eb, epi, grid, s) : launch_up_i431(a, d_frag, pl.L, p.out_eb, epi, grid, s);
}

#endif   // ACDSP_UP_TU

}  // namespace acdsp
