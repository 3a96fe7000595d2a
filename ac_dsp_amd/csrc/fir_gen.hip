// fir_gen.hip -- generalised exact FIR on the matrix cores: wide inputs, wide coefficients, decimation.
//
//     y[m] = sum_k h[k] * x[first + m*R - k]   (mod 2^64),   m = 0, 1, ...
//
// with x in 1..8 byte planes (16/32/64-bit containers), h in 1..4 balanced base-256 digit planes and an
// integer decimation factor R.  It serves
//   * the lossless FIR class for inputs wider than 16 bits (e.g. the 36-bit CIC output feeding the
//     DDC's 127-tap ac_fir_const_coeffs: reference ac_fir_const_coeffs.h:190-199 with IN = <36,21>),
//   * the CIC decimator through its FIR identity  H(z) = z^-(N-1) (1 + ... + z^-(R*M'-1))^N  taken
//     mod 2^outW (reference ac_cic_full_core.h:80-135,198-255; identity checked in
//     tests/test_oracle.py::test_cic_closed_form_fir_identity): the N wide adds per input sample of
//     intStage become a handful of int8 MFMAs per 256 outputs.
//
// Mapping (v_mfma_i32_16x16x64_i8).  One wave = one channel x a chunk of steps; a step is 256 outputs:
// column n = output block m0+16n .. +15, row i = output inside the block.  With W_n = the 16-aligned
// start of column n's input window (W_n = W_0 + 16*R*n) and off = T_n - W_n (constant),
//     D[i][n] = sum_kappa A[i][kappa] * X[kappa][n],   A[i][kappa] = h[off + i*R - kappa],
//     X[kappa][n] = x[W_n + kappa],  kappa in [0, 64*NB).
// Both operands are split into byte planes; plane products with equal weight p+q share one int32
// accumulator (4 VGPRs); the 64-bit recombination, the re-bias correction of the unsigned input planes
// and the OUT_TYPE conversion happen once per output in the epilogue.
//
// Data movement.  Lane <-> 16-sample slot: a lane loads its slot (16*sizeof(TIN) contiguous bytes),
// builds one 16-byte vector per plane (v_perm_b32 byte gathers) and writes it to the plane array in
// LDS; column n / K-group kg / block b then reads slot R*n + 4b + kg.  For even R the slot index is
// padded by slot/R so that the 16 columns of a read hit 16 different 16-byte bank groups.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "fir_kernels.hpp"

// (The window-per-step kernels keep plain loads: their windows overlap and non-temporal loads bypass the L1 the re-reads need -- 13 - 25 %
// slower, profiles/r3_fill_probe.txt.  The ring kernel reads every byte once and takes that policy: fir_gen_ring_kernel.)
// Output tiles of the fast kernels leave with the non-temporal policy (-DACDSP_GEN_ST_PLAIN: plain stores, the A/B reference).  The
// speed of these rows depends on where the driver placed the input / output pair (profiles/r3_placement_modes.txt); the policy
// changes nothing on a fast pair and takes 2 - 4 % off a slow one (poly_dec, eight placements, two processes each).
#ifdef ACDSP_GEN_ST_PLAIN
#define ACDSP_GEN_ST(val, ptr) (*(ptr) = (val))
#else
#define ACDSP_GEN_ST(val, ptr) __builtin_nontemporal_store(val, ptr)
#endif

namespace acdsp {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kGenMaxPX = 8, kGenMaxPC = 3, kGenMaxNB = 8;  // A fragments: [b][q], <= 96 VGPRs

// ---------------------------------------------------------------------------------------------
// host: coefficient planes and A fragments
// ---------------------------------------------------------------------------------------------
bool fir_gen_plan(const int64_t *h, int n_taps, int R, int first_mod16, FirGenPlan *pl, std::vector<uint32_t> *frag) {
  // balanced base-256 digits of every tap
  int pc = 1;
  std::vector<std::vector<int8_t>> dig(kGenMaxPC, std::vector<int8_t>(n_taps, 0));
  if (n_taps < 1 || R < 1 || R > 64) { return false; }
  for (int k = 0; k < n_taps; k++) {
    __int128 v = h[k];
    for (int q = 0; q < kGenMaxPC; q++) {
      int lo = (int)(((v % 256) + 256) % 256);
      if (lo >= 128) { lo -= 256; }
      dig[q][k] = (int8_t)lo;
      v = (v - lo) / 256;
      if (lo != 0 && q + 1 > pc) { pc = q + 1; }
    }
    if (v != 0) { return false; }  // needs more than kGenMaxPC digits (dig[][] has exactly kGenMaxPC rows)
  }
  // window geometry: T_n mod 16 == first mod 16; window start is the 16-aligned floor of T_n - (taps-1)
  const int a = ((first_mod16 - (n_taps - 1)) % 16 + 16) % 16;
  const int off = (n_taps - 1) + a;
  const int nb = (off + 15 * R + 1 + 63) / 64;
  if (nb > kGenMaxNB) { return false; }
  pl->pc = pc; pl->nb = nb; pl->off = off; pl->R = R;
  __int128 sum = 0, sum_abs = 0;
  for (int k = 0; k < n_taps; k++) { sum += (__int128)h[k]; sum_abs += h[k] < 0 ? -(__int128)h[k] : (__int128)h[k]; }
  pl->sum_h = (int64_t)(unsigned long long)sum;  // mod 2^64
  pl->sum_abs_h = sum_abs < ((__int128)1 << 62) ? (int64_t)sum_abs : (int64_t(1) << 62);
  for (int q = 0; q < kGenMaxPC; q++) {
    int64_t sa = 0;
    for (int k = 0; k < n_taps; k++) { sa += dig[q][k] < 0 ? -(int64_t)dig[q][k] : (int64_t)dig[q][k]; }
    pl->dig_abs[q] = sa;
  }
  frag->assign((size_t)pc * nb * 64 * 4, 0u);
  for (int q = 0; q < pc; q++) {
    for (int b = 0; b < nb; b++) {
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, kg = lane >> 4;
        for (int dw = 0; dw < 4; dw++) {
          uint32_t word = 0;
          for (int bj = 0; bj < 4; bj++) {
            const int kappa = 64 * b + 16 * kg + 4 * dw + bj;
            const int tap = off + i * R - kappa;
            const int8_t val = (tap >= 0 && tap < n_taps) ? dig[q][tap] : (int8_t)0;
            word |= (uint32_t)(uint8_t)val << (8 * bj);
          }
          (*frag)[(((size_t)q * nb + b) * 64 + lane) * 4 + dw] = word;
        }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
struct GenArgs {
  FirGenPlan pl;
  int32_t px;                 // input byte planes
  int32_t pad;                // slot -> LDS slot map (gen_slot_map): 1 = two empty slots after every R (R % 4 == 0), 0 = identity
  uint32_t rcp;               // ceil(2^32 / R) for the slot / R division
  int32_t n_slots;            // slots staged per step
  int32_t out_mode;           // 0: FIR class A (shift, ACC wrap, requant)   1: CIC (wrap to w_int, requant from F_in)
  int32_t out_simple;         // CIC: 2 = OUT holds INT_TYPE (one wrap), 1 = same fraction + AC_WRAP (two wraps), 0 = general
  int32_t w_int;
  int64_t corr;               // 128 * sum(h) * sum_{p < px-1} 256^p  (mod 2^64)
  int64_t first;              // local input index of output 0
  int64_t n_out;              // outputs per channel in this call
  int64_t steps_per_wave, n_steps;
  int64_t n16;                // n_in rounded up to 16: rows are readable that far
  int32_t obuf_off;           // byte offset of the 256-output tile in LDS (row-contiguous write-out)
  int32_t chunk0;             // first chunk (of steps_per_wave steps) this launch covers: blockIdx.x + chunk0
  int32_t xcd_map;            // fast kernels: XCD-affine chunk order (xcd_remap below); the host sets it only when grid.x * grid.y % 8 == 0
  int32_t edge_chunk;         // cascade kernel, PART 2: the row's last complete chunk (block x = 1; block x = 0 is chunk 0)
  int32_t ring_delta;         // ring kernel: slots between the first window's start and its 128-byte line (< 8 / sizeof(TIN))
  // branch-free output conversion of the fast kernel (host-derived from out_mode / the formats):
  //   v = wrapS_{64-ka}(y << ls);  v = ((v + rnd) >> rs) << ls2;  v = clamp(v, lo, hi);  v = wrapS_{64-ko}(v)
  int32_t e_ls, e_ka, e_rs, e_ls2, e_ko;
  int64_t e_rnd, e_lo, e_hi;
  int32_t out_vec_ok;         // output rows are 16-byte aligned: whole steps leave as 1 KB-per-instruction stores
  // ring kernel: which stages of that conversion do anything (bit 0 shift / wrap to ACC_TYPE, 1 rounding shift, 2 clamp, 3 final wrap) -- OUT =
  // ACC = <64,32> of the reference testbenches needs none of them -- and whether neighbouring plane accumulators may be combined in 32 bits
  int32_t e_stages, pw;
  // 32-bit limb epilogues of the cascade kernel (host-verified bounds, see launch_cascade): the constant of the
  // recombination (re-bias correction + rounding) enters as balanced base-256 digits = the initial accumulator values
  int32_t dig[8];
  int32_t l_hb_lo, l_hb_hi;   // stage B, AC_SAT: clamp of the high limb in front of the funnel shift
  int32_t l_lo, l_hi, l_w;    // stage B: OUT range (AC_SAT) / OUT width (AC_WRAP: l_lo = l_hi = 0)
  // class B on the ring kernel (LZ instantiations, see fir_gen_ring_kernel): acc = (S + K - sum_e r_e) >> lz_s with S the exact sum of the
  // matrix cores, r_e = (f_e * cl_e + h) mod 2^s the bits the reference's per-tap quantisation drops, K = entries * h
  int32_t lz_s;               // s = F_in + F_c - F_acc (1 .. 15)
  int32_t lz_neg, lz_var_y, lz_p_n, lz_p_woff, lz_p_voff, lz_s_n, lz_s_woff, lz_flush;   // FirLossyPlan
  uint32_t lz_h2, lz_m2;      // rounding constant and mask of the dropped bits, in both 16-bit fields
  int64_t lz_k;
  const uint32_t *lz_tab;     // [kLossyTabWords] slot coefficients (c_S, c_D) per iteration, pair loop first
};


__device__ inline int phys_slot(int s, const GenArgs &a) {
  return a.pad ? s + 2 * (int)__umulhi((unsigned)s, a.rcp) : s;   // s + 2 (s / R) (exact for s < 2^16)
}

// LDS slot map of a plane.  The MFMA operand read of lane (n_col, kg) is slot R n_col + 4 b + kg, a ds_read_b128, which the LDS
// serves in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32: MI355X_MICROARCH.md, LDS) over 64
// dword banks = 16 slots: a group holds eight columns at kg and the other eight at kg + 1.  With R a multiple of 4, TWO empty slots
// after every R make column n start at (R + 2) n: the eight columns of one half land on eight distinct even offsets mod 16 and the
// other half, one slot on, on the odd ones -- no conflict for any K block.  (Round 2 padded ONE slot per R, right for contiguous
// 16-lane groups and 4 extra cycles per K-block read on the real ones: the 24 - 33 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of
// configs 3, 5 and poly_dec in profiles/r2_*.)  R = 2 mod 4 is conflict-free as it stands; odd R keeps the identity (one or two
// extra cycles per group).  The staging writes are consecutive slots, 8 of them (32 banks) per bank pass: holes at multiples of 8
// slots leave them conflict-free (R = 4 mod 8: 2-way on some passes).  An XOR map (bits 1 - 3 of the slot ^ the column index) does
// the same without the holes and was measured first; it is not an add, so the cascade kernel could no longer address its ten
// staging pieces as one base register + immediates, and the register allocation of that kernel (255 VGPRs) fell over: +3 .. +35 %
// on the fused DDC depending on the form.  tools/lds_slot_map_check.py replays the lane groups for any R and map.
// Returns the slots a plane must allocate for n slots.
static int gen_slot_map(GenArgs &a, int R, int n) {
  a.rcp = (uint32_t)((0x100000000ull + R - 1) / R);
  a.pad = (R % 4 == 0) ? 1 : 0;
  return a.pad ? n + 2 * (n / R) : n;
}

// gather byte `p` (0..3 of a dword) of four dwords into one dword
__device__ inline unsigned gather4(unsigned d0, unsigned d1, unsigned d2, unsigned d3, int p) {
  const unsigned sel = 0x0c0c0400u + 0x0101u * (unsigned)p;               // result bytes: [d0.p, d1.p, 0, 0]
  const unsigned lo = __builtin_amdgcn_perm(d1, d0, sel);
  const unsigned hi = __builtin_amdgcn_perm(d3, d2, sel);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// SMALL: at most 3 K blocks x 2 coefficient digits (24 VGPRs of A fragments instead of 96): the common CIC /
// DDC shapes then fit three or four waves per SIMD, which is what hides the HBM latency of this streaming kernel.
// (Three waves per SIMD only while the byte planes leave room: six and more planes of 8-byte samples hold 24 - 32 VGPRs of X fragments
// and 36 - 40 of accumulators beside the A fragments, and spilled at 168 registers.)
template <typename TIN, int PX, bool SMALL>
__global__ void __launch_bounds__(64, (SMALL && PX <= 5) ? 3 : 2) fir_gen_kernel(FirParams p, const v4i *__restrict__ frag, GenArgs a) {
  constexpr int kMaxNB = SMALL ? 3 : kGenMaxNB, kMaxPC = SMALL ? 2 : kGenMaxPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [PX][phys slots][16]
  const int lane = threadIdx.x;
  const int n_col = lane & 15, kg = lane >> 4;
  const int ch = blockIdx.y;
  const int NB = a.pl.nb, PC = a.pl.pc, R = a.pl.R;
  const int plane_bytes = a.obuf_off / PX;

  v4i A[kMaxNB * kMaxPC];   // A[b * kMaxPC + q]; only the (b < NB, q < PC) entries are loaded and used
#pragma unroll
  for (int b = 0; b < kMaxNB; b++) {
#pragma unroll
    for (int q = 0; q < kMaxPC; q++) {
      A[b * kMaxPC + q] = (b < NB && q < PC) ? frag[((size_t)q * NB + b) * 64 + lane] : (v4i){0, 0, 0, 0};
    }
  }

  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  const TIN *hrow = (const TIN *)p.hist + (int64_t)ch * p.hl + p.hl;
  const int64_t s0 = ((int64_t)blockIdx.x + a.chunk0) * a.steps_per_wave;
  const int64_t s1 = (s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps;

  // Slots of the next step are fetched into registers while the current step multiplies (when the
  // per-lane slot count fits the register budget), so the HBM latency overlaps MFMA + epilogue.
  // (not for six and more planes of 8-byte samples: 96 VGPRs of A fragments + 40 of accumulators + 32 of X fragments leave no room
  // for 64 of prefetch -- those instantiations spilled 2 - 27 registers; they load each slot right before staging it)
  constexpr bool CAN_PF = !(sizeof(TIN) == 8 && PX >= 6);
  constexpr int SPLMAX = CAN_PF ? 16 / (int)sizeof(TIN) : 1;            // 64 VGPRs of prefetch
  const int spl = (a.n_slots + 63) / 64;
  const bool prefetch = CAN_PF && spl <= SPLMAX;
  v4i pre[SPLMAX][sizeof(TIN)];
  auto slot_src = [&](int64_t W0, int sl) -> const TIN * {
    const int64_t t = W0 + 16 * (int64_t)sl;
    return (t < 0) ? hrow + t : xrow + ((t < a.n16) ? t : 0);   // beyond the data: any valid address (never used)
  };
  auto fetch = [&](int64_t st) {
    const int64_t W0 = a.first + st * 256 * R - a.pl.off;
#pragma unroll
    for (int j = 0; j < SPLMAX; j++) {
      if (j < spl) {
        const int sl = (lane + 64 * j < a.n_slots) ? lane + 64 * j : a.n_slots - 1;
        const TIN *src = slot_src(W0, sl);
#pragma unroll
        for (int q = 0; q < (int)sizeof(TIN); q++) { pre[j][q] = ((const v4i *)src)[q]; }
      }
    }
  };
  // split one slot (16 samples) into its byte planes and store them
  auto stage_slot = [&](const v4i (&raw)[sizeof(TIN)], int sl) {
    union { v4i v[sizeof(TIN)]; unsigned d[4 * sizeof(TIN)]; } u;
#pragma unroll
    for (int q = 0; q < (int)sizeof(TIN); q++) { u.v[q] = raw[q]; }
    const int ps = phys_slot(sl, a);
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      v4i o;
      if (sizeof(TIN) == 2) {        // 16 samples = 8 dwords, plane pp = byte pp of each 16-bit sample
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        o.x = (int)__builtin_amdgcn_perm(u.d[1], u.d[0], sel); o.y = (int)__builtin_amdgcn_perm(u.d[3], u.d[2], sel);
        o.z = (int)__builtin_amdgcn_perm(u.d[5], u.d[4], sel); o.w = (int)__builtin_amdgcn_perm(u.d[7], u.d[6], sel);
      } else if (sizeof(TIN) == 4) { // 16 samples = 16 dwords
        o.x = (int)gather4(u.d[0], u.d[1], u.d[2], u.d[3], pp); o.y = (int)gather4(u.d[4], u.d[5], u.d[6], u.d[7], pp);
        o.z = (int)gather4(u.d[8], u.d[9], u.d[10], u.d[11], pp); o.w = (int)gather4(u.d[12], u.d[13], u.d[14], u.d[15], pp);
      } else {                       // 16 samples = 32 dwords; byte pp of sample e lives in dword 2e + pp/4
        const int hi = pp >> 2, bp = pp & 3;
        o.x = (int)gather4(u.d[0 + hi], u.d[2 + hi], u.d[4 + hi], u.d[6 + hi], bp);
        o.y = (int)gather4(u.d[8 + hi], u.d[10 + hi], u.d[12 + hi], u.d[14 + hi], bp);
        o.z = (int)gather4(u.d[16 + hi], u.d[18 + hi], u.d[20 + hi], u.d[22 + hi], bp);
        o.w = (int)gather4(u.d[24 + hi], u.d[26 + hi], u.d[28 + hi], u.d[30 + hi], bp);
      }
      if (pp < PX - 1) { o ^= (v4i){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u}; }  // unsigned plane -> signed
      *(v4i *)(lds + pp * plane_bytes + ps * 16) = o;
    }
  };

  if (prefetch) { fetch(s0); }
  for (int64_t st = s0; st < s1; st++) {
    const int64_t m0 = st * 256;
    // (single-wave workgroup: LDS operations of a wave execute in order, so neither the staging writes after the
    // previous step's fragment reads nor the reads below need a barrier -- and a __syncthreads() here would drain the
    // prefetch loads and the stores with s_waitcnt vmcnt(0))
    if (prefetch) {
#pragma unroll
      for (int j = 0; j < SPLMAX; j++) {
        if (j < spl && lane + 64 * j < a.n_slots) { stage_slot(pre[j], lane + 64 * j); }
      }
    } else {
      // W_0: 16-aligned window start of column 0   (T_0 = first + m0*R, off = T_0 - W_0)
      const int64_t W0 = a.first + m0 * R - a.pl.off;
      for (int sl = lane; sl < a.n_slots; sl += 64) {
        v4i raw[sizeof(TIN)];
        const TIN *src = slot_src(W0, sl);
#pragma unroll
        for (int q = 0; q < (int)sizeof(TIN); q++) { raw[q] = ((const v4i *)src)[q]; }
        stage_slot(raw, sl);
      }
    }
    if (prefetch && st + 1 < s1) { fetch(st + 1); }

    // ---- MFMA: plane products of equal weight share an accumulator ----
    v4i acc[kGenMaxPX + kGenMaxPC - 1];
#pragma unroll
    for (int w = 0; w < kGenMaxPX + kGenMaxPC - 1; w++) { acc[w] = (v4i){0, 0, 0, 0}; }
#pragma unroll
    for (int b = 0; b < kMaxNB; b++) {
      if (b < NB) {   // wave-uniform
        const int ps = phys_slot(R * n_col + 4 * b + kg, a);
        v4i X[PX];
#pragma unroll
        for (int pp = 0; pp < PX; pp++) { X[pp] = *(const v4i *)(lds + pp * plane_bytes + ps * 16); }
#pragma unroll
        for (int q = 0; q < kMaxPC; q++) {
          if (q < PC) {
#pragma unroll
            for (int pp = 0; pp < PX; pp++) {
              acc[pp + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[b * kMaxPC + q], X[pp], acc[pp + q], 0, 0, 0);
            }
          }
        }
      }
    }

    // ---- epilogue: lane (n_col, kg) holds outputs m0 + 16 n_col + 4 kg + r ----
    int64_t o[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint64_t y = (uint64_t)a.corr;
#pragma unroll
      for (int w = 0; w < PX + kGenMaxPC - 1; w++) {
        if (w < PX + PC - 1 && w < 8) { y += (uint64_t)(int64_t)acc[w][r] << (8 * w); }
      }
      if (a.out_mode == 1) {
        if (a.out_simple == 2) { o[r] = wrap64((int64_t)y, a.w_int, 1); }
        else if (a.out_simple == 1) { o[r] = wrap64(wrap64((int64_t)y, a.w_int, 1), p.out.W, p.out.S); }
        else { o[r] = requant64(wrap64((int64_t)y, a.w_int, 1), p.in.F, p.out); }
      } else {
        const int64_t accv = wrap64((int64_t)(y << p.lossless_shift), p.acc.W, p.acc.S);
        o[r] = requant64(accv, p.acc.F, p.out);
      }
    }
    if (a.out_vec_ok && m0 + 256 <= a.n_out && p.out_eb >= 4) {
      // The step's 256 outputs are contiguous in the row: pass them through an XOR-swizzled LDS tile (conflict-free
      // ds_write_b128 per column group and ds_read_b128 per row run) and write whole 128-byte lines, 1 KB per
      // instruction.  (8-byte pieces at 32-byte stride straight from the lanes: 17.2 ms on config 3, 11.2 ms with
      // the stores compiled out.)
      unsigned char *ob = lds + a.obuf_off;
      typedef long v2l __attribute__((ext_vector_type(2)));
      if (p.out_eb == 8) {
        const int L = 8 * n_col + 2 * kg;                       // 16-byte slot of outputs r = 0, 1; r = 2, 3 follow
        const int sw = n_col & 7;
        *(v2l *)(ob + ((L ^ sw) * 16)) = (v2l){o[0], o[1]};
        *(v2l *)(ob + (((L + 1) ^ sw) * 16)) = (v2l){o[2], o[3]};
        int64_t *dst = (int64_t *)p.y + (int64_t)ch * p.out_stride + m0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int P = 64 * k + lane;
          const v4i val = *(const v4i *)(ob + ((P ^ ((P >> 3) & 7)) * 16));
          *(v4i *)((char *)dst + 16 * P) = val;
        }
      } else {
        const int L = 4 * n_col + kg;
        *(v4i *)(ob + ((L ^ ((n_col >> 1) & 3)) * 16)) = (v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
        int32_t *dst = (int32_t *)p.y + (int64_t)ch * p.out_stride + m0;
        const v4i val = *(const v4i *)(ob + ((lane ^ ((lane >> 3) & 3)) * 16));
        *(v4i *)((char *)dst + 16 * lane) = val;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t m = m0 + 16 * n_col + 4 * kg + r;
        if (m < a.n_out) { store_raw(p.y, (int64_t)ch * p.out_stride + m, p.out_eb, o[r]); }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fast variant for the shapes the BASELINE configurations use (dispatch table in launch_fir_gen): every loop bound is
// a template parameter, the output conversion is the branch-free shift / clamp / wrap form above and the launch covers
// only chunks whose steps are complete, so the step loop is ONE basic block: the next step's slots stay in flight
// (counted s_waitcnt vmcnt(k)) while this step multiplies, converts and writes out.  The general kernel above had
// ~1500 basic blocks and an s_waitcnt vmcnt(0) in front of every MFMA group -- the prefetch never overlapped anything.
//   PCT / NBT: coefficient digits / K blocks compiled in (fragments beyond the plan's pc / nb are zero)
//   NLD: loads per lane and step -- 16-byte pieces for 2- and 4-byte samples (ceil(n_slots * sizeof(TIN) / 64): the former
//        whole-slot count made config 3 issue 12 loads and 48 staging writes per step where 9 and 36 cover its window), slots for
//        8-byte samples        OEB: output container bytes
#ifndef ACDSP_GEN_FAST_WAVES
#define ACDSP_GEN_FAST_WAVES 2
#endif
template <typename TIN, int PX, int PCT, int NBT, int NLD, int OEB>
__global__ void __launch_bounds__(64, ACDSP_GEN_FAST_WAVES) fir_gen_fast_kernel(FirParams p, const v4i *__restrict__ frag, GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [PX][phys slots][16] + output tile
  const int lane = threadIdx.x;
  const int n_col = lane & 15, kg = lane >> 4;
  int bx, ch;
  xcd_remap(a.xcd_map, bx, ch);
  const int NB = a.pl.nb, PC = a.pl.pc, R = a.pl.R;
  const int plane_bytes = a.obuf_off / PX;

  v4i A[NBT][PCT];
#pragma unroll
  for (int b = 0; b < NBT; b++) {
#pragma unroll
    for (int q = 0; q < PCT; q++) {
      A[b][q] = (b < NB && q < PC) ? frag[((size_t)q * NB + b) * 64 + lane] : (v4i){0, 0, 0, 0};
    }
  }
  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  const TIN *hrow = (const TIN *)p.hist + (int64_t)ch * p.hl + p.hl;
  const int64_t s0 = ((int64_t)bx + a.chunk0) * a.steps_per_wave;
  const int64_t s1 = s0 + a.steps_per_wave;            // the launch covers complete chunks only

  // Loads are coalesced over the window, not over a lane's slot: load k of a lane is 16-byte piece lane + 64 k of the step's
  // window (a slot of 16 samples is S = sizeof(TIN) pieces), so one instruction reads 1 KB contiguous instead of 64 pieces
  // 16 S bytes apart (that form touched every 128-byte line of the window S times: 5.3 TB/s on config 3 where a plain
  // 4 : 1 read / write stream reaches 5.8, tools/copy_probe).  The byte planes of a piece are 16 / S bytes each.
  constexpr int S = (int)sizeof(TIN);
  constexpr bool COAL = S <= 4;
  constexpr int NPC = COAL ? NLD : 1;                         // pieces (COAL) per lane and step
  constexpr int SPL = COAL ? 1 : NLD;                         // slots per lane (8-byte samples)
  v4i pre[COAL ? NPC : SPL][COAL ? 1 : sizeof(TIN)];
  int sl_of[SPL], ps_of[SPL];
  int pc_off[COAL ? NPC : 1], pc_ps[COAL ? NPC : 1];         // COAL: sample offset of the piece in the window, LDS byte offset of its planes
#pragma unroll
  for (int j = 0; j < SPL; j++) {
    sl_of[j] = (lane + 64 * j < a.n_slots) ? lane + 64 * j : a.n_slots - 1;   // surplus lanes repeat the last slot
    ps_of[j] = phys_slot(sl_of[j], a) * 16;
  }
  if constexpr (COAL) {
#pragma unroll
    for (int k = 0; k < NPC; k++) {
      int slot = (lane + 64 * k) / S;
      const int sub = lane % S;
      if (slot >= a.n_slots) { slot = a.n_slots - 1; }         // surplus pieces repeat a piece of the last slot
      pc_off[k] = 16 * slot + (16 / S) * sub;
      pc_ps[k] = phys_slot(slot, a) * 16 + (16 / S) * sub;
    }
  }
  auto fetch = [&](int64_t st) {
    const int64_t W0 = a.first + st * 256 * R - a.pl.off;
    if constexpr (COAL) {
#pragma unroll
      for (int k = 0; k < NPC; k++) {
        const int64_t t = W0 + pc_off[k];
        const TIN *src = (t < 0) ? hrow + t : xrow + ((t < a.n16) ? t : 0);
        pre[k][0] = *(const v4i *)src;
      }
    } else {
#pragma unroll
      for (int j = 0; j < SPL; j++) {
        const int64_t t = W0 + 16 * (int64_t)sl_of[j];
        const TIN *src = (t < 0) ? hrow + t : xrow + ((t < a.n16) ? t : 0);
#pragma unroll
        for (int q = 0; q < (int)sizeof(TIN); q++) { pre[j][q] = ((const v4i *)src)[q]; }
      }
    }
  };
  // COAL: byte plane pp of one 16-byte piece -> 16 / S bytes
  auto stage_piece = [&](const v4i &v, int ps) {
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      if constexpr (S == 2) {
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        unsigned lo = __builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, sel), hi = __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, sel);
        if (pp < PX - 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        *(v2u *)(lds + pp * plane_bytes + ps) = (v2u){lo, hi};
      } else {
        unsigned w = gather4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w, pp);
        if (pp < PX - 1) { w ^= 0x80808080u; }
        *(unsigned *)(lds + pp * plane_bytes + ps) = w;
      }
    }
  };
  auto stage_slot = [&](const v4i (&raw)[sizeof(TIN)], int ps16) {
    union { v4i v[sizeof(TIN)]; unsigned d[4 * sizeof(TIN)]; } u;
#pragma unroll
    for (int q = 0; q < (int)sizeof(TIN); q++) { u.v[q] = raw[q]; }
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      v4i o;
      if (sizeof(TIN) == 2) {
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        o.x = (int)__builtin_amdgcn_perm(u.d[1], u.d[0], sel); o.y = (int)__builtin_amdgcn_perm(u.d[3], u.d[2], sel);
        o.z = (int)__builtin_amdgcn_perm(u.d[5], u.d[4], sel); o.w = (int)__builtin_amdgcn_perm(u.d[7], u.d[6], sel);
      } else if (sizeof(TIN) == 4) {
        o.x = (int)gather4(u.d[0], u.d[1], u.d[2], u.d[3], pp); o.y = (int)gather4(u.d[4], u.d[5], u.d[6], u.d[7], pp);
        o.z = (int)gather4(u.d[8], u.d[9], u.d[10], u.d[11], pp); o.w = (int)gather4(u.d[12], u.d[13], u.d[14], u.d[15], pp);
      } else {
        const int hi = pp >> 2, bp = pp & 3;
        o.x = (int)gather4(u.d[0 + hi], u.d[2 + hi], u.d[4 + hi], u.d[6 + hi], bp);
        o.y = (int)gather4(u.d[8 + hi], u.d[10 + hi], u.d[12 + hi], u.d[14 + hi], bp);
        o.z = (int)gather4(u.d[16 + hi], u.d[18 + hi], u.d[20 + hi], u.d[22 + hi], bp);
        o.w = (int)gather4(u.d[24 + hi], u.d[26 + hi], u.d[28 + hi], u.d[30 + hi], bp);
      }
      if (pp < PX - 1) { o ^= (v4i){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u}; }
      *(v4i *)(lds + pp * plane_bytes + ps16) = o;
    }
  };
  int xs_of[NBT];   // LDS byte offset of this lane's X fragment per K block
#pragma unroll
  for (int b = 0; b < NBT; b++) { xs_of[b] = phys_slot(R * n_col + 4 * b + kg, a) * 16; }
  unsigned char *ob = lds + a.obuf_off;
  char *yrow = (char *)p.y + (int64_t)ch * p.out_stride * OEB;

  // write-out of a finished step: the tile holds its 256 outputs in row order (swizzled 16-byte slots)
  auto flush = [&](int64_t st) {
    const int64_t m0 = st * 256;
    if (OEB == 8) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int P = 64 * k + lane;
        const v4i val = *(const v4i *)(ob + ((P ^ ((P >> 3) & 7)) * 16));
        ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 8 + 16 * P));
      }
    } else if (OEB == 4) {
      const v4i val = *(const v4i *)(ob + ((lane ^ ((lane >> 3) & 3)) * 16));
      ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 4 + 16 * lane));
    } else {
      const int P = lane & 31;                       // both wave halves store the same 512 bytes: no exec-mask branch
      const int x = (P >> 3) & 3;                    // 8-byte units are XORed with x (below): the pair of a 16-byte piece moves by x >> 1, swaps by x & 1
      v4i val = *(const v4i *)(ob + (P ^ (x >> 1)) * 16);
      if (x & 1) { val = (v4i){val.z, val.w, val.x, val.y}; }
      ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 2 + 16 * P));     // (pairing two steps into one 1 KB store: +0.2 %, profiles/r3_ab_store_width.txt -- not kept)
    }
  };
  // One step.  Program order: stage step st (its slots were fetched one step ago), write out step st-1, fetch step
  // st+1, multiply, convert into the LDS tile.  The stores therefore sit BEFORE the fetch in the stream: the
  // s_waitcnt in front of the next staging waits for loads that are the youngest VMEM operations, and the stores get a
  // whole step to drain.
  auto body = [&](int64_t st, auto first_c) {
    if constexpr (COAL) {
#pragma unroll
      for (int k = 0; k < NPC; k++) { stage_piece(pre[k][0], pc_ps[k]); }
    } else {
#pragma unroll
      for (int j = 0; j < SPL; j++) { stage_slot(pre[j], ps_of[j]); }
    }
    if (!decltype(first_c)::value) { flush(st - 1); }
    fetch(st + 1 < s1 ? st + 1 : st);   // the last step re-fetches itself: no branch in the loop

    v4i acc[PX + PCT - 1];
#pragma unroll
    for (int w = 0; w < PX + PCT - 1; w++) { acc[w] = (v4i){0, 0, 0, 0}; }
#pragma unroll
    for (int b = 0; b < NBT; b++) {
      v4i X[PX];
#pragma unroll
      for (int pp = 0; pp < PX; pp++) { X[pp] = *(const v4i *)(lds + pp * plane_bytes + xs_of[b]); }
#pragma unroll
      for (int q = 0; q < PCT; q++) {
#pragma unroll
        for (int pp = 0; pp < PX; pp++) {
          acc[pp + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[b][q], X[pp], acc[pp + q], 0, 0, 0);
        }
      }
    }

    // lane (n_col, kg) holds outputs 16 n_col + 4 kg + r of the step
    int64_t o[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint64_t y = (uint64_t)a.corr;
#pragma unroll
      for (int w = 0; w < PX + PCT - 1; w++) {
        if (w < 8) { y += (uint64_t)(int64_t)acc[w][r] << (8 * w); }
      }
      int64_t v = (int64_t)(y << a.e_ls);
      v = (int64_t)((uint64_t)v << a.e_ka) >> a.e_ka;
      v = (int64_t)((uint64_t)((v + a.e_rnd) >> a.e_rs) << a.e_ls2);
      v = v < a.e_lo ? a.e_lo : (v > a.e_hi ? a.e_hi : v);
      o[r] = (int64_t)((uint64_t)v << a.e_ko) >> a.e_ko;
    }
    typedef long v2l __attribute__((ext_vector_type(2)));
    if (OEB == 8) {
      const int L = 8 * n_col + 2 * kg, sw = n_col & 7;
      *(v2l *)(ob + ((L ^ sw) * 16)) = (v2l){o[0], o[1]};
      *(v2l *)(ob + (((L + 1) ^ sw) * 16)) = (v2l){o[2], o[3]};
    } else if (OEB == 4) {
      const int L = 4 * n_col + kg;
      *(v4i *)(ob + ((L ^ ((n_col >> 1) & 3)) * 16)) = (v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    } else {
      typedef short v4s_ __attribute__((ext_vector_type(4)));
      // unit 4 n_col + kg, low two bits XOR n_col >> 2: the 16 lanes of a ds_write_b64 bank pass (one kg) cover 16 distinct units mod 16
      *(v4s_ *)(ob + ((4 * n_col + kg) ^ ((n_col >> 2) & 3)) * 8) = (v4s_){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
    }
  };

  fetch(s0);
  body(s0, std::integral_constant<bool, true>());
  for (int64_t st = s0 + 1; st < s1; st++) { body(st, std::integral_constant<bool, false>()); }
  flush(s1 - 1);
}

template <typename TIN, int PX, int PCT, int NBT, int NLD, int OEB>
static hipError_t launch_fast(dim3 grid, size_t lds_bytes, hipStream_t s, const FirParams &p, const v4i *frag, const GenArgs &a) {
  hipError_t e = hipFuncSetAttribute((const void *)fir_gen_fast_kernel<TIN, PX, PCT, NBT, NLD, OEB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) { return e; }
  hipLaunchKernelGGL((fir_gen_fast_kernel<TIN, PX, PCT, NBT, NLD, OEB>), grid, dim3(64), lds_bytes, s, p, frag, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ring variant of the fast kernel (round 4): the decimating shapes of the BASELINE rows (R = 8 / 16 on int16 / int32 rows).
//
// The fast kernel above re-stages the whole window of every step: n_slots = 15 R + 4 NBT slots where the step advances 16 R, so the
// tail of a window is fetched twice (L2), the surplus lanes of the last load repeat a piece, and the loads start on 16-sample, not
// 128-byte, boundaries.  Here the byte planes live in an LDS RING of two steps: every step loads exactly its own 16 R new slots -- R S / 4
// full 1 KB wave-loads on 128-byte lines -- and a chunk reads every input byte ONCE (plus a halo of H <= 12 slots at its start).  With
// that the window loads can take the non-temporal policy (they bypass the vector L1, which the old form's re-reads needed).
//
//   ring position q (slots, relative to the chunk's line-aligned origin C0)  ->  q mod 2 ADV, linear range [0, 2 ADV + H):
//     prime   [0, H)                      loaded once at the chunk start
//     step k  [H + k ADV, H + (k+1) ADV)  -> even k: [H, H + ADV); odd k: [H + ADV, H + 2 ADV), whose last H slots are the MIRROR of
//                                            [0, H) and are written to both places (the wrap of the ring without modular addressing)
//     window of step k = [delta + (k & 1) ADV, ... + 15 R + 4 NBT), delta = misalignment of the window start against its line (< 8 / S)
//   The chunk is fully unrolled (SPW steps), so every ring offset is an immediate beside one base register per access class.
//   The slot -> LDS map is gen_slot_map's (two empty slots after every R), which commutes with the ring because ADV and the slots
//   of one 1 KB load are multiples of R.
//   PF = steps the loads run ahead (1 or 2); NT = non-temporal window loads.
//   FB = the output tiles of the whole chunk wait in LDS and leave in one burst at its end (SPW tiles instead of one).
//   LZ = class B (lossy accumulator, AC_TRN / AC_RND into AC_WRAP -- reference ac_fir_prog_coeffs.h:147-227 at the types of its
//        own testbench, <28,6> x <23,7> into <64,32>).  sum_k Q(p_k) = (sum_k p_k + N h - sum_k ((p_k + h) mod 2^s)) / 2^s: the first sum is
//        the exact product of the matrix cores, and the dropped bits of a tap only need the low s bits of its operands.  The low s bits of
//        every sample are staged a second time as a plain ring of 16-bit fields (no holes), so that the four consecutive outputs of a lane
//        read the four samples of a tap with ONE (2-byte aligned) 8-byte LDS read; per tap and four outputs: v_pk_mad_u16 x 2, v_and x 2,
//        add x 2 (+ v_pk_add_u16 x 2 for the pre-add of a folded pair), coefficients as uniform LDS reads.  R = 1 only.
template <typename TIN, int PX, int PCT, int NBT, int R, int OEB, int SPW, int PF, bool NT, bool FB, bool LZ = false>
__global__ void __launch_bounds__(64, 2) fir_gen_ring_kernel(FirParams p, const v4i *__restrict__ frag, GenArgs a) {
  constexpr int S = (int)sizeof(TIN);
  constexpr int LS = 8 / S;                                   // slots per 128-byte line
  constexpr int ADV = 16 * R, NSL = 15 * R + 4 * NBT;
  constexpr int H = (LS - 1 + NSL - ADV + LS - 1) / LS * LS;  // halo slots (line multiple): covers any delta
  constexpr int M = (R * S) % 4 == 0 ? 1 : 2;                 // steps per load group: a plain (R = 1) FIR on int16 advances half a 1 KB load per step, R = 5 two and a half
  constexpr int NLD = R * S * M / 4;                          // 1 KB loads per group of M steps
  constexpr int PPB = 16 / S;                                 // samples of a 16-byte piece = bytes of one of its planes
  constexpr int SPK = 64 / S;                                 // slots per 1 KB load
  static_assert(S == 2 || S == 4, "ring kernel: 2- and 4-byte samples");
  // odd factors above 1 (R = 7: the reference's CIC testbench) and even ones that do not divide the slots of a 1 KB load (R = 10) keep the identity
  // slot map -- gen_slot_map leaves odd factors alone too: one or two extra LDS cycles per fragment read -- so a load need not hold whole groups of R slots
  constexpr bool HOLES = (R % 2 == 0 && SPK % R == 0) || R == 1;   // (even factors that do not divide a 1 KB load -- R = 10: R = 2 mod 4 is conflict-free without holes anyway)
  // -DACDSP_GEN_ODDPAD (round 6 A/B, profiles/r6_lds_ab.txt): odd factors get (2 - R) mod 4 empty slots per R, i.e. a column stride of 2 mod 4 slots like the
  // conflict-free even shapes (tools/lds_slot_map_check.py: R = 7 with three empty slots per seven reads 0 / 2 / 0 / 0 extra cycles per K block, the identity
  // map 4 / 4 / 4 / 4).  A 1 KB load then no longer holds whole groups of R slots, so the staging offsets of a lane's pieces are per-load registers, not immediates.
#ifdef ACDSP_GEN_ODDPAD
  constexpr int PADK = HOLES ? 2 : ((R % 2 == 1 && R > 1 && M == 1) ? (2 - R % 4 + 4) % 4 : 0);
#else
  constexpr int PADK = HOLES ? 2 : 0;
#endif
  constexpr bool GENPAD = !HOLES && PADK > 0;
  static_assert((!HOLES || SPK % R == 0) && H <= ADV && S * H <= 64 && NLD * SPK == M * ADV && NLD >= 1 && SPW % M == 0, "ring geometry");
  constexpr int KSTEP = HOLES ? (SPK + 2 * (SPK / R)) * 16 : SPK * 16;   // LDS bytes from a piece of load k to the same lane's piece of load k + 1
  constexpr int PADV = (ADV + PADK * (ADV / R)) * 16;                   // LDS bytes a step advances
  constexpr int RING = (SPW < 2 ? SPW : 2) * ADV + H;         // chunks of one or two steps never wrap
  // The two empty slots per R sit right in FRONT of ring slots H, H + R, ...: a staging store pass covers eight consecutive slots from
  // H + 8 j on, and must not straddle a hole (with the holes at multiples of R, as in gen_slot_map, every pass of this layout did: 2-way
  // conflicts on two of its eight slots, SQ_LDS_BANK_CONFLICT 31 % of SQ_LDS_IDX_ACTIVE in the first profile of this kernel).  The reads
  // only need one hole per R slots, wherever it sits (tools/lds_slot_map_check.py, and the replay in profiles/r4_ring_sweep.txt).
  constexpr int PH = PADK ? (R - H % R) % R : 0;
  constexpr int PLANE = (RING + PADK * ((RING + PH) / R) + 2) * 16;   // + one dump slot (mirror writes of the lanes that have none)
  constexpr int DUMP = PLANE - 16;
  constexpr int TILE = 256 * OEB;
  static_assert(!LZ || R == 1, "class-B residues: plain FIR only");
  constexpr int LZP = LZ ? (RING + 1) * 32 : 0;               // low-bit ring: 16 samples x 2 bytes per slot, + one dump slot
  constexpr int LZ_DUMP = RING * 32, KSTEP_L = SPK * 32, PADV_L = ADV * 32;
  // (round 6 A/B, -DACDSP_LZ_TWO_COPY, NOT the default) a SECOND copy of the low-bit ring, one dword further on, for the columns 8 .. 15.  The residue loops
  // read aligned dwords at byte 32 n_col + 8 kg + ...: even dword banks only, so the 64 lanes of a read share 32 banks -- columns n and n + 8 collide, the
  // 52 % SQ_LDS_BANK_CONFLICT of the class-B row (LDS 66 % busy beside a VALU at 0.82 of its issue rate).  With the upper columns on the shifted copy (staged
  // with dword writes: it sits off the 16-byte grid) the conflicts halve -- 2.64e8 -> 0.98e8 of 5.1e8 -> 3.6e8 active cycles -- and the kernel is 8 % SLOWER
  // (1.395 -> 1.505 ms, same box, alternating processes): the two base pointers cost 9.5 % more VALU instructions in a VALU-bound loop.  profiles/r6_lds_ab.txt
#ifdef ACDSP_LZ_TWO_COPY
  constexpr int LZB = LZ ? LZP + 16 : 0;                     // byte distance of the second copy's origin (the +4 of the shift is added where it is used)
#else
  constexpr int LZB = 0;
#endif
  __shared__ __attribute__((aligned(16))) unsigned char lds[PX * PLANE + (FB ? SPW : 1) * TILE + LZP + (LZ ? 4 * kLossyTabWords : 0) + (LZB ? LZP + 32 : 0)];
  const int lane = threadIdx.x;
  const int n_col = lane & 15, kg = lane >> 4;
  int bx, ch;
  xcd_remap(a.xcd_map, bx, ch);
  unsigned char *const lzr = lds + PX * PLANE + (FB ? SPW : 1) * TILE;   // low-bit ring, then the 128 coefficient words
  unsigned char *const lzr2 = lzr + LZP + (LZ ? 4 * kLossyTabWords : 0) + 4;   // the second copy: LZB + 4 bytes behind the first once the table is skipped
  if constexpr (LZ) {
#pragma unroll
    for (int q = 0; q < kLossyTabWords / 64; q++) { ((uint32_t *)(lzr + LZP))[lane + 64 * q] = a.lz_tab[lane + 64 * q]; }
  }
  const int NB = a.pl.nb, PC = a.pl.pc;
  auto phys = [](int s) { return PADK ? s + PADK * ((s + PH) / R) : s; };

  v4i A[NBT][PCT];
#pragma unroll
  for (int b = 0; b < NBT; b++) {
#pragma unroll
    for (int q = 0; q < PCT; q++) { A[b][q] = (b < NB && q < PC) ? frag[((size_t)q * NB + b) * 64 + lane] : (v4i){0, 0, 0, 0}; }
  }
  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  const TIN *hrow = (const TIN *)p.hist + (int64_t)ch * p.hl + p.hl;
  const int64_t s0 = ((int64_t)bx + a.chunk0) * SPW;
  // C0: sample index of the chunk's ring origin = the first window's start, moved down onto its 128-byte line
  const int64_t c0 = a.first - a.pl.off - 16 * (int64_t)a.ring_delta + s0 * (256 * R);
  const bool interior = c0 >= 0 && c0 + 16 * (int64_t)(H + SPW * ADV) <= a.n16;   // every load of the chunk lies inside the call's samples

  const int pl_lane = lane < S * H ? lane : S * H - 1;          // prime: S H pieces; the other lanes repeat the last one
  const int pr_off = phys(pl_lane / S) * 16 + (pl_lane % S) * PPB;
  const int st_base = phys(H + lane / S) * 16 + (lane % S) * PPB;
  const int mir_off = lane >= 64 - S * H ? (GENPAD ? phys(H + lane / S - SPK) * 16 + (lane % S) * PPB : st_base - KSTEP) : DUMP;
  int st_q[GENPAD ? NLD : 1];     // GENPAD: LDS byte offset of this lane's piece of load q (no constant stride between the loads)
#pragma unroll
  for (int q = 0; q < (GENPAD ? NLD : 1); q++) { st_q[q] = phys(H + lane / S + q * SPK) * 16 + (lane % S) * PPB; }
  auto st_of = [&](int q) { if constexpr (GENPAD) { return st_q[q]; } else { return st_base + q * KSTEP; } };
  // the same three places in the low-bit ring (linear slots of 32 bytes)
  const int pr_lin = (pl_lane / S) * 32 + (pl_lane % S) * PPB * 2;
  const int st_lin = (H + lane / S) * 32 + (lane % S) * PPB * 2;
  const int mir_lin = lane >= 64 - S * H ? st_lin - KSTEP_L : LZ_DUMP;
  int xs[NBT];
#pragma unroll
  for (int b = 0; b < NBT; b++) { xs[b] = phys(a.ring_delta + R * n_col + 4 * b + kg) * 16; }

  auto ld = [&](const v4i *ptr) -> v4i {
    if constexpr (NT) { return __builtin_nontemporal_load(ptr); } else { return *ptr; }
  };
  // piece `pc` (16 bytes) of the chunk: sample offset tp from C0
  auto piece = [&](int64_t tp, auto fast_c) -> v4i {
    if constexpr (decltype(fast_c)::value) {
      return ld((const v4i *)(xrow + c0 + tp));
    } else {
      const int64_t t = c0 + tp;
      const int64_t th = t < -(int64_t)p.hl ? -(int64_t)p.hl : t;     // before the history: slots below the first window, never read
      const TIN *src = (t < 0) ? hrow + th : xrow + ((t < a.n16) ? t : 0);
      return ld((const v4i *)src);
    }
  };
  auto stage_piece = [&](const v4i &v, int off, int loff) {
    if constexpr (LZ) {
      if constexpr (S == 2) {     // 8 samples: the dwords already hold two 16-bit fields
        const v4i lw = (v4i){(int)((unsigned)v.x & a.lz_m2), (int)((unsigned)v.y & a.lz_m2), (int)((unsigned)v.z & a.lz_m2), (int)((unsigned)v.w & a.lz_m2)};
        *(v4i *)(lzr + loff) = lw;
        if constexpr (LZB != 0) {
          int *c2 = (int *)(lzr2 + loff);
          c2[0] = lw.x; c2[1] = lw.y; c2[2] = lw.z; c2[3] = lw.w;
        }
      } else {                    // 4 samples: low halves of two dwords side by side
        typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
        const v2u_ lw2 = (v2u_){__builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, 0x05040100u) & a.lz_m2,
                                __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, 0x05040100u) & a.lz_m2};
        *(v2u_ *)(lzr + loff) = lw2;
        if constexpr (LZB != 0) {
          unsigned *c2 = (unsigned *)(lzr2 + loff);
          c2[0] = lw2.x; c2[1] = lw2.y;
        }
      }
    }
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      if constexpr (S == 2) {
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        unsigned lo = __builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, sel), hi = __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, sel);
        if (pp < PX - 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        *(v2u *)(lds + pp * PLANE + off) = (v2u){lo, hi};
      } else {
        unsigned w = gather4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w, pp);
        if (pp < PX - 1) { w ^= 0x80808080u; }
        *(unsigned *)(lds + pp * PLANE + off) = w;
      }
    }
  };
  unsigned char *ob0 = lds + PX * PLANE;
  char *yrow = (char *)p.y + (int64_t)ch * p.out_stride * OEB;
  auto flush = [&](int64_t st, int k) {   // as in fir_gen_fast_kernel; k = the step's tile (FB)
    const int64_t m0 = st * 256;
    unsigned char *ob = ob0 + (FB ? k * TILE : 0);
    if (OEB == 8) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int P = 64 * j + lane;
        const v4i val = *(const v4i *)(ob + ((P ^ ((P >> 3) & 7)) * 16));
        ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 8 + 16 * P));
      }
    } else if (OEB == 4) {
      const v4i val = *(const v4i *)(ob + ((lane ^ ((lane >> 3) & 3)) * 16));
      ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 4 + 16 * lane));
    } else {
      // 512 bytes per step: alone, both wave halves store the same bytes (no exec-mask branch); in the chunk-end burst (FB) the
      // halves take steps k and k + 1 -- one 1 KB store (called for even k only)
      const int P = lane & 31;
      const int x = (P >> 3) & 3;
      const int half = FB ? (lane >> 5) : 0;
      v4i val = *(const v4i *)(ob + half * TILE + (P ^ (x >> 1)) * 16);
      if (x & 1) { val = (v4i){val.z, val.w, val.x, val.y}; }
      ACDSP_GEN_ST(val, (v4i *)(yrow + m0 * 2 + half * TILE + 16 * P));
    }
  };

  auto chunk = [&](auto fast_c) {
    v4i pre[PF][NLD];
    auto fetch = [&](int j) {      // the 16 R M new slots of step group j
#pragma unroll
      for (int k = 0; k < NLD; k++) { pre[j % PF][k] = piece(16 * (int64_t)(H + j * M * ADV) + (int64_t)(lane + 64 * k) * PPB, fast_c); }
    };
    constexpr int NG = SPW / M;                                // load groups of the chunk (M = 1: the steps)
    const v4i prm = piece((int64_t)pl_lane * PPB, fast_c);
#pragma unroll
    for (int j = 0; j < PF && j < NG; j++) { fetch(j); }
    stage_piece(prm, pr_off, pr_lin);
#pragma unroll
    for (int k = 0; k < SPW; k++) {
      const int par = (k & 1) * PADV;
      const int par_l = (k & 1) * PADV_L;
      constexpr int dummy_m = M;
      const int g = k / dummy_m;
      if (k % M == 0) {   // (M = 2: one load covers the even step's region and the odd one's behind it)
#pragma unroll
        for (int q = 0; q < NLD; q++) { stage_piece(pre[g % PF][q], st_of(q) + par, st_lin + q * KSTEP_L + par_l); }
      }
      if ((k & 1) && k + 1 < SPW) { stage_piece(pre[g % PF][NLD - 1], mir_off, mir_lin); }   // the mirror is read by step k + 1
      if (!FB && k > 0) { flush(s0 + k - 1, 0); }
      if (k % M == 0 && g + PF < NG) { fetch(g + PF); }
      asm volatile("" ::: "memory");   // keep the loads in front of the step's arithmetic (see cascade_kernel)

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
      // class B: the dropped bits of every tap, for this lane's four outputs (fields: outputs 0 | 1 in rA, 2 | 3 in rB).
      // The low-bit ring holds sample p as a 16-bit field at byte 2 p.  An LDS access off its natural alignment is replayed lane by lane
      // (~75 cycles per wave read measured, profiles/r5_lds_align.txt; the first form of this loop read 8 bytes per tap at 2-byte alignment
      // and ran 13 lane-ops per tap and output), so every read here is an aligned dword of a sliding three-dword window, two taps
      // per iteration: the "direct" slot D takes its four samples as (w1, w2), the "shifted" slot S one sample earlier through two
      // v_alignbit.  Which tap sits in which slot depends on the parity of the window against the taps -- host-side (fir_gen_lossy_table),
      // as slot coefficients; a slot without a tap has coefficient 0 and contributes exactly h, which K counts.
      unsigned rA = 0, rB = 0;
      unsigned rT[4] = {0, 0, 0, 0};   // 32-bit totals per output: the packed fields are emptied into them every lz_flush iterations
      if constexpr (LZ) {
        typedef unsigned short v2us_ __attribute__((ext_vector_type(2)));
        typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
        // byte address of the sample of output r = 0, tap 0: ring position 16 (delta + n_col) + off + 4 kg  (+ the step's parity)
        const unsigned char *p0 = ((LZB != 0 && n_col >= 8) ? lzr2 : lzr) + 2 * (16 * (a.ring_delta + n_col) + a.pl.off + 4 * kg) + par_l;
        const v2u_ *ctab = (const v2u_ *)(lzr + LZP);
        const v2us_ h2 = __builtin_bit_cast(v2us_, a.lz_h2);
        auto ld32 = [&](const unsigned char *q) -> unsigned { return *(const unsigned *)q; };
        auto term = [&](unsigned f, unsigned c2) -> unsigned {
          const v2us_ m = __builtin_bit_cast(v2us_, f) * __builtin_bit_cast(v2us_, c2) + h2;
          return __builtin_bit_cast(unsigned, m) & a.lz_m2;
        };
        auto sh = [](unsigned lo, unsigned hi) -> unsigned { return __builtin_amdgcn_alignbit(hi, lo, 16); };   // fields (lo.hi, hi.lo)
        auto fold = [&](unsigned x, unsigned y, auto neg_c) -> unsigned {
          if constexpr (decltype(neg_c)::value) { return __builtin_bit_cast(unsigned, (v2us_)(__builtin_bit_cast(v2us_, x) - __builtin_bit_cast(v2us_, y))); }
          else { return __builtin_bit_cast(unsigned, (v2us_)(__builtin_bit_cast(v2us_, x) + __builtin_bit_cast(v2us_, y))); }
        };
        // 16-bit fields hold two slots of up to 2^s - 1 per iteration: emptied into the 32-bit totals before they can wrap (never inside a
        // loop for s <= 8 at these tap counts; every iteration at s = 15)
        int cnt = 0;
        auto flush = [&]() { rT[0] += rA & 0xffffu; rT[1] += rA >> 16; rT[2] += rB & 0xffffu; rT[3] += rB >> 16; rA = 0; rB = 0; cnt = 0; };
        auto pairs = [&](auto neg_c, auto vary_c, auto fl_c) {
          const unsigned char *wp = p0 + 2 * a.lz_p_woff, *vp = p0 + 2 * a.lz_p_voff;
          unsigned w1 = ld32(wp + 4), w2 = ld32(wp + 8), v0 = ld32(vp), v1 = ld32(vp + 4);
          for (int m = 0; m < a.lz_p_n; m++) {
            const unsigned w0 = ld32(wp - 4 * m), v2 = ld32(vp + 4 * m + 8);
            const v2u_ c = ctab[m];
            const unsigned cS = c.x, cD = c.y;
            unsigned fS0, fS1, fD0, fD1;
            if constexpr (decltype(vary_c)::value) {   // even tap count: the mirror of the shifted slot is direct (v1, v2), of the direct slot shifted
              fS0 = fold(sh(w0, w1), v1, neg_c); fS1 = fold(sh(w1, w2), v2, neg_c);
              fD0 = fold(w1, sh(v0, v1), neg_c); fD1 = fold(w2, sh(v1, v2), neg_c);
            } else {
              fS0 = fold(sh(w0, w1), sh(v0, v1), neg_c); fS1 = fold(sh(w1, w2), sh(v1, v2), neg_c);
              fD0 = fold(w1, v0, neg_c); fD1 = fold(w2, v1, neg_c);
            }
            rA += term(fS0, cS) + term(fD0, cD);
            rB += term(fS1, cS) + term(fD1, cD);
            w2 = w1; w1 = w0; v0 = v1; v1 = v2;
            if constexpr (decltype(fl_c)::value) { if (++cnt == a.lz_flush) { flush(); } }
          }
        };
        typedef std::integral_constant<bool, true> T_; typedef std::integral_constant<bool, false> F_;
        // (the counter and its branch cost the loops 25 % when they were unconditional: only sets that need a flush inside a loop pay for it)
        const bool fl = a.lz_flush < a.lz_p_n + a.lz_s_n;
        if (a.lz_p_n > 0) {
          if (fl) {
            if (a.lz_neg) { if (a.lz_var_y) { pairs(T_(), T_(), T_()); } else { pairs(T_(), F_(), T_()); } }
            else { if (a.lz_var_y) { pairs(F_(), T_(), T_()); } else { pairs(F_(), F_(), T_()); } }
          } else {
            if (a.lz_neg) { if (a.lz_var_y) { pairs(T_(), T_(), F_()); } else { pairs(T_(), F_(), F_()); } }
            else { if (a.lz_var_y) { pairs(F_(), T_(), F_()); } else { pairs(F_(), F_(), F_()); } }
          }
        }
        auto singles = [&](auto fl_c) {
          const unsigned char *wp = p0 + 2 * a.lz_s_woff;
          const v2u_ *tab = ctab + a.lz_p_n;
          unsigned w1 = ld32(wp + 4), w2 = ld32(wp + 8);
#pragma unroll 2
          for (int m = 0; m < a.lz_s_n; m++) {
            const unsigned w0 = ld32(wp - 4 * m);
            const v2u_ c = tab[m];
            const unsigned cS = c.x, cD = c.y;
            rA += term(sh(w0, w1), cS) + term(w1, cD);
            rB += term(sh(w1, w2), cS) + term(w2, cD);
            w2 = w1; w1 = w0;
            if constexpr (decltype(fl_c)::value) { if (++cnt == a.lz_flush) { flush(); } }
          }
        };
        if (fl) { singles(T_()); } else { singles(F_()); }
        flush();
      }
      int64_t o[4];
      constexpr int NACC = PX + PCT - 1, NPR = (NACC + 1) / 2;
      if (a.pw) {
        // |acc[w]| small enough (host: dig_abs) that acc[2m] + (acc[2m+1] << 8) is exact in int32: pairs first, then 32-bit limbs with
        // carries -- 8 instead of ~24 VALU instructions per output against a 64-bit shift-and-add per plane
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int pr[NPR];
#pragma unroll
          for (int m = 0; m < NPR; m++) { pr[m] = (2 * m + 1 < NACC) ? (int)(((unsigned)acc[2 * m + 1][r] << 8) + (unsigned)acc[2 * m][r]) : acc[2 * m][r]; }
          unsigned lo = (unsigned)a.corr, hi = (unsigned)((uint64_t)a.corr >> 32);
          { const unsigned s0 = lo + (unsigned)pr[0]; hi += (unsigned)(pr[0] >> 31) + (s0 < lo); lo = s0; }
          if constexpr (NPR > 1) { const unsigned t1 = (unsigned)pr[1] << 16, s1 = lo + t1; hi += (unsigned)(pr[1] >> 16) + (s1 < lo); lo = s1; }
          if constexpr (NPR > 2) { hi += (unsigned)pr[2]; }
          if constexpr (NPR > 3) { hi += (unsigned)pr[3] << 16; }
          o[r] = (int64_t)(((uint64_t)hi << 32) | lo);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint64_t y = (uint64_t)a.corr;
#pragma unroll
          for (int w = 0; w < NACC; w++) {
            if (w < 8) { y += (uint64_t)(int64_t)acc[w][r] << (8 * w); }
          }
          o[r] = (int64_t)y;
        }
      }
      if constexpr (LZ) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          o[r] = (int64_t)((uint64_t)o[r] + (uint64_t)a.lz_k - (uint64_t)rT[r]) >> a.lz_s;
        }
      }
      // the conversion, stage by stage (uniform branches: a stage whose constants are trivial is skipped for all four outputs)
      if (a.e_stages & 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) { o[r] = (int64_t)(((uint64_t)o[r] << a.e_ls) << a.e_ka) >> a.e_ka; }
      }
      if (a.e_stages & 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) { o[r] = (int64_t)((uint64_t)((o[r] + a.e_rnd) >> a.e_rs) << a.e_ls2); }
      }
      if (a.e_stages & 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) { o[r] = o[r] < a.e_lo ? a.e_lo : (o[r] > a.e_hi ? a.e_hi : o[r]); }
      }
      if (a.e_stages & 8) {
#pragma unroll
        for (int r = 0; r < 4; r++) { o[r] = (int64_t)((uint64_t)o[r] << a.e_ko) >> a.e_ko; }
      }
      typedef long v2l __attribute__((ext_vector_type(2)));
      unsigned char *ob = ob0 + (FB ? k * TILE : 0);
      if (OEB == 8) {
        const int L = 8 * n_col + 2 * kg, sw = n_col & 7;
        *(v2l *)(ob + ((L ^ sw) * 16)) = (v2l){o[0], o[1]};
        *(v2l *)(ob + (((L + 1) ^ sw) * 16)) = (v2l){o[2], o[3]};
      } else if (OEB == 4) {
        const int L = 4 * n_col + kg;
        *(v4i *)(ob + ((L ^ ((n_col >> 1) & 3)) * 16)) = (v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
      } else {
        typedef short v4s_ __attribute__((ext_vector_type(4)));
        *(v4s_ *)(ob + ((4 * n_col + kg) ^ ((n_col >> 2) & 3)) * 8) = (v4s_){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
      }
    }
    if constexpr (FB) {
#pragma unroll
      for (int k = 0; k < SPW; k += (OEB == 2 ? 2 : 1)) { flush(s0 + k, k); }
    } else {
      flush(s0 + SPW - 1, 0);
    }
  };
  if (interior) { chunk(std::integral_constant<bool, true>()); } else { chunk(std::integral_constant<bool, false>()); }
}

template <typename TIN, int PX, int PCT, int NBT, int R, int OEB, int SPW, int PF, bool NT, bool FB, bool LZ = false>
static hipError_t launch_ring1(dim3 grid, hipStream_t s, const FirParams &p, const v4i *frag, const GenArgs &a) {
  hipLaunchKernelGGL((fir_gen_ring_kernel<TIN, PX, PCT, NBT, R, OEB, SPW, PF, NT, FB, LZ>), grid, dim3(64), 0, s, p, frag, a);
  return hipGetLastError();
}
// variant = spw (steps per chunk), pf (steps the loads run ahead), nt (non-temporal loads), fb (chunk-end store burst)
template <typename TIN, int PX, int PCT, int NBT, int R, int OEB>
static hipError_t launch_ring(int spw, int pf, int nt, int fb, dim3 grid, hipStream_t s, const FirParams &p, const v4i *frag, const GenArgs &a) {
#define ACDSP_RING_CASE(SPWV, PFV, NTV, FBV) \
  if (spw == SPWV && pf == PFV && nt == NTV && fb == FBV) { return launch_ring1<TIN, PX, PCT, NBT, R, OEB, SPWV, PFV, (NTV != 0), (FBV != 0)>(grid, s, p, frag, a); }
  ACDSP_RING_CASE(2, 2, 1, 0) ACDSP_RING_CASE(2, 2, 1, 1)                                 // the defaults (launch_fir_gen)
  ACDSP_RING_CASE(1, 1, 1, 0) ACDSP_RING_CASE(4, 1, 1, 0) ACDSP_RING_CASE(4, 2, 1, 1) ACDSP_RING_CASE(4, 4, 1, 1)   // A/B set
#undef ACDSP_RING_CASE
  return hipErrorInvalidValue;
}

template <typename TIN>
static hipError_t launch_px(int px, dim3 grid, size_t lds_bytes, hipStream_t s, const FirParams &p, const v4i *frag, const GenArgs &a) {
  const bool small = a.pl.nb <= 3 && a.pl.pc <= 2;
#define ACDSP_GEN_CASE(PXV)                                                                                         \
  case PXV: {                                                                                                       \
    if (small) {                                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void *)fir_gen_kernel<TIN, PXV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      if (e != hipSuccess) { return e; }                                                                            \
      hipLaunchKernelGGL((fir_gen_kernel<TIN, PXV, true>), grid, dim3(64), lds_bytes, s, p, frag, a);             \
    } else {                                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void *)fir_gen_kernel<TIN, PXV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      if (e != hipSuccess) { return e; }                                                                            \
      hipLaunchKernelGGL((fir_gen_kernel<TIN, PXV, false>), grid, dim3(64), lds_bytes, s, p, frag, a);            \
    }                                                                                                               \
    return hipGetLastError();                                                                                       \
  }
  // byte planes never exceed the container (launch_fir_gen checks px <= in_eb): only those instantiations exist
  if (px > (int)sizeof(TIN)) { return hipErrorInvalidValue; }
  if constexpr (sizeof(TIN) == 2) {
    switch (px) { ACDSP_GEN_CASE(1) ACDSP_GEN_CASE(2) default: return hipErrorInvalidValue; }
  } else if constexpr (sizeof(TIN) == 4) {
    switch (px) { ACDSP_GEN_CASE(1) ACDSP_GEN_CASE(2) ACDSP_GEN_CASE(3) ACDSP_GEN_CASE(4) default: return hipErrorInvalidValue; }
  } else {
    switch (px) {
      ACDSP_GEN_CASE(1) ACDSP_GEN_CASE(2) ACDSP_GEN_CASE(3) ACDSP_GEN_CASE(4)
      ACDSP_GEN_CASE(5) ACDSP_GEN_CASE(6) ACDSP_GEN_CASE(7) ACDSP_GEN_CASE(8)
      default: return hipErrorInvalidValue;
    }
  }
#undef ACDSP_GEN_CASE
}

// Branch-free output conversion of the fast kernels (GenArgs::e_*): false if the formats need the general requant64.
static bool gen_conv_params(const FirParams &p, int out_mode, int w_int, GenArgs *a, int src_bits = 0) {
  bool conv_ok = p.out.S && (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) && (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) &&
                 p.out.W >= 2 && p.out.W <= 64;
  int f_src = 0;
  if (out_mode == 1) { a->e_ls = 0; a->e_ka = 64 - w_int; f_src = p.in.F; conv_ok = conv_ok && w_int >= 2 && w_int <= 64; }
  else {
    a->e_ls = p.lossless_shift; a->e_ka = 64 - p.acc.W; f_src = p.acc.F;
    conv_ok = conv_ok && p.acc.S && p.acc.W >= 2 && p.acc.W <= 64 && p.lossless_shift >= 0 && p.lossless_shift < 64;
  }
  // src_bits > 0: the host has bounded the accumulator VALUE to that many bits (sign included), whatever its type's width
  const int rs = f_src - p.out.F, src_w0 = out_mode == 1 ? w_int : p.acc.W, src_w = (src_bits > 0 && src_bits < src_w0) ? src_bits : src_w0;
  a->e_rs = rs > 0 ? rs : 0; a->e_ls2 = rs < 0 ? -rs : 0;
  a->e_rnd = (p.out.Q == ACDSP_RND && rs > 0 && rs < 63) ? (int64_t(1) << (rs - 1)) : 0;
  // neither the rounding add nor the left shift can leave int64 (a 63- / 64-bit source with neither of them -- AC_TRN or rs <= 0, no left shift:
  // the reference testbench's ACC = OUT <64,32> -- is safe as it is)
  conv_ok = conv_ok && a->e_rs <= 62 && (src_w + a->e_ls2 <= 62 || (a->e_rnd == 0 && a->e_ls2 == 0));
  if (p.out.O == ACDSP_SAT) {
    a->e_hi = (int64_t)((uint64_t(1) << (p.out.W - 1)) - 1); a->e_lo = -a->e_hi - 1; a->e_ko = 0;
  } else {
    a->e_hi = INT64_MAX; a->e_lo = INT64_MIN; a->e_ko = 64 - p.out.W;
  }
  return conv_ok;
}

static int64_t gen_rebias_corr(int px, int64_t sum_h) {   // 128 * sum(h) * sum_{p < px-1} 256^p  (mod 2^64)
  unsigned __int128 bias = 0;
  for (int q = 0; q < px - 1; q++) { bias += (unsigned __int128)1 << (8 * q); }
  return (int64_t)(unsigned long long)((unsigned __int128)128 * bias * (unsigned long long)sum_h);
}

// Slot tables of the LZ residue loops (see fir_gen_ring_kernel).  With P0 the ring sample of (output r = 0, tap 0) of a lane -- parity
// b = off & 1 for every lane -- iteration m of a loop over taps T0 + i' reads the dwords holding samples P0 + woff - 2 m .. + 5:
//   b' = (b - T0) & 1;  woff = -T0 + b' - 2;  b' = 0: slot D = tap 2m, slot S = tap 2m + 1;  b' = 1: slot S = tap 2m, slot D = tap 2m - 1.
// The mirror sample of tap i is P0 - (N-1-i): a second window that slides UP from voff = -N + 1 - b (N odd: the mirror of a direct
// slot is direct) or -N - b (N even: it is the shifted one and vice versa).
bool fir_gen_lossy_table(const FirGenPlan &pl, const int64_t *coeffs, int n_taps, int n_pair, int n_single, int single0, int neg, int s, bool rnd,
                         FirLossyPlan *out, std::vector<uint32_t> *tab) {
  const uint32_t m1 = (1u << s) - 1, hh = rnd ? (1u << (s - 1)) : 0u;
  auto cl2 = [&](int tap) -> uint32_t { const uint32_t v = (uint32_t)((uint64_t)coeffs[tap] & m1); return v | (v << 16); };
  tab->assign(kLossyTabWords, 0u);
  const int b = pl.off & 1;
  // one loop over `cnt` taps from T0 on: iterations and slot taps
  auto build = [&](int T0, int cnt, int word0, int *n_it, int *woff) {
    const int bp = (b - T0) & 1;
    *woff = -T0 + bp - 2;
    *n_it = cnt <= 0 ? 0 : (bp == 0 ? (cnt + 1) / 2 : cnt / 2 + 1);
    for (int m = 0; m < *n_it; m++) {
      const int iS = bp == 0 ? 2 * m + 1 : 2 * m, iD = bp == 0 ? 2 * m : 2 * m - 1;
      if (word0 + 2 * m + 1 >= kLossyTabWords) { return false; }
      (*tab)[word0 + 2 * m] = (iS >= 0 && iS < cnt) ? cl2(T0 + iS) : 0u;
      (*tab)[word0 + 2 * m + 1] = (iD >= 0 && iD < cnt) ? cl2(T0 + iD) : 0u;
    }
    return true;
  };
  int p_n = 0, p_woff = 0, s_n = 0, s_woff = 0;
  if (!build(0, n_pair, 0, &p_n, &p_woff) || !build(single0, n_single, 2 * p_n, &s_n, &s_woff)) { return false; }
  const int slots = 2 * (p_n + s_n);
  if (slots < 1 || s > 15) { return false; }
  out->flush = (int32_t)(65535u / (2u * m1));      // iterations (two slots each) a 16-bit field holds: >= 1 for s <= 15
  out->s = s; out->neg = neg; out->var_y = (n_taps & 1) ? 0 : 1;
  out->p_n = p_n; out->p_woff = p_woff; out->p_voff = (n_taps & 1) ? -n_taps + 1 - b : -n_taps - b;
  out->s_n = s_n; out->s_woff = s_woff;
  out->h2 = hh | (hh << 16); out->m2 = m1 | (m1 << 16); out->k = (int64_t)slots * hh;
  return true;
}

// class-B ring shapes (the R = 1 shapes 20 / 21 / 23 / 24 / 25 below with the residue loop compiled in)
static int lossy_ring_shape(int in_eb, int oeb, const FirGenPlan &pl) {
  if (pl.R != 1 || pl.pc > 3 || pl.nb > 3) { return 0; }
  if (in_eb == 2) { return oeb == 8 ? 23 : (oeb == 4 ? 24 : (oeb == 2 ? 25 : 0)); }
  if (in_eb == 4) { return oeb == 8 ? 20 : (oeb == 4 ? 21 : 0); }
  return 0;
}
bool fir_gen_lossy_shape_ok(const FirParams &p, const FirGenPlan &pl, int acc_bits) {
  GenArgs a;
  FirParams q = p;
  q.lossless_shift = 0;
  return lossy_ring_shape(p.in_eb, p.out_eb, pl) != 0 && gen_conv_params(q, 0, 0, &a, acc_bits);
}

// p.n = inputs of this call; outputs m with first + m*R < n.  hist must hold >= off + 16 samples.
hipError_t launch_fir_gen(const FirParams &p_in, const FirGenPlan &pl, const uint32_t *d_frag, int out_mode, int w_int,
                          int64_t first, int64_t n_out, hipStream_t s, const FirLossyPlan *lz, int64_t *covered) {
  FirParams p = p_in;
  if (covered) { *covered = 0; }
  if (lz) { p.lossless_shift = 0; }   // class B: the exact sum is shifted RIGHT by lz->s in front of the ACC_TYPE wrap
  // (out_mode 1) OUT_TYPE with INT_TYPE's fraction and AC_WRAP converts by bit-field wraps only
  const int out_simple = (out_mode == 1 && p.out.F == p.in.F && p.out.O == ACDSP_WRAP) ? ((p.out.S && p.out.W >= w_int) ? 2 : 1) : 0;
  if (n_out <= 0) { return hipSuccess; }
  GenArgs a;
  a.pl = pl;
  a.ring_delta = 0;
  const int in_bits = p.in.W + (p.in.S ? 0 : 1);
  a.px = (in_bits + 7) / 8;
  if (a.px > p.in_eb) { return hipErrorInvalidValue; }
  if (lz) {
    a.px = p.in_eb;                    // the containers are sign-extended: every byte of them is a plane (the ring shapes are compiled for that)
    a.lz_s = lz->s; a.lz_neg = lz->neg; a.lz_var_y = lz->var_y; a.lz_p_n = lz->p_n; a.lz_p_woff = lz->p_woff; a.lz_p_voff = lz->p_voff;
    a.lz_s_n = lz->s_n; a.lz_s_woff = lz->s_woff; a.lz_flush = lz->flush;
    a.lz_h2 = lz->h2; a.lz_m2 = lz->m2; a.lz_k = lz->k; a.lz_tab = lz->d_tab;
  } else {
    a.lz_s = 0; a.lz_neg = a.lz_var_y = a.lz_p_n = a.lz_p_woff = a.lz_p_voff = a.lz_s_n = a.lz_s_woff = a.lz_flush = 0; a.lz_h2 = a.lz_m2 = 0; a.lz_k = 0; a.lz_tab = nullptr;
  }
  a.n_slots = 15 * pl.R + 4 * pl.nb;
  a.out_mode = out_mode; a.w_int = w_int; a.out_simple = out_simple;
  a.corr = gen_rebias_corr(a.px, pl.sum_h);
  a.first = first; a.n_out = n_out;
  a.n_steps = (n_out + 255) / 256;
  // Chunk length: four steps per wave (16 - 32 KB of input on the BASELINE shapes).  Workgroups are dispatched in memory order
  // (chunk index fastest), so the bytes the resident waves touch form a compact window that advances through both streams; short
  // spans keep that window small (DRAM page locality: tools/copy_probe2.hip -- 16 - 32 KB spans copy at 5.8 - 6.1 TB/s, 128+ KB
  // spans at 5.2).  The former rule (>= 16384 waves, i.e. 0.5 - 4 MB spans on the BASELINE shapes) ran cic_dec 6 %, polydec 11 %
  // slower; 3 - 4 steps measured best on both, 1 - 2 and 8+ lose 2 - 4 % (profiles/r3_span_sweep.txt).
  int64_t spw = 4;
  ACDSP_TUNE_ENV(spw_env, "ACDSP_GEN_SPW");   // tuning knob: steps (of 256 outputs) per wave
  if (spw_env && atoi(spw_env) > 0) { spw = atoi(spw_env); }
  a.steps_per_wave = spw;
  a.n16 = (p.n + 15) / 16 * 16;
  a.out_vec_ok = ((uintptr_t)p.y % 16 == 0) && ((p.out_stride * p.out_eb) % 16 == 0);
  a.chunk0 = 0;
  const int spl = (a.n_slots + 63) / 64;

  // Fast kernel: table of compiled shapes (BASELINE configs 3, 5a, 5b and the poly_dec row); nbt = K blocks compiled in.
  int nbt = 0;
  const int in_eb = p.in_eb, oeb = p.out_eb, pc = pl.pc, nb = pl.nb;
  const int npc = (a.n_slots * in_eb + 63) / 64;   // 16-byte pieces per lane
  if (in_eb == 4 && a.px == 4 && pc <= 2 && nb <= 3 && spl == 3 && npc <= 9 && oeb == 8) { nbt = 3; }        // CIC R8 N5 on int32 -> int64
  else if (in_eb == 2 && a.px == 2 && pc <= 3 && nb <= 6 && spl == 5 && npc <= 9 && oeb == 8) { nbt = 6; }   // CIC R16 N5 on int16 -> int64
  else if (in_eb == 8 && a.px == 5 && pc <= 2 && nb <= 3 && spl == 1 && oeb == 4) { nbt = 3; }   // 127-tap FIR on 36-bit words -> int32
  else if (in_eb == 2 && a.px == 2 && pc <= 2 && nb <= 4 && spl == 3 && npc <= 5 && oeb == 2) { nbt = 4; }   // 128-tap decimate-by-8 on int16 -> int16
  // ... and a conversion the branch-free form covers: signed wrapping accumulator, signed OUT, TRN/RND, WRAP/SAT
  const bool conv_ok = gen_conv_params(p, out_mode, w_int, &a, lz ? lz->acc_bits : 0);
  a.e_stages = ((a.e_ls != 0 || a.e_ka != 0) ? 1 : 0) | ((a.e_rnd != 0 || a.e_rs != 0 || a.e_ls2 != 0) ? 2 : 0) |
               ((a.e_lo != INT64_MIN || a.e_hi != INT64_MAX) ? 4 : 0) | (a.e_ko != 0 ? 8 : 0);
  auto set_pairwise = [&]() {
    // plane accumulator w sums the products of (input plane pp, coefficient digit q), pp + q = w, |plane byte| <= 128
    int64_t bw[kGenMaxPX + kGenMaxPC] = {0};
    for (int w = 0; w < a.px + pl.pc - 1; w++) {
      for (int q = 0; q < pl.pc; q++) { if (w - q >= 0 && w - q < a.px) { bw[w] += 128 * pl.dig_abs[q]; } }
    }
    bool pw_ok = true;
    for (int w = 0; w + 1 < kGenMaxPX + kGenMaxPC; w += 2) { pw_ok = pw_ok && (bw[w + 1] * 256 + bw[w] < (int64_t(1) << 31)); }
    static const bool no_pw = getenv("ACDSP_GEN_NO_PW") != nullptr;   // A/B knob
    a.pw = (pw_ok && !no_pw) ? 1 : 0;
  };
  set_pairwise();

  // Ring variant (fir_gen_ring_kernel) for the decimating BASELINE shapes.  ACDSP_GEN_RING=0: the window-per-step kernel (A/B
  // reference); ACDSP_GEN_RING=spw,pf,nt,fb picks another compiled variant (steps per chunk, load distance, non-temporal loads,
  // chunk-end store burst).  Defaults from same-process sweeps on three boxes (profiles/r4_ring_sweep.txt): TWO-step chunks with
  // every load of the chunk issued up front (16 KB bursts on config 3, 8 KB on poly_dec) beat longer chunks by 3 - 5 % and the
  // window-per-step kernel by 7 - 11 %; the store burst pays for 2-byte outputs (one 1 KB store per chunk instead of two half-wave
  // ones) and costs occupancy for 8-byte outputs.
  ACDSP_TUNE_ENV(ring_env, "ACDSP_GEN_RING");
  int r_spw = 2, r_pf = 2, r_nt = 1, r_fb = (oeb == 2) ? 1 : 0, ring_shape = 0;
  // 17 .. 24-bit samples in int32 containers (three planes) take the four-plane ring shapes where one fits: the containers are sign-extended, the
  // fourth plane is the sign -- 8 MFMAs more per step against the ring's better stream (CIC R8 M2 N3 on <24,8>: 0.71 on the window kernel)
  const bool promote = !lz && in_eb == 4 && a.px == 3;
  const int px = promote ? 4 : a.px;
  bool ring_on = true;
  if (ring_env) {
    int v[4];
    if (sscanf(ring_env, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) { r_spw = v[0]; r_pf = v[1]; r_nt = v[2]; r_fb = v[3]; }
    else { ring_on = atoi(ring_env) != 0; }
  }
  if (lz) {
    if (!conv_ok || !a.out_vec_ok || !(ring_shape = lossy_ring_shape(in_eb, oeb, pl))) { return hipSuccess; }   // nothing covered: the caller's exact-order kernel takes the call
    r_spw = in_eb == 2 ? 16 : 8;
  } else if (ring_on && conv_ok && a.out_vec_ok) {
    if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 3 && oeb == 8 && pl.R == 8) { ring_shape = 1; }          // CIC R8 N5 on int32 -> int64
    else if (in_eb == 2 && px == 2 && pc <= 3 && nb <= 6 && oeb == 8 && pl.R == 16) { ring_shape = 2; }   // CIC R16 N5 on int16 -> int64
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 4 && oeb == 2 && pl.R == 8) { ring_shape = 3; }    // 128-tap decimate-by-8 on int16 -> int16
    // further CIC decimator shapes (tools/cic_sweep.py): no BASELINE config, same kernel
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 3 && oeb == 4 && pl.R == 8) { ring_shape = 4; }    // CIC R8 on int16 -> int32 (INT_TYPE <= 32 bits)
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 2 && oeb == 8 && pl.R == 4) { ring_shape = 5; if (!ring_env) { r_spw = 4; r_pf = 4; } }   // CIC R4 on int32: 4 KB per step
    // ac_poly_dec on int16 at other factors / output widths (tools/poly_shapes.py): about 8 KB of input per wave, every load up front
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 2 && (oeb == 2 || oeb == 8) && pl.R == 2 && !ring_env) { ring_shape = oeb == 2 ? 10 : 11; r_spw = 8; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 3 && (oeb == 2 || oeb == 8) && pl.R == 4 && !ring_env) { ring_shape = oeb == 2 ? 12 : 13; r_spw = 4; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 4 && oeb == 8 && pl.R == 8 && !ring_env) { ring_shape = 14; r_spw = 2; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 8 && (oeb == 2 || oeb == 8) && pl.R == 16 && !ring_env) { ring_shape = oeb == 2 ? 15 : 16; r_spw = 2; }
    // plain (R = 1) FIR on 16-bit samples whose coefficients need more than the int8 kernel's 16 bits (two load groups of two steps per ... )
    else if (in_eb == 2 && px == 2 && pc <= 3 && nb <= 3 && (oeb == 8 || oeb == 4 || oeb == 2) && pl.R == 1 && !ring_env) { ring_shape = oeb == 8 ? 23 : (oeb == 4 ? 24 : 25); r_spw = 16; }
    // plain (R = 1) FIR on 32-bit samples, up to three coefficient digits and three K-blocks (~130 taps): one 1 KB load per 256-output step
    else if (in_eb == 4 && px == 4 && pc <= 3 && nb <= 3 && (oeb == 8 || oeb == 4) && pl.R == 1 && !ring_env) { ring_shape = oeb == 8 ? 20 : 21; r_spw = 8; }
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 4 && oeb == 8 && pl.R == 7 && !ring_env) { ring_shape = 7; r_spw = 2; }   // CIC R7 (M2 N4: the reference testbench) on int32: 7 KB per step
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 3 && (oeb == 4 || oeb == 8) && pl.R == 5 && !ring_env) { ring_shape = oeb == 4 ? 8 : 9; r_spw = 4; }
    // decimal rates: R = 10 on int32 (10 KB per step, two steps per wave) and on int16 (5 KB per step, four steps)
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 4 && oeb == 8 && pl.R == 10 && !ring_env) { ring_shape = 30; r_spw = 2; }
    else if (in_eb == 2 && px == 2 && pc <= 3 && nb <= 4 && (oeb == 4 || oeb == 8) && pl.R == 10 && !ring_env) { ring_shape = oeb == 4 ? 31 : 32; r_spw = 4; }
    // further small and decimal rates (multistage decimators: 3, 6, 12, 20), same kernel, about 6 - 12 KB of input per wave
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 3 && (oeb == 4 || oeb == 8) && pl.R == 3 && !ring_env) { ring_shape = oeb == 4 ? 33 : 34; r_spw = 8; }
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 3 && oeb == 8 && pl.R == 3 && !ring_env) { ring_shape = 35; r_spw = 4; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 3 && (oeb == 4 || oeb == 8) && pl.R == 6 && !ring_env) { ring_shape = oeb == 4 ? 36 : 37; r_spw = 4; }
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 3 && oeb == 8 && pl.R == 6 && !ring_env) { ring_shape = 38; r_spw = 2; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 5 && (oeb == 4 || oeb == 8) && pl.R == 12 && !ring_env) { ring_shape = oeb == 4 ? 39 : 40; r_spw = 2; }
    else if (in_eb == 2 && px == 2 && pc <= 3 && nb <= 8 && (oeb == 4 || oeb == 8) && pl.R == 20 && !ring_env) { ring_shape = oeb == 4 ? 41 : 42; r_spw = 2; }
    else if (in_eb == 4 && px == 4 && pc <= 2 && nb <= 3 && oeb == 8 && pl.R == 5 && !ring_env) { ring_shape = 43; r_spw = 2; }
    else if (in_eb == 2 && px == 2 && pc <= 2 && nb <= 4 && (oeb == 4 || oeb == 8) && pl.R == 7 && !ring_env) { ring_shape = oeb == 4 ? 44 : 45; r_spw = 4; }   // CIC R5 on int16: 2.5 KB per step, one load group per two steps
    else if (in_eb == 4 && px == 4 && pc <= 3 && nb <= 6 && oeb == 8 && pl.R == 16) { ring_shape = 6; if (!ring_env) { r_spw = 1; r_pf = 1; } }   // CIC R16 N5 on int32: 16 KB per step = one step per wave (LDS: 18 KB of planes per step)
  }
  if (promote && ring_shape) { a.px = 4; a.corr = gen_rebias_corr(a.px, pl.sum_h); set_pairwise(); }
  static const bool trace = getenv("ACDSP_GEN_TRACE") != nullptr;   // diagnostic: which compiled shape a launch resolves to
  if (trace) {
    fprintf(stderr, "[acdsp] fir_gen: in_eb %d px %d pc %d nb %d oeb %d R %d conv_ok %d out_vec_ok %d -> ring shape %d, window shape nbt %d\n", in_eb, a.px, pc, nb, oeb,
            pl.R, (int)conv_ok, (int)a.out_vec_ok, ring_shape, nbt);
  }
  if (ring_shape) {
    spw = r_spw; a.steps_per_wave = spw;
    const int ls = 8 / in_eb;
    const int64_t w0slot = (first - pl.off) / 16;                      // exact: the plan aligns the window start to a slot
    a.ring_delta = (int32_t)(((w0slot % ls) + ls) % ls);
  }
  const int phys_nb = nbt > pl.nb ? nbt : pl.nb;                     // zero-fragment blocks still read their (stale) slots
  const int slots_alloc = 15 * pl.R + 4 * phys_nb;
  const int phys = gen_slot_map(a, pl.R, slots_alloc);
  a.obuf_off = a.px * (phys + 1) * 16;
  ACDSP_TUNE_ENV(lds_pad_env, "ACDSP_GEN_LDS_PAD");   // diagnostic: extra LDS bytes per wave (lowers the occupancy)
  const size_t lds_bytes = (size_t)a.obuf_off + 2048 + (lds_pad_env ? (size_t)atoi(lds_pad_env) : 0);
  const int64_t n_chunks = (a.n_steps + spw - 1) / spw;
  const int64_t fast_chunks = ((nbt || ring_shape) && conv_ok && a.out_vec_ok) ? n_out / (spw * 256) : 0;   // chunks made of complete steps only
  const v4i *fr = (const v4i *)d_frag;
  hipError_t e = hipSuccess;
  a.xcd_map = 0;
  if (fast_chunks > 0) {
    dim3 grid((unsigned)fast_chunks, (unsigned)p.n_ch);
    // XCD-affine chunk order (xcd_remap): measured per shape, same box, three passes (profiles/r3_xcd_map.txt) -- config 3 (CIC R8 on
    // int32, 8 KB of input per step) +2.5 % every time, poly_dec -0.6 % twice and +6.6 % once, the fused DDC -3 %: on for the
    // int32 decimator shape only.  ACDSP_XCD_MAP=0 / 1 forces it off / on for every shape (A/B knob).
    a.xcd_map = (xcd_map_wanted(out_mode == 1 && in_eb == 4) && ((int64_t)grid.x * grid.y) % 8 == 0) ? 1 : 0;
    if (lz && ring_shape == 23) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 8, 16, 8, true, false, true>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 8, 16, 8, true, false, true>(grid, s, p, fr, a); }
    else if (lz && ring_shape == 24) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 4, 16, 8, true, false, true>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 4, 16, 8, true, false, true>(grid, s, p, fr, a); }
    else if (lz && ring_shape == 25) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 2, 16, 8, true, true, true>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 2, 16, 8, true, true, true>(grid, s, p, fr, a); }
    else if (lz && ring_shape == 20) { e = nb == 1 ? launch_ring1<int32_t, 4, 3, 1, 1, 8, 8, 8, true, false, true>(grid, s, p, fr, a) : launch_ring1<int32_t, 4, 3, 3, 1, 8, 8, 8, true, false, true>(grid, s, p, fr, a); }
    else if (lz && ring_shape == 21) { e = nb == 1 ? launch_ring1<int32_t, 4, 3, 1, 1, 4, 8, 8, true, false, true>(grid, s, p, fr, a) : launch_ring1<int32_t, 4, 3, 3, 1, 4, 8, 8, true, false, true>(grid, s, p, fr, a); }
    else if (lz) { e = hipErrorInvalidValue; }
    else if (ring_shape == 1) { e = launch_ring<int32_t, 4, 2, 3, 8, 8>(r_spw, r_pf, r_nt, r_fb, grid, s, p, fr, a); }
    else if (ring_shape == 2) { e = launch_ring<int16_t, 2, 3, 6, 16, 8>(r_spw, r_pf, r_nt, r_fb, grid, s, p, fr, a); }
    else if (ring_shape == 3) { e = launch_ring<int16_t, 2, 2, 4, 8, 2>(r_spw, r_pf, r_nt, r_fb, grid, s, p, fr, a); }
    else if (ring_shape == 4) { e = launch_ring<int16_t, 2, 2, 3, 8, 4>(r_spw, r_pf, r_nt, r_fb, grid, s, p, fr, a); }
    else if (ring_shape == 5) { e = launch_ring1<int32_t, 4, 2, 2, 4, 8, 4, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 6) { e = launch_ring1<int32_t, 4, 3, 6, 16, 8, 1, 1, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 7) { e = launch_ring1<int32_t, 4, 2, 4, 7, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 8) { e = launch_ring1<int16_t, 2, 2, 3, 5, 4, 4, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 30) { e = launch_ring1<int32_t, 4, 2, 4, 10, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 31) { e = launch_ring1<int16_t, 2, 3, 4, 10, 4, 4, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 32) { e = launch_ring1<int16_t, 2, 3, 4, 10, 8, 4, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 33) { e = launch_ring1<int16_t, 2, 2, 3, 3, 4, 8, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 34) { e = launch_ring1<int16_t, 2, 2, 3, 3, 8, 8, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 35) { e = launch_ring1<int32_t, 4, 2, 3, 3, 8, 4, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 36) { e = launch_ring1<int16_t, 2, 2, 3, 6, 4, 4, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 37) { e = launch_ring1<int16_t, 2, 2, 3, 6, 8, 4, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 38) { e = launch_ring1<int32_t, 4, 2, 3, 6, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 39) { e = launch_ring1<int16_t, 2, 2, 5, 12, 4, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 40) { e = launch_ring1<int16_t, 2, 2, 5, 12, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 41) { e = launch_ring1<int16_t, 2, 3, 8, 20, 4, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 42) { e = launch_ring1<int16_t, 2, 3, 8, 20, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 43) { e = launch_ring1<int32_t, 4, 2, 3, 5, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 44) { e = launch_ring1<int16_t, 2, 2, 4, 7, 4, 4, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 45) { e = launch_ring1<int16_t, 2, 2, 4, 7, 8, 4, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 9) { e = launch_ring1<int16_t, 2, 2, 3, 5, 8, 4, 2, true, false>(grid, s, p, fr, a); }
    // (plans of ONE K-block -- up to ~49 taps: the reference testbenches' 27 / 29 -- have their own instantiations: the three-block shapes
    // issue 3 x the MFMAs on zero fragments, 36 instead of 12 per 256 outputs on 32-bit samples -- the matrix pipe, not HBM, bounded them)
    else if (ring_shape == 23) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 8, 16, 8, true, false>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 8, 16, 8, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 24) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 4, 16, 8, true, false>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 4, 16, 8, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 25) { e = nb == 1 ? launch_ring1<int16_t, 2, 3, 1, 1, 2, 16, 8, true, true>(grid, s, p, fr, a) : launch_ring1<int16_t, 2, 3, 3, 1, 2, 16, 8, true, true>(grid, s, p, fr, a); }
    else if (ring_shape == 20) { e = nb == 1 ? launch_ring1<int32_t, 4, 3, 1, 1, 8, 8, 8, true, false>(grid, s, p, fr, a) : launch_ring1<int32_t, 4, 3, 3, 1, 8, 8, 8, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 21) { e = nb == 1 ? launch_ring1<int32_t, 4, 3, 1, 1, 4, 8, 8, true, false>(grid, s, p, fr, a) : launch_ring1<int32_t, 4, 3, 3, 1, 4, 8, 8, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 10) { e = launch_ring1<int16_t, 2, 2, 2, 2, 2, 8, 8, true, true>(grid, s, p, fr, a); }
    else if (ring_shape == 11) { e = launch_ring1<int16_t, 2, 2, 2, 2, 8, 8, 8, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 12) { e = launch_ring1<int16_t, 2, 2, 3, 4, 2, 4, 4, true, true>(grid, s, p, fr, a); }
    else if (ring_shape == 13) { e = launch_ring1<int16_t, 2, 2, 3, 4, 8, 4, 4, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 14) { e = launch_ring1<int16_t, 2, 2, 4, 8, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (ring_shape == 15) { e = launch_ring1<int16_t, 2, 2, 8, 16, 2, 2, 2, true, true>(grid, s, p, fr, a); }
    else if (ring_shape == 16) { e = launch_ring1<int16_t, 2, 2, 8, 16, 8, 2, 2, true, false>(grid, s, p, fr, a); }
    else if (in_eb == 4) { e = launch_fast<int32_t, 4, 2, 3, 9, 8>(grid, lds_bytes, s, p, fr, a); }
    else if (in_eb == 8) { e = launch_fast<int64_t, 5, 2, 3, 1, 4>(grid, lds_bytes, s, p, fr, a); }
    else if (oeb == 8) { e = launch_fast<int16_t, 2, 3, 6, 9, 8>(grid, lds_bytes, s, p, fr, a); }
    else { e = launch_fast<int16_t, 2, 2, 4, 5, 2>(grid, lds_bytes, s, p, fr, a); }
    if (e != hipSuccess) { return e; }
  }
  if (lz) {
    if (covered) { *covered = fast_chunks * spw * 256; }
    return hipSuccess;
  }
  if (fast_chunks >= n_chunks) { return hipSuccess; }
  a.xcd_map = 0;
  a.chunk0 = (int32_t)fast_chunks;                                    // ragged tail (and every unlisted shape): general kernel
  dim3 grid((unsigned)(n_chunks - fast_chunks), (unsigned)p.n_ch);
  switch (p.in_eb) {
    case 2: return launch_px<int16_t>(a.px, grid, lds_bytes, s, p, fr, a);
    case 4: return launch_px<int32_t>(a.px, grid, lds_bytes, s, p, fr, a);
    default: return launch_px<int64_t>(a.px, grid, lds_bytes, s, p, fr, a);
  }
}

// =============================================================================================
// Fused decimator -> FIR cascade (BASELINE config 5): stage A = a decimating FIR on int16 input (the
// CIC decimator through its FIR identity, outputs = the lossless INT_TYPE words), stage B = a FIR on those words.
// One wave walks a chunk of 256-output steps; the 256 stage-A outputs of a step never leave the CU: their byte planes
// go straight into a 32-slot LDS ring that is stage B's operand (B needs outputs m0-128 .. m0+255 of A: the ring keeps
// two steps).  A chunk starts one step early (stage A only) to fill the ring -- the stage-B state is recomputed from the
// input history instead of being carried, so the handle keeps 256*R + window input samples per channel.
// Against the two-kernel path this saves writing and re-reading the 8-byte intermediate (config 5: 14.0 -> 9.7 GB).
//   GUARD: chunks with incomplete steps (ragged end of the call): element-wise guarded stores.
#ifndef ACDSP_CASC_BARRIER
#define ACDSP_CASC_BARRIER 1
#endif
//   PART:  0 = every chunk of the grid; 1 = interior chunks only (every window inside the call's samples: uniform base + immediate
//          addressing), 2 = the edge chunks only (a row's first chunk, which reaches into the history, and its last complete one when
//          that window passes the end of the row) on a (2, n_ch) grid.  One kernel holding both forms kept the edge form's per-piece
//          offsets alive across the interior loop: 2 VGPRs of scratch at 256 registers.
//   (Store bursts -- the tiles of 2 / 4 / 8 steps leaving as one contiguous burst -- were built and measured in round 4: no gain,
//   profiles/r4_cascade_ablation.txt; removed.)
template <int PXA, int PCA, int NBA, int SPLA, int PXB, int PCB, int NBB, bool GUARD, bool LIMB, int PART>
__global__ void __launch_bounds__(64, 2) cascade_kernel(FirParams pa, FirParams pb, const v4i *__restrict__ fragA,
                                                       const v4i *__restrict__ fragB, GenArgs a, GenArgs b) {
  typedef int16_t TIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [A planes][B ring: PXB x 32 slots][1 KB output tile]
  const int lane = threadIdx.x;
  const int n_col = lane & 15, kg = lane >> 4;
  int bx, ch;
  xcd_remap(a.xcd_map, bx, ch);
  if constexpr (PART == 2) { bx = blockIdx.x == 0 ? 0 : a.edge_chunk; ch = blockIdx.y; }
  const int R = a.pl.R;
  const int plane_bytes = a.obuf_off / PXA;
  unsigned char *ring = lds + a.obuf_off;
  unsigned char *ob = ring + PXB * 512;

  v4i AA[NBA][PCA], AB[NBB][PCB];
#pragma unroll
  for (int bb = 0; bb < NBA; bb++) {
#pragma unroll
    for (int q = 0; q < PCA; q++) { AA[bb][q] = (bb < a.pl.nb && q < a.pl.pc) ? fragA[((size_t)q * a.pl.nb + bb) * 64 + lane] : (v4i){0, 0, 0, 0}; }
  }
#pragma unroll
  for (int bb = 0; bb < NBB; bb++) {
#pragma unroll
    for (int q = 0; q < PCB; q++) { AB[bb][q] = (bb < b.pl.nb && q < b.pl.pc) ? fragB[((size_t)q * b.pl.nb + bb) * 64 + lane] : (v4i){0, 0, 0, 0}; }
  }
  const TIN *xrow = (const TIN *)pa.x + (int64_t)ch * pa.in_stride;
  const TIN *hrow = (const TIN *)pa.hist + (int64_t)ch * pa.hl + pa.hl;
  const int64_t s0 = ((int64_t)bx + a.chunk0) * a.steps_per_wave;
  const int64_t s1 = GUARD ? ((s0 + a.steps_per_wave < a.n_steps) ? s0 + a.steps_per_wave : a.n_steps) : s0 + a.steps_per_wave;

  // window loads coalesced over 16-byte pieces (8 samples; a slot is two pieces), as in fir_gen_fast_kernel
  constexpr int NPCA = SPLA;   // 16-byte pieces per lane and step (ceil(n_slots * 2 / 64): 9 for R = 16, six K blocks)
  v4i pre[NPCA];
  int pc_off[NPCA], pc_ps[NPCA];
#pragma unroll
  for (int k = 0; k < NPCA; k++) {
    int slot = (lane + 64 * k) / 2;
    const int sub = lane & 1;
    if (slot >= a.n_slots) { slot = a.n_slots - 1; }
    pc_off[k] = 16 * slot + 8 * sub;
    pc_ps[k] = phys_slot(slot, a) * 16 + 8 * sub;
  }
  // Interior chunks (every window of the chunk lies inside the call's samples: all but a row's first and last chunk) address
  // their pieces as uniform base + 16 bytes x lane + 1024 bytes x k -- no per-piece 64-bit selects; only the last two pieces,
  // whose surplus lanes repeat the last slot, keep an offset register.  (A per-step uniform branch instead of two loops
  // was measured: no gain over the general form -- the split loop body schedules worse.)
  const int64_t W_first = a.first + (s0 - 1) * 256 * R - a.pl.off, W_last = a.first + (s1 - 1) * 256 * R - a.pl.off;
  const bool interior = !GUARD && W_first >= 0 && W_last + 16 * a.n_slots <= a.n16;
  if constexpr (PART == 1) { if (!interior) { return; } }
  if constexpr (PART == 2) { if (interior) { return; } }
  const unsigned lane16 = 16u * (unsigned)lane;
  auto fetch = [&](int64_t st, auto fast_c) {
    const int64_t W0 = a.first + st * 256 * R - a.pl.off;
    if constexpr (decltype(fast_c)::value) {
      const char *base = (const char *)(xrow + W0);
#pragma unroll
      for (int k = 0; k < NPCA; k++) {
        const unsigned off = k < NPCA - 2 ? lane16 + 1024u * k : 2u * (unsigned)pc_off[k];
#ifdef ACDSP_CASC_ABL_LOAD   // timing-only ablation: one load per step instead of nine
        if (k > 0) { pre[k] = pre[0]; continue; }
#endif
        pre[k] = *(const v4i *)(base + off);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NPCA; k++) {
        const int64_t t = W0 + pc_off[k];
        const TIN *src = (t < 0) ? hrow + t : xrow + ((t < a.n16) ? t : 0);
        pre[k] = *(const v4i *)src;
      }
    }
  };
  auto stage_piece = [&](const v4i &v, int ps) {
#pragma unroll
    for (int pp = 0; pp < PXA; pp++) {
      const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
      unsigned lo = __builtin_amdgcn_perm((unsigned)v.y, (unsigned)v.x, sel), hi = __builtin_amdgcn_perm((unsigned)v.w, (unsigned)v.z, sel);
      if (pp < PXA - 1) { lo ^= 0x80808080u; hi ^= 0x80808080u; }
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      *(v2u *)(lds + pp * plane_bytes + ps) = (v2u){lo, hi};
    }
  };
  int xs_of[NBA];
#pragma unroll
  for (int bb = 0; bb < NBA; bb++) { xs_of[bb] = phys_slot(R * n_col + 4 * bb + kg, a) * 16; }
  char *yrow = (char *)pb.y + (int64_t)ch * pb.out_stride * 4;

  auto flush = [&](int64_t st) {   // 256 int32 outputs of a finished step: one 1 KB store
    const v4i val = *(const v4i *)(ob + ((lane ^ ((lane >> 3) & 3)) * 16));
#ifdef ACDSP_CASC_ABL_STORE  // timing-only ablation: no output stores (except a never-true one that keeps the values alive)
    if (a.n_out >= 0) { return; }
#endif
    ACDSP_GEN_ST(val, (v4i *)(yrow + st * 1024 + 16 * lane));
  };
  // One step.  WARM: stage A only (fills the ring for the chunk's first stage-B step).  FLUSH: step st-1 waits in the tile.
  auto body = [&](int64_t st, auto warm_c, auto flush_c, auto fast_c) {
    constexpr bool WARM = decltype(warm_c)::value, FLUSH = decltype(flush_c)::value;
#pragma unroll
    for (int k = 0; k < NPCA; k++) {
      if constexpr (LIMB) {
        // LDS offsets re-derived per piece rather than ten resident registers (the loop must not spill); the slot map is an add, so
        // the compiler folds the regular pieces into one base register + immediate offsets.
        int slot = (lane >> 1) + 32 * k;
        if (k >= NPCA - 2 && slot >= a.n_slots) { slot = a.n_slots - 1; }
        stage_piece(pre[k], phys_slot(slot, a) * 16 + 8 * (lane & 1));
      } else {
        stage_piece(pre[k], pc_ps[k]);
      }
    }
#ifdef ACDSP_CASC_FLUSH_FIRST   // A/B: the round-2/3 order (store, then the next step's loads)
    if (FLUSH && !GUARD) { flush(st - 1); }
    fetch(st + 1 < s1 ? st + 1 : st, fast_c);
#else
    // The next step's loads first, THEN the store of the previous step's tile: VMEM operations of a wave retire in order (one vmcnt
    // for loads and stores on gfx9), so the wait for these loads at the top of the next step no longer includes the store's write
    // acknowledge -- with the stores compiled out the kernel runs 1.87 -> 1.60 ms although they are 1/9 of its traffic
    // (profiles/r4_cascade_ablation.txt); the store now has two steps to retire.
    fetch(st + 1 < s1 ? st + 1 : st, fast_c);
    if (FLUSH && !GUARD) { flush(st - 1); }
#endif
    // The next step's loads must leave before this step's arithmetic: left alone, the scheduler sinks them to the end of the
    // step and the wave eats the whole HBM latency at the top of the next one.  A compiler-level memory barrier pins them
    // between the staging writes above and the fragment reads below; ALU work stays free to move.
    if (ACDSP_CASC_BARRIER) { asm volatile("" ::: "memory"); }

    // ---- stage A: 256 outputs of the decimating FIR, wrapped to the INT_TYPE width ----
    v4i accA[PXA + PCA - 1];
#pragma unroll
    for (int w = 0; w < PXA + PCA - 1; w++) { accA[w] = LIMB ? (v4i){a.dig[w], a.dig[w], a.dig[w], a.dig[w]} : (v4i){0, 0, 0, 0}; }
#pragma unroll
    for (int bb = 0; bb < NBA; bb++) {
      v4i X[PXA];
#pragma unroll
      for (int pp = 0; pp < PXA; pp++) { X[pp] = *(const v4i *)(lds + pp * plane_bytes + xs_of[bb]); }
#pragma unroll
      for (int q = 0; q < PCA; q++) {
#pragma unroll
        for (int pp = 0; pp < PXA; pp++) {
#ifdef ACDSP_CASC_ABL_A      // timing-only ablation: one K-block of stage A instead of six
          if (bb > 0) { continue; }
#endif
          accA[pp + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AA[bb][q], X[pp], accA[pp + q], 0, 0, 0);
        }
      }
    }
    // byte planes of the four words of this lane -> ring slot (st * 16 + n_col) & 31, bytes 4 kg .. 4 kg + 3
    const int wslot = (((int)(st & 1) * 16 + n_col) & 31) * 16 + 4 * kg;
    if constexpr (LIMB) {
      // y = sum_w acc_w 2^(8w) is the exact word (constant folded into the accumulators): a byte carry chain in 32-bit
      // registers gives its bytes -- u_w = acc_w + (u_{w-1} >> 8), byte w = u_w & 255, the last u holds y >> 8 (NW - 1)
      static_assert(PXA + PCA - 1 == 4 && PXB == 5, "limb epilogue: four weight classes into five byte planes");
      int u[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        u[0][r] = accA[0][r];
#pragma unroll
        for (int w = 1; w < 4; w++) { u[w][r] = accA[w][r] + (u[w - 1][r] >> 8); }
      }
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        const unsigned d = gather4((unsigned)u[pp][0], (unsigned)u[pp][1], (unsigned)u[pp][2], (unsigned)u[pp][3], 0) ^ 0x80808080u;
        *(unsigned *)(ring + pp * 512 + wslot) = d;
      }
      // top plane: bits 32 .. w_int - 1 of the word, sign-extended (the wrap to INT_TYPE)
      unsigned tb[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { tb[r] = (unsigned)__builtin_amdgcn_sbfe(u[3][r], 8, a.w_int - 32); }
      *(unsigned *)(ring + 4 * 512 + wslot) = gather4(tb[0], tb[1], tb[2], tb[3], 0);
    } else {
      unsigned lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint64_t y = (uint64_t)a.corr;
#pragma unroll
        for (int w = 0; w < PXA + PCA - 1; w++) { y += (uint64_t)(int64_t)accA[w][r] << (8 * w); }
        const int64_t v = (int64_t)(y << a.e_ka) >> a.e_ka;        // INT_TYPE word (signed, w_int bits)
        lo[r] = (unsigned)v; hi[r] = (unsigned)((uint64_t)v >> 32);
      }
#pragma unroll
      for (int pp = 0; pp < PXB; pp++) {
        unsigned d = pp < 4 ? gather4(lo[0], lo[1], lo[2], lo[3], pp) : gather4(hi[0], hi[1], hi[2], hi[3], pp - 4);
        if (pp < PXB - 1) { d ^= 0x80808080u; }
        *(unsigned *)(ring + pp * 512 + wslot) = d;
      }
    }
    if (WARM) { return; }

    // ---- stage B: 256 outputs of the FIR on ring slots st*16 - 8 .. st*16 + 18 ----
    v4i accB[PXB + PCB - 1];
#pragma unroll
    for (int w = 0; w < PXB + PCB - 1; w++) { accB[w] = LIMB ? (v4i){b.dig[w], b.dig[w], b.dig[w], b.dig[w]} : (v4i){0, 0, 0, 0}; }
    const int rbase = (int)(st & 1) * 16 - 8 + n_col + kg;
#pragma unroll
    for (int bb = 0; bb < NBB; bb++) {
      v4i X[PXB];
#pragma unroll
      for (int pp = 0; pp < PXB; pp++) { X[pp] = *(const v4i *)(ring + pp * 512 + (((rbase + 4 * bb) & 31) * 16)); }
#pragma unroll
      for (int q = 0; q < PCB; q++) {
#pragma unroll
        for (int pp = 0; pp < PXB; pp++) {
#ifdef ACDSP_CASC_ABL_B      // timing-only ablation: one K-block of stage B instead of three
          if (bb > 0) { continue; }
#endif
          accB[pp + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AB[bb][q], X[pp], accB[pp + q], 0, 0, 0);
        }
      }
    }
    int o[4];
    if constexpr (LIMB) {
      // z = sum_w acc_w 2^(8w) (exact; correction and rounding constant folded in) as two 32-bit limbs through 16-bit
      // carries; OUT = clamp(z >> rs): the high limb is clamped first so that the funnel shift cannot leave int32
      static_assert(PXB + PCB - 1 == 6, "limb epilogue: six weight classes");
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int p01 = accB[0][r] + (int)((unsigned)accB[1][r] << 8);
        const int p23 = accB[2][r] + (int)((unsigned)accB[3][r] << 8);
        const int p45 = accB[4][r] + (int)((unsigned)accB[5][r] << 8);
        const int v1 = p23 + (p01 >> 16);
        int H = p45 + (v1 >> 16);                                                      // z >> 32
        const unsigned L = __builtin_amdgcn_perm((unsigned)v1, (unsigned)p01, 0x05040100u);   // z mod 2^32
        // AC_SAT: both clamps act; AC_WRAP: the host sets them to the int32 range and the bit-field extract wraps
        H = H < b.l_hb_lo ? b.l_hb_lo : (H > b.l_hb_hi ? b.l_hb_hi : H);
        int q = (int)__builtin_amdgcn_alignbit((unsigned)H, L, (unsigned)b.e_rs);
        q = q < b.l_lo ? b.l_lo : (q > b.l_hi ? b.l_hi : q);
        o[r] = __builtin_amdgcn_sbfe(q, 0, b.l_w);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        uint64_t y = (uint64_t)b.corr;
#pragma unroll
        for (int w = 0; w < PXB + PCB - 1; w++) { y += (uint64_t)(int64_t)accB[w][r] << (8 * w); }
        int64_t v = (int64_t)(y << b.e_ls);
        v = (int64_t)((uint64_t)v << b.e_ka) >> b.e_ka;
        v = (int64_t)((uint64_t)((v + b.e_rnd) >> b.e_rs) << b.e_ls2);
        v = v < b.e_lo ? b.e_lo : (v > b.e_hi ? b.e_hi : v);
        o[r] = (int)((int64_t)((uint64_t)v << b.e_ko) >> b.e_ko);
      }
    }
    if (GUARD) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t m = st * 256 + 16 * n_col + 4 * kg + r;
        if (m < a.n_out) { *(int *)(yrow + 4 * m) = o[r]; }
      }
    } else {
      const int L = 4 * n_col + kg;
      *(v4i *)(ob + ((L ^ ((n_col >> 1) & 3)) * 16)) = (v4i){o[0], o[1], o[2], o[3]};
    }
  };
  typedef std::integral_constant<bool, true> T;
  typedef std::integral_constant<bool, false> F;
  if (PART != 2 && (PART == 1 || interior)) {
    // The warm-up step exists for stage B's ring: B's first real step reads A's outputs m0 - 128 ... m0 - 1 only (plb.off = 128), i.e.
    // columns 8 .. 15 of the warm-up step, whose windows start at slot 8 R.  The 1 KB pieces entirely below that slot are not
    // loaded (zeros are staged instead): with six steps per chunk the warm-up re-read was +15.6 % of HBM fetches (PMC, round 3).
    {
      const int64_t W0 = a.first + (s0 - 1) * 256 * R - a.pl.off;
      const char *base = (const char *)(xrow + W0);
      const int kskip = (8 * R * 32) / 1024;
#pragma unroll
      for (int k = 0; k < NPCA; k++) {
        const unsigned off = k < NPCA - 2 ? lane16 + 1024u * k : 2u * (unsigned)pc_off[k];
        if (k < NPCA - 2 && k < kskip) { pre[k] = (v4i){0, 0, 0, 0}; }
        else { pre[k] = *(const v4i *)(base + off); }
      }
    }
    body(s0 - 1, T(), F(), T());
    body(s0, F(), F(), T());
    for (int64_t st = s0 + 1; st < s1; st++) { body(st, F(), T(), T()); }
  } else {
    fetch(s0 - 1, F());
    body(s0 - 1, T(), F(), F());
    body(s0, F(), F(), F());
    for (int64_t st = s0 + 1; st < s1; st++) { body(st, F(), T(), F()); }
  }
  if (!GUARD) { flush(s1 - 1); }
}

// (Round 4 built and measured two other forms of this kernel for the interior chunks of the config-5 class, both REMOVED again -- the
// record is profiles/r4_cascade_ablation.txt: (1) a workgroup of four waves sharing stage A's outputs through LDS, two steps per wave
// with every load issued up front and one warm-up per workgroup: +4 % in time; (2) this kernel's step with the ring kernel's data
// movement -- eight aligned non-temporal 1 KB loads per step, every byte once, the halo kept in registers -- and loads TWO steps ahead
// (stage B's fragments in LDS to make room): +2 %.  Neither the access pattern nor the bytes in flight bound this kernel.)

// pa: stage A (decimator) -- in / x / hist / hl / in_stride / n / n_ch as for launch_fir_gen with out_mode 1;
// pb: stage B formats (in = the INT_TYPE, acc, out, lossless_shift) and the output buffer (y, out_stride, int32 containers).
// Returns hipErrorNotSupported when the shapes are not the compiled ones (the caller then runs the two kernels).
hipError_t launch_cascade(const FirParams &pa, const FirGenPlan &pla, const uint32_t *d_fragA, int w_int, int64_t first,
                          const FirParams &pb, const FirGenPlan &plb, const uint32_t *d_fragB, int64_t n_out, hipStream_t s) {
  if (n_out <= 0) { return hipSuccess; }
  GenArgs a, b;
  memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
  a.pl = pla; b.pl = plb;
  a.px = (pa.in.W + (pa.in.S ? 0 : 1) + 7) / 8;
  b.px = (w_int + 7) / 8;
  a.n_slots = 15 * pla.R + 4 * pla.nb;
  const int spl = (a.n_slots + 63) / 64;
  const bool shape_ok = pa.in_eb == 2 && a.px == 2 && pla.pc <= 3 && pla.nb <= 6 && spl == 5 && (a.n_slots * 2 + 63) / 64 <= 9 && pla.R >= 2 &&
                        b.px == 5 && plb.pc <= 2 && plb.nb <= 3 && plb.R == 1 && plb.off == 128 && pb.out_eb == 4;
  a.out_mode = 1; a.w_int = w_int; a.out_simple = 2;
  FirParams pint = pa;                       // stage A's "OUT_TYPE" is the INT_TYPE itself: wrap only
  pint.out = pb.in;
  const bool conv_a = gen_conv_params(pint, 1, w_int, &a) && a.e_rs == 0 && a.e_ls2 == 0 && a.e_ko <= a.e_ka && a.e_hi == INT64_MAX;
  const bool conv_b = gen_conv_params(pb, 0, 0, &b);
  const bool out_ok = ((uintptr_t)pb.y % 16 == 0) && ((pb.out_stride * 4) % 16 == 0);
  if (!shape_ok || !conv_a || !conv_b || !out_ok) { return hipErrorNotSupported; }
  a.corr = gen_rebias_corr(a.px, pla.sum_h);
  b.corr = gen_rebias_corr(b.px, plb.sum_h);
  a.first = first; a.n_out = n_out; b.first = 0; b.n_out = n_out;
  a.n_steps = (n_out + 255) / 256;
  int64_t spw = 8;                           // short spans (see launch_fir_gen) against one (half-loaded) warm-up step per chunk: 4: 0.605, 6: 0.642 - 0.667, 8: 0.653 - 0.669 of the roofline (profiles/r3_span_sweep.txt, last block)
  ACDSP_TUNE_ENV(cspw_env, "ACDSP_CASC_SPW");   // tuning knob: steps per wave of the fused cascade
  if (cspw_env && atoi(cspw_env) > 0) { spw = atoi(cspw_env); }
  if (spw < 2) { spw = 2; }
  a.steps_per_wave = spw;
  a.n16 = (pa.n + 15) / 16 * 16;
  a.out_vec_ok = 1; a.chunk0 = 0;
  b.pad = 0; b.rcp = 0; b.n_slots = 0; b.out_mode = 0; b.w_int = 0; b.out_simple = 0; b.n_steps = a.n_steps; b.steps_per_wave = spw;
  b.n16 = 0; b.obuf_off = 0; b.chunk0 = 0; b.out_vec_ok = 1;
  const int slots_alloc = 15 * pla.R + 4 * 6;
  const int phys = gen_slot_map(a, pla.R, slots_alloc);
  a.obuf_off = a.px * (phys + 1) * 16;
  ACDSP_TUNE_ENV(clp_env, "ACDSP_CASC_LDS_PAD");   // diagnostic: extra LDS bytes per wave (lowers the occupancy)
  const size_t lds_bytes = (size_t)a.obuf_off + 5 * 512 + 1024 + (clp_env ? (size_t)atoi(clp_env) : 0);
  const int64_t n_chunks = (a.n_steps + spw - 1) / spw, fast_chunks = n_out / (spw * 256);
  const v4i *fa = (const v4i *)d_fragA, *fb = (const v4i *)d_fragB;

  // 32-bit limb epilogues: the exact stage results are small enough that byte / 16-bit carry chains in 32-bit registers
  // reproduce them (no 64-bit arithmetic per output).  Conditions: the true |y| of both stages bounded well inside int64
  // (sum |h| x the input range), no ACC_TYPE wrap in stage B, a right shift of 0..31 bits into an OUT_TYPE of <= 31 bits,
  // and for AC_SAT an OUT range that reaches into the high limb.
  static const bool no_limb = getenv("ACDSP_NO_LIMB") != nullptr;   // A/B knob
  bool limb = !no_limb && w_int >= 33 && w_int <= 40 && pla.sum_abs_h < (int64_t(1) << 38) && plb.sum_abs_h < (int64_t(1) << 20);
  const int64_t bound_b = limb ? plb.sum_abs_h * (int64_t(1) << (w_int - 1)) + b.e_rnd : 0;   // |stage-B dot product + rounding constant|
  limb = limb && b.e_ls == 0 && b.e_ls2 == 0 && b.e_rs <= 31 && pb.out.W <= 31 && bound_b < (int64_t(1) << 61) &&
         (pb.acc.W >= 63 || bound_b < (int64_t(1) << (pb.acc.W - 1)));
  if (limb) {
    auto digits = [](int64_t c, int nw, int32_t *dig) {   // balanced base-256 digits, the last one takes the rest
      for (int w = 0; w < nw - 1; w++) {
        int lo = (int)(((c % 256) + 256) % 256);
        if (lo >= 128) { lo -= 256; }
        dig[w] = lo;
        c = (c - lo) / 256;
      }
      dig[nw - 1] = (int32_t)c;
      return c > -(int64_t(1) << 20) && c < (int64_t(1) << 20);
    };
    limb = digits(a.corr, 4, a.dig) && digits(b.corr + b.e_rnd, 6, b.dig);
    if (pb.out.O == ACDSP_SAT) {
      const int hb = pb.out.W - 1 + b.e_rs - 32;           // OUT range in units of the high limb: [-2^hb, 2^hb)
      limb = limb && hb >= 0 && hb <= 29;
      b.l_hb_lo = -(1 << (hb < 0 ? 0 : hb)) - 1; b.l_hb_hi = 1 << (hb < 0 ? 0 : hb);
      b.l_lo = (int32_t)b.e_lo; b.l_hi = (int32_t)b.e_hi; b.l_w = pb.out.W;
    } else {
      b.l_hb_lo = b.l_lo = INT32_MIN; b.l_hb_hi = b.l_hi = INT32_MAX; b.l_w = pb.out.W;
    }
  }
  hipError_t e;
#define ACDSP_CASCADE_LAUNCH(GUARD_, LIMB_, PART_, GRID_)                                                                            \
  e = hipFuncSetAttribute((const void *)cascade_kernel<2, 3, 6, 9, 5, 2, 3, GUARD_, LIMB_, PART_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds_bytes);                                                                                           \
  if (e != hipSuccess) { return e; }                                                                                                 \
  hipLaunchKernelGGL((cascade_kernel<2, 3, 6, 9, 5, 2, 3, GUARD_, LIMB_, PART_>), GRID_, dim3(64), lds_bytes, s, pa, pb, fa, fb, a, b); \
  if ((e = hipGetLastError()) != hipSuccess) { return e; }
  a.xcd_map = 0; a.edge_chunk = 0;
  if (fast_chunks > 0) {
    const dim3 grid((unsigned)fast_chunks, (unsigned)pa.n_ch);
    a.xcd_map = (xcd_map_wanted(false) && ((int64_t)grid.x * grid.y) % 8 == 0) ? 1 : 0;   // off by default here: -3 % on config 5 (profiles/r3_xcd_map.txt)
    {
      if (limb) { ACDSP_CASCADE_LAUNCH(false, true, 1, grid) } else { ACDSP_CASCADE_LAUNCH(false, false, 1, grid) }
      // the complete chunks whose windows leave the call's samples: chunk 0 (history) and possibly the last one (its window ends
      // 16 n_slots - 256 R - off < 16 R samples behind the last output's input: every earlier chunk ends a whole chunk before that)
      a.xcd_map = 0;
      a.edge_chunk = (int32_t)(fast_chunks - 1);
      const dim3 egrid(fast_chunks > 1 ? 2u : 1u, (unsigned)pa.n_ch);
      if (limb) { ACDSP_CASCADE_LAUNCH(false, true, 2, egrid) } else { ACDSP_CASCADE_LAUNCH(false, false, 2, egrid) }
    }
  }
  if (fast_chunks < n_chunks) {
    a.xcd_map = 0;
    a.chunk0 = (int32_t)fast_chunks;
    const dim3 grid((unsigned)(n_chunks - fast_chunks), (unsigned)pa.n_ch);
    if (limb) { ACDSP_CASCADE_LAUNCH(true, true, 0, grid) } else { ACDSP_CASCADE_LAUNCH(true, false, 0, grid) }
  }
#undef ACDSP_CASCADE_LAUNCH
  return hipSuccess;
}

}  // namespace acdsp
