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
// Mapping (v_mfma_i32_32x32x32_i8).  One wave = one channel x a chunk of steps; a step is 32 input slots of 16 samples
// = 512 L outputs.  MFMA column c = input slot S0 + c; its 16 L outputs are the rows of RG = 16 L / 32 row groups:
//     row u = 32 rg + i  ->  input d = u / L inside the slot, phase j = u % L,
//     D_rg[i][c] = sum_kappa A_rg[i][kappa] * X[kappa][c],   A_rg[i][kappa] = E_j[16 HS + d - kappa],
//     X[kappa][c] = x[16 (S0 + c - HS) + kappa],   kappa in [0, 32 NB),  HS = 2 NB - 1 history slots.
// Every row group multiplies the SAME X fragments (one aligned 16-byte LDS read per lane, K block and byte plane);
// only the Toeplitz fragments differ, and they stay in registers.  Operands are split into byte planes exactly as in
// fir_gen.hip (x: 1..4 planes, lower ones re-biased to signed; taps: balanced base-256 digits), products of equal weight
// share an int32 accumulator, the 64-bit recombination runs once per output.  The re-bias correction depends on the
// phase: 128 * sum_k E_j[k] * sum_{p < PX-1} 256^p, a small per-lane table (rows of a lane repeat with period L <= 32).
//
// Data movement.  One 16-sample slot per lane is loaded a step ahead (coalesced: 33..35 consecutive slots), split into
// byte planes (v_perm_b32) and staged in LDS.  Outputs are 8 x (16 for L = 16) the input volume, so the write-out decides
// the speed: every row group (or, for 2-byte outputs, four of them) is converted into a padded LDS tile holding the
// outputs of each column as one contiguous run, and leaves as 8-byte-per-lane stores that form 256-byte runs
// (tools/power_probe: full-wave contiguous stores reach 5.8 - 5.9 TB/s at 4, 8 and 16 bytes per lane alike).
#include <type_traits>
#include <vector>

#include "fir_kernels.hpp"

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

// index into the per-lane phase table of accumulator register r (row i = (r & 3) + 8 (r >> 2) + 4 h; L divides 32, so
// the phase (32 rg + i) % L does not depend on the row group)
template <int L> __device__ constexpr int up_tab_idx(int r) { return L >= 32 ? r : (L == 16 ? (r & 3) + 4 * ((r >> 2) & 1) : (r & 3)); }
template <int L> constexpr int up_tab_size() { return L >= 32 ? 16 : (L == 16 ? 8 : 4); }

}  // namespace

struct UpArgs {
  FirParams p;                // x, in_stride, y, out_stride, formats; p.lossless_shift / p.acc used by mode 0
  int64_t slot0;              // first input slot (16 samples) of the launch; >= HS
  int64_t n_steps;            // steps of 32 slots
  int64_t steps_per_wave;
  int64_t out_off;            // output element index = n * L + j + out_off
  int32_t mode;               // 0: poly_intr ((V << shift) >> sh_j, ACC -> OUT)   1: CIC (wrap to w_int, IN fraction -> OUT)
  int32_t w_int, out_simple;
  uint32_t sh_mask;           // mode 0: bit j set = phase j is a symmetric pair: halve
  const int64_t *corr;        // [L] re-bias correction per phase (mod 2^64)
  // EPI 1 (mode 0, every intermediate inside int32, AC_TRN / AC_RND into AC_WRAP / AC_SAT):
  //   q = (V + (rnd << sh_j)) >> (rs + sh_j);  q = clamp(q, lo, hi);  q = ((q << w) >> w) & mask       (all branch-free)
  // EPI 2 (mode 1, OUT_TYPE has INT_TYPE's fraction and AC_WRAP):  v = (y << w) >> w;  o = ((v << rs) >> rs) & mask
  int32_t e_rs, e_rnd, e_lo, e_hi, e_w;
  uint64_t e_mask;
};

// EPI 0: 64-bit recombination + the generic conversions (any Q / O mode; uniform branches per output).
// EPI 1: poly_intr with every intermediate inside int32 and a shift / clamp / wrap conversion (host-checked).
// EPI 2: CIC with a bit-field wrap conversion.  1 and 2 are branch-free: the step loop stays one basic block.
// PCT: coefficient digit planes compiled in (2 or 3; the fragment array always has 3 per K block).
template <typename TIN, int PX, int PCT, int NBT, int L, int OEB, int EPI>
__global__ void __launch_bounds__(64, 2) fir_up_kernel(UpArgs a, const v4i *__restrict__ frag) {
  constexpr int RG = (16 * L + 31) / 32;                      // row groups per column
  constexpr int HS = 2 * NBT - 1;                             // history slots in front of a column's own slot
  constexpr int NSLOT = 32 + HS;                              // slots staged per step
  constexpr int PLB = 66 * 16;                                // bytes of one plane array: NSLOT slots, a pad slot, private sinks of the surplus lanes
  constexpr int FG = (RG < 256 / (32 * OEB)) ? RG : (256 / (32 * OEB) > 0 ? 256 / (32 * OEB) : 1);   // row groups per write-out
  constexpr int RUN = FG * 32 * OEB;                          // contiguous output bytes of one column per write-out
  constexpr int RUNP = RUN + (OEB == 2 ? 8 : 16);             // padded column pitch of the tile (conflict-free stores)
  constexpr int NACC = PX + PCT - 1;
  constexpr int TS = up_tab_size<L>();
  static_assert(32 % L == 0 || L == 32, "phase tables assume L divides 32");
  static_assert(RG % FG == 0, "row groups per write-out must divide the row groups");
  __shared__ __attribute__((aligned(16))) unsigned char lds[PX * PLB + 32 * RUNP];
  unsigned char *tile = lds + PX * PLB;
  const FirParams &p = a.p;
  const int lane = threadIdx.x;
  const int c = lane & 31, h = lane >> 5;
  const int ch = blockIdx.y;

  v4i A[RG][NBT][PCT];
#pragma unroll
  for (int rg = 0; rg < RG; rg++) {
#pragma unroll
    for (int b = 0; b < NBT; b++) {
#pragma unroll
      for (int q = 0; q < PCT; q++) { A[rg][b][q] = frag[(((size_t)rg * NBT + b) * kUpMaxPC + q) * 64 + lane]; }
    }
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
    corr_t[t] = a.corr[j];
    sh_t |= sh << t;
    corr32_t[t] = (int)a.corr[j] + (a.e_rnd << sh);   // ((V >> sh) + rnd) >> rs == (V + (rnd << sh)) >> (rs + sh)
    shift_t[t] = a.e_rs + (int)sh;
  }

  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  char *yrow = (char *)p.y + ((int64_t)ch * p.out_stride + a.out_off) * OEB;
  const int64_t st0 = (int64_t)blockIdx.x * a.steps_per_wave;
  const int64_t st1 = (st0 + a.steps_per_wave < a.n_steps) ? st0 + a.steps_per_wave : a.n_steps;

  const int sl = lane < NSLOT ? lane : NSLOT - 1;             // surplus lanes repeat the last slot ...
  const int wsl = lane < NSLOT ? lane : lane + 1;             // ... and store it into a private sink (same-address stores serialise)
  v4i pre[sizeof(TIN)];
  auto fetch = [&](int64_t st) {
    const TIN *src = xrow + 16 * (a.slot0 + 32 * st + sl - HS);
#pragma unroll
    for (int q = 0; q < (int)sizeof(TIN); q++) { pre[q] = ((const v4i *)src)[q]; }
  };
  auto stage = [&]() {
    union { v4i v[sizeof(TIN)]; unsigned d[4 * sizeof(TIN)]; } u;
#pragma unroll
    for (int q = 0; q < (int)sizeof(TIN); q++) { u.v[q] = pre[q]; }
#pragma unroll
    for (int pp = 0; pp < PX; pp++) {
      v4i o;
      if (sizeof(TIN) == 2) {
        const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
        o.x = (int)__builtin_amdgcn_perm(u.d[1], u.d[0], sel); o.y = (int)__builtin_amdgcn_perm(u.d[3], u.d[2], sel);
        o.z = (int)__builtin_amdgcn_perm(u.d[5], u.d[4], sel); o.w = (int)__builtin_amdgcn_perm(u.d[7], u.d[6], sel);
      } else {
        o.x = (int)up_gather4(u.d[0], u.d[1], u.d[2], u.d[3], pp); o.y = (int)up_gather4(u.d[4], u.d[5], u.d[6], u.d[7], pp);
        o.z = (int)up_gather4(u.d[8], u.d[9], u.d[10], u.d[11], pp); o.w = (int)up_gather4(u.d[12], u.d[13], u.d[14], u.d[15], pp);
      }
      if (pp < PX - 1) { o ^= (v4i){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u}; }
      *(v4i *)(lds + pp * PLB + wsl * 16) = o;
    }
  };

  // write-out of one finished unit (FG row groups of one step): 32 columns x RUN contiguous bytes, 8 bytes per lane and
  // instruction.  e_unit = output element (before out_off) of column 0, first row of the unit.
  auto flush = [&](int64_t e_unit) {
#pragma unroll
    for (int k = 0; k < 32 * RUN / 512; k++) {
      const int lin = (k * 64 + lane) * 8;
      const int cc = lin / RUN, w = lin % RUN;
      const long val = *(const long *)(tile + cc * RUNP + w);
#ifdef UP_ABLATE_STORES
      if (a.n_steps < 0)
#endif
      *(long *)(yrow + (e_unit + (int64_t)cc * 16 * L) * OEB + w) = val;
    }
  };
  // One step.  VMEM program order: [wait for this step's slots] -> stores of the previous step's last unit -> loads of the
  // next step -> (per unit) stores of the unit before.  hipcc waits vmcnt(0) at the top of the loop (its entry path has
  // the loads as the youngest operations), so every store a step issues AFTER its loads is waited for at the next top:
  // those are the units 0 .. last-1, written out one unit behind the arithmetic -- they had a unit's time or more to land.
  auto body = [&](int64_t st, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    // (single-wave workgroup: the LDS operations of a wave execute in order, no barrier needed)
    stage();
    if (!FIRST) { flush(16 * (a.slot0 + 32 * (st - 1)) * (int64_t)L + 32 * (RG - FG)); }
    fetch(st + 1 < st1 ? st + 1 : st);
    __builtin_amdgcn_sched_barrier(0);
    v4i X[NBT][PX];
#pragma unroll
    for (int b = 0; b < NBT; b++) {
#pragma unroll
      for (int pp = 0; pp < PX; pp++) { X[b][pp] = *(const v4i *)(lds + pp * PLB + (c + 2 * b + h) * 16); }
    }
    const int64_t e_col = 16 * (a.slot0 + 32 * st) * (int64_t)L;   // output element (before out_off) of column 0, row 0
#pragma unroll
    for (int g0 = 0; g0 < RG; g0 += FG) {
      if (g0 > 0) { flush(e_col + 32 * (g0 - FG)); }
#pragma unroll
      for (int gl = 0; gl < FG; gl++) {
        const int rg = g0 + gl;
        v16i acc[NACC];
#pragma unroll
        for (int w = 0; w < NACC; w++) { acc[w] = (v16i){0}; }
#pragma unroll
        for (int b = 0; b < NBT; b++) {
#pragma unroll
          for (int q = 0; q < PCT; q++) {
#pragma unroll
            for (int pp = 0; pp < PX; pp++) {
              acc[pp + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[rg][b][q], X[b][pp], acc[pp + q], 0, 0, 0);
            }
          }
        }
        // epilogue: lane (c, h), register r: row i = (r & 3) + 8 (r >> 2) + 4 h of row group rg.  The plane accumulators are
        // recombined pairwise in 32 bits first (|acc| < 2^22, so a + (b << 8) is exact), then in 64 bits.
#pragma unroll
        for (int g = 0; g < 4; g++) {
          int64_t o[4];
          int o32[4];
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int r = 4 * g + rr;
            const int t = up_tab_idx<L>(r);
            int pr[(NACC + 1) / 2];
#pragma unroll
            for (int m = 0; m < (NACC + 1) / 2; m++) {
              pr[m] = (2 * m + 1 < NACC) ? (int)(((unsigned)acc[2 * m + 1][r] << 8) + (unsigned)acc[2 * m][r]) : acc[2 * m][r];
            }
            if constexpr (EPI == 1) {
              static_assert(EPI != 1 || NACC <= 4, "32-bit epilogue: two accumulator pairs");
              // branch-free (a uniform branch per output would split the step loop into hundreds of basic blocks, each
              // with its own s_waitcnt vmcnt(0) on the write-out stores): clamp bounds are the int32 range when OUT_TYPE
              // wraps, the wrap shift is 0 for full-width containers
              const int y32 = (int)(((unsigned)pr[(NACC + 1) / 2 - 1] << 16) + (unsigned)pr[0]) + corr32_t[t];
              int q = y32 >> shift_t[t];
              q = q < a.e_lo ? a.e_lo : (q > a.e_hi ? a.e_hi : q);
              q = (int)(((unsigned)((int)((unsigned)q << a.e_w) >> a.e_w)) & (unsigned)a.e_mask);
              o32[rr] = q;
            } else {
              // y = corr + sum_m sext(pr[m]) << 16 m, in 32-bit halves (carry chains instead of 64-bit shifts)
              uint64_t y;
              {
                unsigned lo = (unsigned)corr_t[t], hi = (unsigned)((uint64_t)corr_t[t] >> 32);
#pragma unroll
                for (int m = 0; m < (NACC + 1) / 2; m++) {
                  if (m == 0) { const unsigned s0 = lo + (unsigned)pr[0]; hi += (unsigned)(pr[0] >> 31) + (s0 < lo); lo = s0; }
                  else if (m == 1) { const unsigned t1 = (unsigned)pr[1] << 16, s1 = lo + t1; hi += (unsigned)(pr[1] >> 16) + (s1 < lo); lo = s1; }
                  else if (m == 2) { hi += (unsigned)pr[2]; }
                  else { hi += (unsigned)pr[3] << 16; }
                }
                y = ((uint64_t)hi << 32) | lo;
              }
              if constexpr (EPI == 2) {
                // CIC: wrap to INT_TYPE, then to OUT_TYPE (same fraction, AC_WRAP); both are wider than 32 bits (host-checked),
                // so the wraps are bit-field extracts of the high word (64-bit shifts run at a quarter of the 32-bit rate)
                int hi = (int)(y >> 32);
                hi = (int)((unsigned)hi << a.e_w) >> a.e_w;
                hi = (int)((unsigned)((int)((unsigned)hi << a.e_rs) >> a.e_rs) & (unsigned)a.e_mask);
                o[rr] = (int64_t)(((uint64_t)(unsigned)hi << 32) | (uint32_t)y);
              } else if (a.mode == 1) {
                o[rr] = requant64(wrap64((int64_t)y, a.w_int, 1), p.in.F, p.out);
              } else {
                const int64_t v = (int64_t)(y << p.lossless_shift) >> ((sh_t >> t) & 1u);
                o[rr] = requant64(v, p.acc.F, p.out);
              }
            }
          }
          unsigned char *dst = tile + c * RUNP + (gl * 32 + 8 * g + 4 * h) * OEB;
          if (EPI == 1) {
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
        // keep the row groups apart: interleaved, their accumulators and 64-bit temporaries exceed the register file
        if constexpr (EPI != 1) { __builtin_amdgcn_sched_barrier(0); }
      }
    }
  };
  if (st0 >= st1) { return; }
  fetch(st0);
  body(st0, std::integral_constant<bool, true>());
  for (int64_t st = st0 + 1; st < st1; st++) { body(st, std::integral_constant<bool, false>()); }
  flush(16 * (a.slot0 + 32 * (st1 - 1)) * (int64_t)L + 32 * (RG - FG));
}

// ---------------------------------------------------------------------------------------------
// host: digit planes, Toeplitz fragments, correction table
// ---------------------------------------------------------------------------------------------
bool fir_up_plan(const int64_t *E, int L, int nt, int px, FirUpPlan *pl, std::vector<uint32_t> *frag, std::vector<int64_t> *corr) {
  if (L < 2 || L > 32 || (32 % L) != 0 || nt < 1 || px < 1 || px > 4) { return false; }
  // K blocks: the window of a column spans its 16 samples and nt - 1 earlier ones
  int nb = 1;
  while (16 * (2 * nb - 1) < nt - 1) { nb++; }
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
  pl->L = L; pl->nt = nt; pl->pc = pc; pl->nb = nb; pl->hs = 2 * nb - 1;
  const int RG = (16 * L + 31) / 32, HS = 2 * nb - 1;
  frag->assign((size_t)RG * nb * kUpMaxPC * 64 * 4, 0u);
  for (int rg = 0; rg < RG; rg++) {
    for (int b = 0; b < nb; b++) {
      for (int q = 0; q < kUpMaxPC; q++) {
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 31, kg = lane >> 5;
          const int u = 32 * rg + i;
          if (u >= 16 * L) { continue; }
          const int d = u / L, j = u % L;
          for (int dw = 0; dw < 4; dw++) {
            uint32_t word = 0;
            for (int bj = 0; bj < 4; bj++) {
              const int kappa = 32 * b + 16 * kg + 4 * dw + bj;
              const int tap = 16 * HS + d - kappa;
              const int8_t val = (tap >= 0 && tap < nt) ? dig[q][(size_t)j * nt + tap] : (int8_t)0;
              word |= (uint32_t)(uint8_t)val << (8 * bj);
            }
            (*frag)[((((size_t)rg * nb + b) * kUpMaxPC + q) * 64 + lane) * 4 + dw] = word;
          }
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
  if (nb < 1 || nb > 2 || (L != 8 && L != 16)) { return false; }
  if (in_eb == 2 && px == 2) { return out_eb == 2 || out_eb == 8; }
  if (in_eb == 4 && px == 4) { return nb == 1 && out_eb == 8; }
  return false;
}

// compiled shapes: poly_intr = int16 samples, 3 digit planes (the pair taps E_j - E_cj have 17 bits), 2- or 8-byte outputs;
// CIC = int16 / int32 samples, 2 digit planes (boxcar^N taps of the BASELINE shapes fit 16 bits), 8-byte outputs (2-byte ones
// for int16 samples)
template <typename TIN, int PX, int PCT, int NBT, int L>
static hipError_t launch_up_oeb(const UpArgs &a, const uint32_t *d_frag, int out_eb, int epi, dim3 grid, hipStream_t s) {
  const v4i *f = (const v4i *)d_frag;
  if (out_eb == 8) {
    if (epi == 2) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 8, 2>), grid, dim3(64), 0, s, a, f); }
    else { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 8, 0>), grid, dim3(64), 0, s, a, f); }
  } else if (out_eb == 2) {
    if constexpr (sizeof(TIN) == 2) {
      if (epi == 1) {
        if constexpr (PCT == 3) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 1>), grid, dim3(64), 0, s, a, f); }
        else { return hipErrorNotSupported; }
      } else if (epi == 2) {
        if constexpr (PCT == 2) { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 2>), grid, dim3(64), 0, s, a, f); }
        else { return hipErrorNotSupported; }
      } else { hipLaunchKernelGGL((fir_up_kernel<TIN, PX, PCT, NBT, L, 2, 0>), grid, dim3(64), 0, s, a, f); }
    } else { return hipErrorNotSupported; }
  } else {
    return hipErrorNotSupported;
  }
  return hipGetLastError();
}

template <typename TIN, int PX, int PCT, int NBT>
static hipError_t launch_up_l(const UpArgs &a, const uint32_t *d_frag, int L, int out_eb, int epi, dim3 grid, hipStream_t s) {
  switch (L) {
    case 8: return launch_up_oeb<TIN, PX, PCT, NBT, 8>(a, d_frag, out_eb, epi, grid, s);
    case 16: return launch_up_oeb<TIN, PX, PCT, NBT, 16>(a, d_frag, out_eb, epi, grid, s);
    default: return hipErrorNotSupported;
  }
}

// Input slots [slot0, slot0 + 32 n_steps) of every channel; the caller covers everything else with the VALU kernels.
hipError_t launch_fir_up(const FirParams &p, const FirUpPlan &pl, int px, const uint32_t *d_frag, const int64_t *d_corr, int mode, int w_int,
                         int out_simple, uint32_t sh_mask, int64_t max_abs_v, int64_t slot0, int64_t n_steps, int64_t out_off, hipStream_t s) {
  if (n_steps <= 0) { return hipSuccess; }
  if (!fir_up_shape_ok(p.in_eb, px, pl.nb, pl.L, p.out_eb) || slot0 < pl.hs) { return hipErrorNotSupported; }
  UpArgs a;
  a.p = p; a.slot0 = slot0; a.n_steps = n_steps; a.out_off = out_off; a.mode = mode; a.w_int = w_int; a.out_simple = out_simple;
  a.sh_mask = sh_mask; a.corr = d_corr;
  a.e_rs = a.e_rnd = a.e_w = 0; a.e_lo = INT32_MIN; a.e_hi = INT32_MAX; a.e_mask = ~uint64_t(0);
  int epi = 0;
  const int rs = p.acc.F - p.out.F;
  if (mode == 0 && px == 2 && p.lossless_shift == 0 && rs >= 0 && rs <= 30 && p.out.W <= 32 && p.out_eb == 2 &&
      (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) && (p.out.O == ACDSP_WRAP || p.out.O == ACDSP_SAT) &&
      max_abs_v >= 0 && max_abs_v < (int64_t(1) << 30)) {
    // 32-bit epilogue: poly_intr, no left shift into ACC_TYPE, |V| (+ rounding constant) inside int32
    epi = 1;
    a.e_rs = rs;
    a.e_rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (1 << (rs - 1)) : 0;
    if (p.out.O == ACDSP_SAT) { a.e_lo = (int32_t)p.out.lo; a.e_hi = (int32_t)p.out.hi; }
    else {
      a.e_w = 32 - p.out.W;
      if (!p.out.S) { a.e_mask = (uint64_t)((uint32_t)(-1) >> (32 - p.out.W)); }
    }
  } else if (mode == 1 && out_simple >= 1) {
    // bit-field wraps of the high word: to INT_TYPE, then (out_simple 1) to an OUT_TYPE of the same fraction with AC_WRAP
    if (w_int > 32 && (out_simple == 2 || p.out.W > 32)) {
      epi = 2;
      a.e_w = 64 - w_int;
      if (out_simple == 1) {
        a.e_rs = 64 - p.out.W;
        if (!p.out.S) { a.e_mask = (uint64_t)(~uint32_t(0) >> (64 - p.out.W)); }
      }
    }
  }
  // >= ~8192 waves when the problem allows it
  int64_t spw = (n_steps * p.n_ch + 8191) / 8192;
  if (spw < 1) { spw = 1; }
  a.steps_per_wave = spw;
  dim3 grid((unsigned)((n_steps + spw - 1) / spw), (unsigned)p.n_ch);
  if (p.in_eb == 2) {
    if (mode == 0) {
      return pl.nb == 1 ? launch_up_l<int16_t, 2, 3, 1>(a, d_frag, pl.L, p.out_eb, epi, grid, s)
                        : launch_up_l<int16_t, 2, 3, 2>(a, d_frag, pl.L, p.out_eb, epi, grid, s);
    }
    if (pl.pc > 2 || pl.nb != 1) { return hipErrorNotSupported; }
    return launch_up_l<int16_t, 2, 2, 1>(a, d_frag, pl.L, p.out_eb, epi, grid, s);
  }
  if (pl.pc > 2 || pl.nb != 1) { return hipErrorNotSupported; }
  return launch_up_l<int32_t, 4, 2, 1>(a, d_frag, pl.L, p.out_eb, epi, grid, s);
}

}  // namespace acdsp
