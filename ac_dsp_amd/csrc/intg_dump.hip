// intg_dump.hip -- integrate-and-dump: reference include/ac_dsp/ac_intg_dump.h:93-147.
//
// The reference walks an interleaved stream (round-major, channel-minor): per block it reads n_sample, then adds one
// sample per channel and round into ACC_TYPE temp[i] (`temp[i] = temp[i] + data_in`, every add quantised, :97) and, in
// round j == n_sample, writes OUT(temp[i]) and clears it (:98-102).  A block whose n_sample is 0 or > NS runs NS rounds
// and dumps nothing: its sums carry into the next block.  Here one thread owns one (object, block, channel): it starts
// from the handle's temp[] (first chain of the call) or 0, replays the rounds of its carry chain in order and writes
// the block's output; one more thread per (object, channel) leaves the trailing, undumped sum in the handle.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "fir_kernels.hpp"

namespace acdsp {

__device__ inline i128 id_shl128(i128 v, int s) { return (i128)((u128)v << s); }

__device__ int64_t intg_chain(const IntgDumpParams &p, int obj, int i, int first_blk, int last_blk) {
  int64_t acc = (first_blk == 0) ? p.temp[((int64_t)obj * p.chn) + i] : 0;
  if (p.lossless) {   // wrapping ACC_TYPE with at least IN_TYPE's fraction bits: every add is exact mod 2^W -> one wrap at the end
    uint64_t sum = (uint64_t)acc;
    const int sh = p.acc.F - p.in.F;
    for (int b = first_blk; b <= last_blk; b++) {
      const int64_t r0 = p.blk_off[b];
      for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
        sum += (uint64_t)load_raw(p.x, (int64_t)obj * p.in_stride + (r0 + r) * p.chn + i, p.in_eb, p.in.S) << sh;
      }
    }
    return wrap64((int64_t)sum, p.acc.W, p.acc.S);
  }
  const int f = p.in.F > p.acc.F ? p.in.F : p.acc.F;
  for (int b = first_blk; b <= last_blk; b++) {
    const int64_t r0 = p.blk_off[b];
    for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
      const int64_t x = load_raw(p.x, (int64_t)obj * p.in_stride + (r0 + r) * p.chn + i, p.in_eb, p.in.S);
      acc = requant128(id_shl128((i128)acc, f - p.acc.F) + id_shl128((i128)x, f - p.in.F), f, p.acc);
    }
  }
  return acc;
}

__global__ void intg_dump_kernel(IntgDumpParams p, int64_t *temp_next) {
  const int obj = blockIdx.y;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_work = (int64_t)(p.n_blocks + 1) * p.chn;     // blocks, then the state slot
  if (tid >= n_work) { return; }
  const int b = (int)(tid / p.chn), i = (int)(tid % p.chn);
  if (b < p.n_blocks) {
    if (p.blk_out[b] < 0) { return; }                            // no dump in this block
    const int64_t acc = intg_chain(p, obj, i, p.blk_chain[b], b);
    store_raw(p.y, (int64_t)obj * p.out_stride + p.blk_out[b] * p.chn + i, p.out_eb, requant64(acc, p.acc.F, p.out));
  } else {
    int64_t acc = 0;
    if (p.n_blocks == 0) { acc = p.temp[(int64_t)obj * p.chn + i]; }
    else if (p.blk_out[p.n_blocks - 1] < 0) { acc = intg_chain(p, obj, i, p.blk_chain[p.n_blocks - 1], p.n_blocks - 1); }
    temp_next[(int64_t)obj * p.chn + i] = acc;
  }
}

// Lossless class, every block dumps, nothing carried in: a workgroup takes 256 / CHN consecutive blocks of one object, pulls
// their (contiguous) samples into LDS with coalesced loads and lets one thread per (block, channel) add them up from there.
// (The general kernel's per-thread walk fetches 8-byte pieces 2 * CHN * n_sample bytes apart: 1.6 TB/s on the bench row.)
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <typename TIN>
__global__ void __launch_bounds__(256) intg_dump_tile_kernel(IntgDumpParams p, int blocks_per_wg, int lds_elems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char id_lds[];
  TIN *tile = (TIN *)id_lds;
  const int obj = blockIdx.y;
  const int b0 = blockIdx.x * blocks_per_wg;
  const int b1 = (b0 + blocks_per_wg < p.n_blocks) ? b0 + blocks_per_wg : p.n_blocks;
  const int64_t e0 = p.blk_off[b0] * p.chn;                                        // first element of the span
  const int64_t e1 = (p.blk_off[b1 - 1] + p.blk_rounds[b1 - 1]) * p.chn;
  const TIN *row = (const TIN *)p.x + (int64_t)obj * p.in_stride;
  const bool staged = e1 - e0 <= lds_elems;
  // LDS image padded by one dword per 512 bytes: the per-thread walks below start a block length apart (a power of two
  // in the common case) and would otherwise all sit on one bank
  constexpr int kPerDw = 4 / (int)sizeof(TIN) > 0 ? 4 / (int)sizeof(TIN) : 1;      // elements per pad
  constexpr int kShift = sizeof(TIN) == 2 ? 8 : (sizeof(TIN) == 4 ? 7 : 6);        // log2(elements per 512 bytes)
  auto pe = [&](int64_t e) -> int64_t { return e + (e >> kShift) * kPerDw; };
  if (staged) {
    const int64_t n_e = e1 - e0;
    if (sizeof(TIN) < 8 && ((uintptr_t)(row + e0) % 16) == 0) {                    // 16-byte loads, dword stores into the padded image
      constexpr int EPV = 16 / (int)sizeof(TIN);
      const int64_t n_v = n_e / EPV;
      for (int64_t v = threadIdx.x; v < n_v; v += 256) {
        const v4i_t q = ((const v4i_t *)(row + e0))[v];
        int *dst = (int *)(tile + pe(v * EPV));                                    // a 16-byte chunk never straddles a pad
        dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
      }
      for (int64_t e = n_v * EPV + threadIdx.x; e < n_e; e += 256) { tile[pe(e)] = row[e0 + e]; }
    } else {
      for (int64_t e = threadIdx.x; e < n_e; e += 256) { tile[pe(e)] = row[e0 + e]; }
    }
    __syncthreads();
  }
  const int b = b0 + (int)threadIdx.x / p.chn, i = (int)threadIdx.x % p.chn;
  if (b >= b1 || (int)threadIdx.x >= blocks_per_wg * p.chn) { return; }
  const int64_t base = p.blk_off[b] * p.chn - e0 + i;
  uint64_t sum = 0;
  const int sh = p.acc.F - p.in.F;
  for (int64_t r = 0; r < p.blk_rounds[b]; r++) {
    int64_t x = staged ? (int64_t)tile[pe(base + r * p.chn)] : (int64_t)row[e0 + base + r * p.chn];
    if (!p.in.S) { x &= (int64_t)((uint64_t)(-1) >> (64 - 8 * (int)sizeof(TIN))); }
    sum += (uint64_t)x << sh;
  }
  const int64_t acc = wrap64((int64_t)sum, p.acc.W, p.acc.S);
  store_raw(p.y, (int64_t)obj * p.out_stride + p.blk_out[b] * p.chn + i, p.out_eb, requant64(acc, p.acc.F, p.out));
}

// Streaming kernel for the common shape: lossless class, every block dumps the same number of rounds, CHN divides the
// elements of a 16-byte load and a block is a whole number of loads.  No LDS: a lane adds the elements of its loads into CHN
// partial sums (element j of any load belongs to channel j % CHN, because blocks and loads start on channel 0), the lanes
// of a block (GS = 1 .. 64 of them; blocks longer than a wave's 1 KB accumulate LPR loads per lane first) combine with DPP
// adds (quad_perm, row_half_mirror, row_mirror: every lane of a 16-lane row ends with the row's sum; row_bcast15 / 31 carry
// it into the last row of a 32- / 64-lane group), and the last min(GS, 16) lanes of the group convert and store one
// channel each.  Eight independent 16-byte loads per lane are in flight per batch; a wave streams a contiguous span of
// one object's row.  (First version: ds_bpermute butterfly + the general 128-bit requant inlined per load: 0.59 ms on the
// bench row, a third of the wave time waiting on the five dependent LDS round trips.)
struct IdConv {               // branch-free ACC -> OUT conversion (host-checked: AC_TRN / AC_RND into AC_WRAP / AC_SAT, < 2^62)
  int32_t ok, sh, ka, rs, ls2, ko;
  uint64_t am, om;            // masks after the sign-extending wraps (all ones for signed types)
  int64_t rnd, lo, hi;
};

template <int CTRL, int ROWMASK>
__device__ inline int dpp_get(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, false); }
template <int CTRL, int ROWMASK>
__device__ inline int64_t dpp_get(int64_t x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, ROWMASK, 0xF, false);
  return (int64_t)(((uint64_t)(unsigned)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROWMASK>
__device__ inline int dpp_sum(int x) { return x + __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, false); }
template <int CTRL, int ROWMASK>
__device__ inline int64_t dpp_sum(int64_t x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, ROWMASK, 0xF, false);
  return x + (int64_t)(((uint64_t)(unsigned)hi << 32) | (unsigned)lo);
}

// all-reduce steps that leave every lane of the group with the group's sum
template <typename TS> __device__ inline TS id_xor16_sum(TS t) {
  if constexpr (sizeof(TS) == 4) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false); return (TS)(r[0] + r[1]); }
  else {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)((uint64_t)t >> 32), (unsigned)((uint64_t)t >> 32), false, false);
    return (TS)((((uint64_t)hi[0] << 32) | lo[0]) + (((uint64_t)hi[1] << 32) | lo[1]));
  }
}
template <typename TS> __device__ inline TS id_xor32_sum(TS t) {
  if constexpr (sizeof(TS) == 4) { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false); return (TS)(r[0] + r[1]); }
  else {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)((uint64_t)t >> 32), (unsigned)((uint64_t)t >> 32), false, false);
    return (TS)((((uint64_t)hi[0] << 32) | lo[0]) + (((uint64_t)hi[1] << 32) | lo[1]));
  }
}

// SCAT (CHN 2 or 4, groups of >= CHN lanes): the first one / two butterfly levels are a reduce-scatter -- a lane keeps half of
// its channels and receives the partner's partial sums of those -- so the remaining levels move one value per lane instead
// of CHN.  The half a lane keeps is chosen by lane bit 0 / 1 XOR lane bit 2, which makes the assignment symmetric under the
// lane reversals of row_half_mirror and row_mirror; lanes 0 .. CHN-1 of a group end with channel 2 * bit0 + bit1 (CHN 4) or
// bit0 (CHN 2) and store it.
// lpb > 0 (round 4): blocks of lpb lane-loads that are NOT a multiple of 64 (NS = 1000 ...): one block per reduce, its loads start at
// lane-load red * lpb wherever that falls, and the lanes of a block's last load that lie past its end contribute nothing.
template <typename TIN, int CHN, bool SGN, bool SCAT>
__global__ void __launch_bounds__(256) intg_dump_stream_kernel(IntgDumpParams p, IdConv cv, int gs, int lpr, int64_t reds_per_wave, int64_t n_reds, int lpb = 0) {
  typedef typename std::conditional<sizeof(TIN) == 2, int, int64_t>::type TS;   // int16: rounds < 2^15 keep the sums inside int32
  constexpr int EPV = 16 / (int)sizeof(TIN);
#ifndef ID_BATCH
#define ID_BATCH 8
#endif
  constexpr int BATCH = ID_BATCH;
  static_assert(!SCAT || CHN == 2 || CHN == 4, "reduce-scatter levels exist for 2 and 4 channels");
  const int lane = threadIdx.x & 63;
  const int obj = blockIdx.y;
  const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: scalar address math
  const int64_t r0 = wave * reds_per_wave;
  const int64_t r1 = (r0 + reds_per_wave < n_reds) ? r0 + reds_per_wave : n_reds;
  if (r0 >= r1) { return; }
  const int bpr = 64 / gs;                                      // blocks per reduce
  const bool b0 = ((lane ^ (lane >> 2)) & 1) != 0, b1 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
  // writer lanes and their channel
  const int i_grp = lane & (gs - 1);
  const int nwl = SCAT ? CHN : (gs < 16 ? gs : 16);             // plain butterfly: the group's last nwl lanes (row_bcast steps)
  const int wl = SCAT ? i_grp : i_grp - (gs - nwl);
  const int my_ch = SCAT ? (CHN == 4 ? 2 * (lane & 1) + ((lane >> 1) & 1) : (lane & 1)) : 0;
  TS acc[CHN];
#pragma unroll
  for (int c = 0; c < CHN; c++) { acc[c] = 0; }
  const int64_t q0 = r0 * lpr, q1 = r1 * lpr;                   // 1 KB wave-loads of this wave
  const v4i_t *row = (const v4i_t *)((const TIN *)p.x + (int64_t)obj * p.in_stride);
  const v4i_t *src = row + 64 * q0 + lane;
  int k_in_red = 0;
  int64_t red = r0;
  int64_t f_red = r0;                                            // ragged blocks: (block, load in block) of the next load to issue
  int f_k = 0;
  for (int64_t q = q0; q < q1; q += BATCH, src += 64 * BATCH) {
    const int rem = (int)(q1 - q < BATCH ? q1 - q : BATCH);
    v4i_t v[BATCH];
    if (lpb > 0) {
#pragma unroll
      for (int k = 0; k < BATCH; k++) {
        const int in_blk = 64 * f_k + lane;
        const bool ok = k < rem && in_blk < lpb;
        // dead loads of a partial last batch (k >= rem) re-read the wave's first block: behind the last valid load f_red == r1, whose
        // first lane-load is past the consumed samples -- and past the allocation for the last object of a dense block
        const v4i_t t = __builtin_nontemporal_load(row + (k < rem ? f_red : r0) * lpb + (ok ? in_blk : 0));
        v[k] = ok ? t : (v4i_t){0, 0, 0, 0};
        if (k < rem) { if (++f_k == lpr) { f_k = 0; f_red++; } }
      }
    } else {
#pragma unroll
    for (int k = 0; k < BATCH; k++) { v[k] = __builtin_nontemporal_load(src + 64 * (k < rem ? k : rem - 1)); }   // read once: streaming policy (6.05 -> 6.75 TB/s in tools/copy_probe)
    }
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      if (k >= rem) { break; }
      const unsigned d[4] = {(unsigned)v[k].x, (unsigned)v[k].y, (unsigned)v[k].z, (unsigned)v[k].w};
      if constexpr (sizeof(TIN) == 2 && CHN <= 4) {
        // pairs of elements of one channel -> one dword (v_perm_b32) -> v_dot2 with (1, 1) adds both, sign- or zero-extended
        typedef short v2s __attribute__((ext_vector_type(2)));
        typedef unsigned short v2us __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int m = 0; m < 4; m++) {
          // CHN 4: (e_c, e_c+4), c = m;  CHN 2: (e_c, e_c+2) and (e_c+4, e_c+6), c = m & 1;  CHN 1: the dwords as they are
          unsigned w;
          if (CHN == 4) { w = __builtin_amdgcn_perm(d[2 + (m >> 1)], d[m >> 1], (m & 1) ? 0x07060302u : 0x05040100u); }
          else if (CHN == 2) { w = __builtin_amdgcn_perm(d[2 * (m >> 1) + 1], d[2 * (m >> 1)], (m & 1) ? 0x07060302u : 0x05040100u); }
          else { w = d[m]; }
          const int c = CHN == 4 ? m : (CHN == 2 ? (m & 1) : 0);
          if (SGN) { acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, w), (v2s){1, 1}, acc[c], false); }
          else { acc[c] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(v2us, w), (v2us){1, 1}, (unsigned)acc[c], false); }
        }
      } else {
#pragma unroll
        for (int j = 0; j < EPV; j++) {
          TS e;
          if constexpr (sizeof(TIN) == 2) {
            const unsigned w = d[j >> 1];
            e = (j & 1) ? (SGN ? (int)w >> 16 : (int)(w >> 16)) : (SGN ? (int)(short)w : (int)(w & 0xffffu));
          } else {
            e = SGN ? (int64_t)(int)d[j] : (int64_t)d[j];
          }
          acc[j % CHN] += e;
        }
      }
      if (++k_in_red == lpr) {
        k_in_red = 0;
        TS mine;
        if constexpr (SCAT) {
          TS t;
          if constexpr (CHN == 4) {
            TS kx = b0 ? acc[2] : acc[0], ky = b0 ? acc[3] : acc[1];
            const TS sx = b0 ? acc[0] : acc[2], sy = b0 ? acc[1] : acc[3];
            kx += dpp_get<0xB1, 0xF>(sx); ky += dpp_get<0xB1, 0xF>(sy);
            t = b1 ? ky : kx;
            const TS st = b1 ? kx : ky;
            t += dpp_get<0x4E, 0xF>(st);
          } else {
            t = b0 ? acc[1] : acc[0];
            const TS st = b0 ? acc[0] : acc[1];
            t += dpp_get<0xB1, 0xF>(st);
            if (gs >= 4) { t = dpp_sum<0x4E, 0xF>(t); }
          }
          if (gs >= 8) { t = dpp_sum<0x141, 0xF>(t); }           // row_half_mirror
          if (gs >= 16) { t = dpp_sum<0x140, 0xF>(t); }          // row_mirror
          if (gs >= 32) { t = id_xor16_sum(t); }
          if (gs >= 64) { t = id_xor32_sum(t); }
          mine = t;
        } else {
#pragma unroll
          for (int c = 0; c < CHN; c++) {
            TS t = acc[c];
            if (gs >= 2) { t = dpp_sum<0xB1, 0xF>(t); }            // quad_perm [1,0,3,2]
            if (gs >= 4) { t = dpp_sum<0x4E, 0xF>(t); }            // quad_perm [2,3,0,1]
            if (gs >= 8) { t = dpp_sum<0x141, 0xF>(t); }           // row_half_mirror (the quads are uniform by now)
            if (gs >= 16) { t = dpp_sum<0x140, 0xF>(t); }          // row_mirror
            if (gs >= 32) { t = dpp_sum<0x142, 0xA>(t); }          // row_bcast15 into rows 1 and 3
            if (gs >= 64) { t = dpp_sum<0x143, 0xC>(t); }          // row_bcast31 into rows 2 and 3
            acc[c] = t;
          }
          mine = acc[0];
        }
        const int64_t b = red * bpr + lane / gs;
#pragma unroll
        for (int pi = 0; pi < (SCAT ? 1 : CHN); pi++) {
          if (pi * nwl >= CHN) { break; }
          int cch = my_ch;
          if constexpr (!SCAT) {
            cch = pi * nwl + wl;
#pragma unroll
            for (int c = 1; c < CHN; c++) { mine = (cch == c) ? acc[c] : mine; }
          }
          if (wl >= 0 && wl < nwl && cch < CHN) {
            const int64_t a = (int64_t)(((uint64_t)((int64_t)((uint64_t)(int64_t)mine << (cv.sh + cv.ka)) >> cv.ka)) & cv.am);   // wrap to ACC_TYPE
            int64_t qv = (int64_t)((uint64_t)((a + cv.rnd) >> cv.rs) << cv.ls2);
            qv = qv < cv.lo ? cv.lo : (qv > cv.hi ? cv.hi : qv);
            const int64_t o = (int64_t)(((uint64_t)((int64_t)((uint64_t)qv << cv.ko) >> cv.ko)) & cv.om);
            store_raw(p.y, (int64_t)obj * p.out_stride + b * CHN + cch, p.out_eb, o);
          }
        }
#pragma unroll
        for (int c = 0; c < CHN; c++) { acc[c] = 0; }
        red++;
      }
    }
  }
}

// Batched form of the streaming kernel for int16 rows whose blocks fit one 1 KB wave-load (LPR = 1).  The eight loads of a
// batch leave 8 * CHN partial sums in every lane, and the GS lanes of a block combine them with a reduce-scatter: at each
// butterfly level a lane keeps half of its values and receives the partner's partial sums of those (2 selects + 1 DPP add
// per surviving value; the 16- and 32-lane levels are one v_permlane*_swap + 1 add), so every (load, block, channel) sum
// ends in exactly one lane of its group (in GS / (8 CHN) lanes when the group is larger than the batch's value count,
// the last levels being plain all-reduce steps).  One ACC -> OUT conversion and one store instruction per batch on all 64
// lanes -- a 256-byte run of int32 outputs for 8 * CHN = GS -- instead of one of each per load on 8 lanes: the per-load
// kernel above spends 59 VALU instructions per KB (profiles/r2_intgdump_rocprof.txt), this one about 25.
// The half kept at level l is chosen by keep bit k_l of the lane number: k0 = b0 ^ b2, k1 = b1 ^ b2, k_l = b_l above, which
// is symmetric under the lane reversal of row_half_mirror (level 2); levels 0, 1, 3 are the XOR partners quad_perm
// [1,0,3,2], [2,3,0,1] and row_ror:8.
template <int L> __device__ inline int id_partner(int x) {
  static_assert(L >= 0 && L < 4, "DPP levels");
  constexpr int CTRL = L == 0 ? 0xB1 : (L == 1 ? 0x4E : (L == 2 ? 0x141 : 0x128));
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false);
}
// lanes [31:16] of a trade places with lanes [15:0] of b (L = 4), or the wave halves (L = 5): r[0] + r[1] is a's pair sum
// in the lower half of each lane pair's span and b's in the upper half
template <int L> __device__ inline int id_swap_sum(int a, int b) {
  static_assert(L == 4 || L == 5, "permlane swap levels");
  if constexpr (L == 4) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false); return (int)(r[0] + r[1]); }
  else { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false); return (int)(r[0] + r[1]); }
}
__device__ inline bool id_keep_bit(int level, int lane) {
  return level < 2 ? (((lane >> level) ^ (lane >> 2)) & 1) != 0 : ((lane >> level) & 1) != 0;
}
template <int L, int LG, int N, int VMAX>
__device__ inline void id_reduce_levels(int (&v)[VMAX], int lane) {
  if constexpr (L < LG) {
    if constexpr (N > 1) {
      const bool kb = id_keep_bit(L, lane);
#pragma unroll
      for (int j = 0; j < N / 2; j++) {
        const int a = v[2 * j], b = v[2 * j + 1];
        if constexpr (L < 4) { v[j] = (kb ? b : a) + id_partner<L>(kb ? a : b); }
        else { v[j] = id_swap_sum<L>(a, b); }
      }
      id_reduce_levels<L + 1, LG, N / 2, VMAX>(v, lane);
    } else {
      if constexpr (L < 4) { v[0] += id_partner<L>(v[0]); }
      else { v[0] = id_swap_sum<L>(v[0], v[0]); }
      id_reduce_levels<L + 1, LG, 1, VMAX>(v, lane);
    }
  }
}
constexpr int id_log2(int x) { return x <= 1 ? 0 : 1 + id_log2(x / 2); }

template <int CHN, int GS, bool SGN>
__global__ void __launch_bounds__(256) intg_dump_batch_kernel(IntgDumpParams p, IdConv cv, int64_t loads_per_wave, int64_t n_loads, int xcd_map) {
  constexpr int BATCH = 8, V = BATCH * CHN, LG = id_log2(GS), LV = id_log2(V);
  constexpr int NSC = LG < LV ? LG : LV;         // reduce-scatter levels; the remaining LG - NSC levels are all-reduce steps
  constexpr int NREM = V >> NSC;                 // values a lane ends with
  constexpr int BPL = 64 / GS;                   // blocks per wave-load
  typedef short v2s __attribute__((ext_vector_type(2)));
  typedef unsigned short v2us __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  int bx, obj;
  xcd_remap(xcd_map, bx, obj);
  const int64_t wave = (int64_t)bx * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t q0 = wave * loads_per_wave;
  const int64_t q1 = (q0 + loads_per_wave < n_loads) ? q0 + loads_per_wave : n_loads;
  if (q0 >= q1) { return; }
  int low = 0;                                   // index bits of the values this lane ends with: its keep bits
#pragma unroll
  for (int l = 0; l < NSC; l++) { low |= (id_keep_bit(l, lane) ? 1 : 0) << l; }
  const bool writer = LG == NSC || ((lane & (GS - 1)) >> NSC) == 0;   // all-reduce levels leave copies: the first sub-group stores
  const int blk = lane / GS;
  const v4i_t *src = (const v4i_t *)((const int16_t *)p.x + (int64_t)obj * p.in_stride) + 64 * q0 + lane;
  const int64_t ybase = (int64_t)obj * p.out_stride;
  for (int64_t q = q0; q < q1; q += BATCH, src += 64 * BATCH) {
    const int rem = (int)(q1 - q < BATCH ? q1 - q : BATCH);
    v4i_t v[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; k++) { v[k] = __builtin_nontemporal_load(src + 64 * (k < rem ? k : rem - 1)); }
    int vals[V];
#pragma unroll
    for (int k = 0; k < BATCH; k++) {
      const unsigned d[4] = {(unsigned)v[k].x, (unsigned)v[k].y, (unsigned)v[k].z, (unsigned)v[k].w};
      int acc[CHN];
#pragma unroll
      for (int c = 0; c < CHN; c++) { acc[c] = 0; }
#pragma unroll
      for (int m = 0; m < 4; m++) {   // as in the per-load kernel: element pairs of one channel -> v_dot2 with (1, 1)
        unsigned w;
        if (CHN == 4) { w = __builtin_amdgcn_perm(d[2 + (m >> 1)], d[m >> 1], (m & 1) ? 0x07060302u : 0x05040100u); }
        else if (CHN == 2) { w = __builtin_amdgcn_perm(d[2 * (m >> 1) + 1], d[2 * (m >> 1)], (m & 1) ? 0x07060302u : 0x05040100u); }
        else { w = d[m]; }
        const int c = CHN == 4 ? m : (CHN == 2 ? (m & 1) : 0);
        if (SGN) { acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, w), (v2s){1, 1}, acc[c], false); }
        else { acc[c] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(v2us, w), (v2us){1, 1}, (unsigned)acc[c], false); }
      }
#pragma unroll
      for (int c = 0; c < CHN; c++) { vals[k * CHN + c] = acc[c]; }
    }
    id_reduce_levels<0, LG, V, V>(vals, lane);
#pragma unroll
    for (int j = 0; j < NREM; j++) {
      const int idx = (j << NSC) | low, c = idx % CHN, k = idx / CHN;
      if (writer && k < rem) {
        const int64_t a = (int64_t)(((uint64_t)((int64_t)((uint64_t)(int64_t)vals[j] << (cv.sh + cv.ka)) >> cv.ka)) & cv.am);   // wrap to ACC_TYPE
        int64_t qv = (int64_t)((uint64_t)((a + cv.rnd) >> cv.rs) << cv.ls2);
        qv = qv < cv.lo ? cv.lo : (qv > cv.hi ? cv.hi : qv);
        const int64_t o = (int64_t)(((uint64_t)((int64_t)((uint64_t)qv << cv.ko) >> cv.ko)) & cv.om);
        store_raw(p.y, ybase + ((q + k) * BPL + blk) * CHN + c, p.out_eb, o);
      }
    }
  }
}

template <int CHN, int GS>
static void launch_batch2(const IntgDumpParams &p, const IdConv &cv, int64_t lpw, int64_t n_loads, dim3 grid, hipStream_t s) {
  const int xm = (xcd_map_wanted(false) && ((int64_t)grid.x * grid.y) % 8 == 0) ? 1 : 0;
  if (p.in.S) { hipLaunchKernelGGL((intg_dump_batch_kernel<CHN, GS, true>), grid, dim3(256), 0, s, p, cv, lpw, n_loads, xm); }
  else { hipLaunchKernelGGL((intg_dump_batch_kernel<CHN, GS, false>), grid, dim3(256), 0, s, p, cv, lpw, n_loads, xm); }
}
template <int CHN>
static bool launch_batch(const IntgDumpParams &p, const IdConv &cv, int gs, int64_t lpw, int64_t n_loads, dim3 grid, hipStream_t s) {
  switch (gs) {
    case 8: launch_batch2<CHN, 8>(p, cv, lpw, n_loads, grid, s); return true;
    case 16: launch_batch2<CHN, 16>(p, cv, lpw, n_loads, grid, s); return true;
    case 32: launch_batch2<CHN, 32>(p, cv, lpw, n_loads, grid, s); return true;
    case 64: launch_batch2<CHN, 64>(p, cv, lpw, n_loads, grid, s); return true;
    default: return false;
  }
}

// ACC -> OUT as shifts, a clamp and bit-field wraps where the modes allow it
static bool make_conv(const IntgDumpParams &p, IdConv &cv) {
  memset(&cv, 0, sizeof cv);
  cv.sh = p.acc.F - p.in.F;
  const int rs = p.acc.F - p.out.F;
  cv.ka = 64 - p.acc.W; cv.am = p.acc.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.acc.W));
  cv.rs = rs > 0 ? rs : 0; cv.ls2 = rs < 0 ? -rs : 0;
  cv.rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (int64_t(1) << (rs - 1)) : 0;
  if (p.out.O == ACDSP_SAT) { cv.lo = p.out.lo; cv.hi = p.out.hi; cv.ko = 0; cv.om = ~uint64_t(0); }
  else { cv.lo = INT64_MIN; cv.hi = INT64_MAX; cv.ko = 64 - p.out.W; cv.om = p.out.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.out.W)); }
  // 64-bit arithmetic of the conversion: the rounding add and (for AC_SAT) the left shift must not leave int64; an unsigned ACC_TYPE of 63 / 64
  // bits is a negative int64 here, which only a pure bit copy (no right shift, AC_WRAP) survives; AC_SAT bounds of an unsigned OUT_TYPE must fit
  // int64.  Round 5: the header's usage example (ac_intg_dump.h:47-51: <32,16> into ACC = OUT <64,32>) ran the tiled kernel at 0.30 of the
  // roofline because this test asked for W_acc <= 61 whatever the shifts.
  const bool wrap_o = p.out.O == ACDSP_WRAP;
  cv.ok = (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) && (wrap_o || p.out.O == ACDSP_SAT) && cv.rs <= 60 &&
          (cv.rnd == 0 || p.acc.W <= 61) && (cv.ls2 == 0 || p.acc.W + cv.ls2 <= 61 || wrap_o) &&
          (p.acc.S || p.acc.W <= 62 || (rs == 0 && wrap_o)) && p.acc.W <= 64 &&
          (wrap_o || p.out.S ? p.out.W <= 64 : p.out.W <= 62) &&
          cv.sh >= 0 && cv.sh + cv.ka < 64;                       // the IN -> ACC cast `<< (sh + ka)` stays a defined shift (else: tiled kernel)
  return cv.ok != 0;
}

template <typename TIN, int CHN>
static bool launch_stream(const IntgDumpParams &p, int gs, int lpr, int64_t rpw, int64_t n_reds, dim3 grid, hipStream_t s, int rag = 0) {
  IdConv cv;
  if (!make_conv(p, cv)) { return false; }                        // other modes: the tiled kernel and its general requant
  if constexpr (CHN == 2 || CHN == 4) {
    if (gs >= CHN) {
      if (p.in.S) { hipLaunchKernelGGL((intg_dump_stream_kernel<TIN, CHN, true, true>), grid, dim3(256), 0, s, p, cv, gs, lpr, rpw, n_reds, rag); }
      else { hipLaunchKernelGGL((intg_dump_stream_kernel<TIN, CHN, false, true>), grid, dim3(256), 0, s, p, cv, gs, lpr, rpw, n_reds, rag); }
      return true;
    }
  }
  if (p.in.S) { hipLaunchKernelGGL((intg_dump_stream_kernel<TIN, CHN, true, false>), grid, dim3(256), 0, s, p, cv, gs, lpr, rpw, n_reds, rag); }
  else { hipLaunchKernelGGL((intg_dump_stream_kernel<TIN, CHN, false, false>), grid, dim3(256), 0, s, p, cv, gs, lpr, rpw, n_reds, rag); }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Channel counts that do not divide a 16-byte load (round 6: CHN = 3, 5, 6, 7, 9 ... 16 -- ac_intg_dump.h:127-151 loops any CHN; the LDS-tiled
// kernel ran them at 0.30 of the roofline).  The column sums of a block, out[c] = sum_r x[r CHN + c], are a matrix product with a 0 / 1 selection
// matrix, and the matrix cores take it at no cost beside the stream (v_mfma_i32_16x16x64_i8):
//   column n  = block blk0 + n of the tile (16 blocks per wave tile),   K = 64 consecutive samples of that block (segment j),
//   A[c][k]   = 1 where sample 64 j + k of a block belongs to channel c, i.e. (64 j + k) mod CHN == c (a block starts on a channel-0 sample),
//               and 64 j + k lies inside the block;  the pattern repeats with period P = CHN / gcd(64, CHN) segments: P fragments in registers,
//   B         = the samples' byte planes (the low planes re-biased to signed bytes, 128 x rounds added back at the end),
//   D[c][n]  += A_j B_j over the segments of the block: lane (n, kg) ends up with channels 4 kg .. 4 kg + 3 of block n in int32 per plane.
// Every lane loads 16 consecutive samples (32 / 64 bytes) per segment: the wave reads 16 x 128 (256) contiguous bytes, each byte once.
// Conditions (launch_intg_dump): wrapping / sat-free accumulator, every block dumps the same number of rounds, nothing carried in, a block is a
// multiple of 16 samples, 16-byte aligned rows, CHN <= 16, fewer than 2^23 rounds (int32 plane sums).
template <typename TIN, int CHN>
__global__ void __launch_bounds__(64) intg_dump_mfma_kernel(IntgDumpParams p, int tiles_per_wave, int n_tiles) {
  constexpr int S = (int)sizeof(TIN), PX = S;
  constexpr int G64 = (CHN % 16 == 0) ? 16 : ((CHN % 8 == 0) ? 8 : ((CHN % 4 == 0) ? 4 : ((CHN % 2 == 0) ? 2 : 1)));   // gcd(64, CHN), CHN <= 16
  constexpr int P = CHN / G64;
  const int lane = threadIdx.x, n = lane & 15, kg = lane >> 4, obj = blockIdx.y;
  const int64_t B = p.uni_rounds * CHN;                 // samples per block (a multiple of 16)
  const int nseg = (int)((B + 63) / 64), rem = (int)(B - 64 * (int64_t)(nseg - 1));
  // selection fragments: lane (row c = lane & 15, kg) holds A[c][16 kg .. 16 kg + 15] of phase ph = (64 j) mod CHN, j mod P = q
  v4i_t A[P];
#pragma unroll
  for (int q = 0; q < P; q++) {
    const int ph = (64 * q) % CHN;
    unsigned w[4];
#pragma unroll
    for (int dw = 0; dw < 4; dw++) {
      w[dw] = 0;
#pragma unroll
      for (int bj = 0; bj < 4; bj++) {
        const int k = 16 * kg + 4 * dw + bj;
        w[dw] |= (unsigned)(((k + ph) % CHN == n && n < CHN) ? 1u : 0u) << (8 * bj);
      }
    }
    A[q] = (v4i_t){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  }
  const v4i_t zero4 = {0, 0, 0, 0};
  const TIN *row = (const TIN *)p.x + (int64_t)obj * p.in_stride;
  const int sh = p.acc.F - p.in.F;
  const int nbias = p.in.S ? PX - 1 : PX;               // planes that hold unsigned bytes
  const int t0 = blockIdx.x * tiles_per_wave, t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
  auto fetch = [&](int64_t e, v4i_t (&d)[S]) {
#pragma unroll
    for (int q = 0; q < S; q++) { d[q] = ((const v4i_t *)(row + e))[q]; }
  };
  for (int t = t0; t < t1; t++) {
    const int b = 16 * t + n, bc = b < p.n_blocks ? b : p.n_blocks - 1;
    const int64_t eb = (int64_t)bc * B;
    v4i_t acc[PX];
#pragma unroll
    for (int pp = 0; pp < PX; pp++) { acc[pp] = zero4; }
    v4i_t cur[S], nxt[S];
    fetch(eb + (16 * kg < B ? 16 * kg : 0), cur);
    auto segment = [&](int j, v4i_t a) __attribute__((always_inline)) {
      const int64_t on = 64 * (int64_t)(j + 1) + 16 * kg;
      fetch(eb + (on < B ? on : 0), nxt);               // beyond the block: any sample of it (the fragment is zero there)
      // byte planes of the lane's 16 samples (fir_gen.hip: stage_slot)
      union { v4i_t v[S]; unsigned d[4 * S]; } u;
#pragma unroll
      for (int q = 0; q < S; q++) { u.v[q] = cur[q]; }
      if (j == nseg - 1 && 16 * kg >= rem) { a = zero4; }
#pragma unroll
      for (int pp = 0; pp < PX; pp++) {
        v4i_t o;
        if constexpr (S == 2) {
          const unsigned sel = pp == 0 ? 0x06040200u : 0x07050301u;
          o.x = (int)__builtin_amdgcn_perm(u.d[1], u.d[0], sel); o.y = (int)__builtin_amdgcn_perm(u.d[3], u.d[2], sel);
          o.z = (int)__builtin_amdgcn_perm(u.d[5], u.d[4], sel); o.w = (int)__builtin_amdgcn_perm(u.d[7], u.d[6], sel);
        } else {
          auto g4 = [](unsigned d0, unsigned d1, unsigned d2, unsigned d3, int bp) {
            const unsigned sel = 0x0c0c0400u + 0x0101u * (unsigned)bp;
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(d3, d2, sel), __builtin_amdgcn_perm(d1, d0, sel), 0x05040100u);
          };
          o.x = (int)g4(u.d[0], u.d[1], u.d[2], u.d[3], pp); o.y = (int)g4(u.d[4], u.d[5], u.d[6], u.d[7], pp);
          o.z = (int)g4(u.d[8], u.d[9], u.d[10], u.d[11], pp); o.w = (int)g4(u.d[12], u.d[13], u.d[14], u.d[15], pp);
        }
        if (pp < nbias) { o ^= (v4i_t){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u}; }
        acc[pp] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, o, acc[pp], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < S; q++) { cur[q] = nxt[q]; }
    };
    for (int j0 = 0; j0 < nseg; j0 += P) {              // the fragment of segment j is A[j mod P]: P segments per trip, static register choice
#pragma unroll
      for (int q = 0; q < P; q++) {
        if (j0 + q < nseg) { segment(j0 + q, A[q]); }
      }
    }
    // lane (n, kg) holds channels 4 kg + r of block b: recombine the planes, ACC_TYPE wrap, OUT_TYPE conversion
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = 4 * kg + r;
      if (c < CHN && b < p.n_blocks) {
        uint64_t sum = 0;
#pragma unroll
        for (int pp = 0; pp < PX; pp++) {
          const int64_t v = (int64_t)acc[pp][r] + (pp < nbias ? 128 * p.uni_rounds : 0);
          sum += (uint64_t)v << (8 * pp);
        }
        const int64_t accv = wrap64((int64_t)(sum << sh), p.acc.W, p.acc.S);
        store_raw(p.y, (int64_t)obj * p.out_stride + (int64_t)b * CHN + c, p.out_eb, requant64(accv, p.acc.F, p.out));
      }
    }
  }
}

template <typename TIN>
static bool launch_mfma_t(const IntgDumpParams &p, int tpw, int n_tiles, dim3 grid, hipStream_t s) {
#define ACDSP_ID_CASE(C) case C: hipLaunchKernelGGL((intg_dump_mfma_kernel<TIN, C>), grid, dim3(64), 0, s, p, tpw, n_tiles); return true;
  switch (p.chn) {
    ACDSP_ID_CASE(1) ACDSP_ID_CASE(2) ACDSP_ID_CASE(3) ACDSP_ID_CASE(4) ACDSP_ID_CASE(5) ACDSP_ID_CASE(6) ACDSP_ID_CASE(7) ACDSP_ID_CASE(8)
    ACDSP_ID_CASE(9) ACDSP_ID_CASE(10) ACDSP_ID_CASE(11) ACDSP_ID_CASE(12) ACDSP_ID_CASE(13) ACDSP_ID_CASE(14) ACDSP_ID_CASE(15) ACDSP_ID_CASE(16)
    default: return false;
  }
#undef ACDSP_ID_CASE
}

// true: launched
static bool try_mfma(const IntgDumpParams &p, hipStream_t s) {
  static const bool off = getenv("ACDSP_NO_INTG_MFMA") != nullptr;   // A/B knob: the LDS-tiled kernel
  if (off || !p.tile_ok || p.uni_rounds <= 0 || p.uni_rounds >= (1 << 23) || (p.in_eb != 2 && p.in_eb != 4) || p.chn < 1 || p.chn > 16) { return false; }
  const int64_t B = p.uni_rounds * p.chn;
  if (B % 16 != 0 || ((uintptr_t)p.x % 16) != 0 || (p.in_stride * p.in_eb) % 16 != 0 || p.acc.F < p.in.F || p.acc.F - p.in.F >= 32) { return false; }
  const int n_tiles = (p.n_blocks + 15) / 16;
  // ~32 KB per wave, at least ~16 K waves when the problem allows it
  int64_t tpw = (32768 + 16 * B * p.in_eb - 1) / (16 * B * p.in_eb);
  if (tpw < 1) { tpw = 1; }
  while (tpw > 1 && ((n_tiles + tpw - 1) / tpw) * (int64_t)p.n_obj < 16384) { tpw /= 2; }
  dim3 grid((unsigned)((n_tiles + tpw - 1) / tpw), (unsigned)p.n_obj);
  return p.in_eb == 2 ? launch_mfma_t<int16_t>(p, (int)tpw, n_tiles, grid, s) : launch_mfma_t<int32_t>(p, (int)tpw, n_tiles, grid, s);
}

// true: launched.  Shape conditions of the streaming kernel (see above); rows must start on 16-byte boundaries.
static bool try_stream(const IntgDumpParams &p, hipStream_t s) {
  if (!p.tile_ok || p.uni_rounds <= 0 || p.uni_rounds >= 32768 || (p.in_eb != 2 && p.in_eb != 4)) { return false; }
  const int epv = 16 / p.in_eb;
  if (p.chn < 1 || p.chn > epv || epv % p.chn != 0) { return false; }
  const int64_t be = p.uni_rounds * p.chn;                     // elements per block
  if (be % epv != 0 || ((uintptr_t)p.x % 16) != 0 || (p.in_stride * p.in_eb) % 16 != 0) { return false; }
  const int64_t lpb = be / epv;                                // lane-loads per block
  int gs, lpr;
  int rag = 0;   // blocks that are no multiple of a wave-load: one block per reduce, masked last load (at least half a wave-load per block)
  if (lpb <= 64) {
    if (lpb & (lpb - 1)) { if (lpb < 32) { return false; } gs = 64; lpr = 1; rag = (int)lpb; }
    else { gs = (int)lpb; lpr = 1; }
  } else if (lpb % 64) {
    if (lpb > (1 << 20)) { return false; }
    gs = 64; lpr = (int)((lpb + 63) / 64); rag = (int)lpb;
  } else { gs = 64; lpr = (int)(lpb / 64); }
  const int bpr = 64 / gs;
  if (p.n_blocks % bpr != 0) { return false; }                 // (a ragged last reduce would read past the call's samples)
  const int64_t n_reds = p.n_blocks / bpr;
  static const bool no_batch = getenv("ACDSP_NO_INTG_BATCH") != nullptr;   // A/B knob: per-load kernel
  if (!no_batch && !rag && p.in_eb == 2 && lpr == 1 && gs >= 8 && (p.chn == 1 || p.chn == 2 || p.chn == 4)) {
    IdConv cv;
    if (!make_conv(p, cv)) { return false; }
    // 8 KB per wave (one 8-load batch; 8: 0.361 ms, 16: 0.373, 32: 0.375 on the bench row, profiles/r3_span_sweep.txt)
    int64_t lpw = 8;
    while (lpw > 8 && (n_reds / lpw) * p.n_obj < 16384) { lpw /= 2; }
    ACDSP_TUNE_ENV(lpw_env, "ACDSP_INTG_RPW");
    if (lpw_env && atoi(lpw_env) > 0) { lpw = (atoi(lpw_env) + 7) / 8 * 8; }
    const int64_t waves_b = (n_reds + lpw - 1) / lpw;
    dim3 grid_b((unsigned)((waves_b + 3) / 4), (unsigned)p.n_obj);
    switch (p.chn) {
      case 1: return launch_batch<1>(p, cv, gs, lpw, n_reds, grid_b, s);
      case 2: return launch_batch<2>(p, cv, gs, lpw, n_reds, grid_b, s);
      default: return launch_batch<4>(p, cv, gs, lpw, n_reds, grid_b, s);
    }
  }
  // ~32 KB per wave (8 .. 32 KB measured alike, 64 KB 4 % slower), >= ~16 K waves when the problem allows it
  int64_t rpw = (32 + lpr - 1) / lpr;
  while (rpw > 1 && (n_reds / rpw) * p.n_obj < 16384) { rpw /= 2; }
  ACDSP_TUNE_ENV(rpw_env, "ACDSP_INTG_RPW");   // tuning knob: reduces per wave
  if (rpw_env && atoi(rpw_env) > 0) { rpw = atoi(rpw_env); }
  const int64_t waves = (n_reds + rpw - 1) / rpw;
  dim3 grid((unsigned)((waves + 3) / 4), (unsigned)p.n_obj);
  if (p.in_eb == 2) {
    switch (p.chn) {
      case 1: return launch_stream<int16_t, 1>(p, gs, lpr, rpw, n_reds, grid, s, rag);
      case 2: return launch_stream<int16_t, 2>(p, gs, lpr, rpw, n_reds, grid, s, rag);
      case 4: return launch_stream<int16_t, 4>(p, gs, lpr, rpw, n_reds, grid, s, rag);
      default: return launch_stream<int16_t, 8>(p, gs, lpr, rpw, n_reds, grid, s, rag);
    }
  } else {
    switch (p.chn) {
      case 1: return launch_stream<int32_t, 1>(p, gs, lpr, rpw, n_reds, grid, s, rag);
      case 2: return launch_stream<int32_t, 2>(p, gs, lpr, rpw, n_reds, grid, s, rag);
      default: return launch_stream<int32_t, 4>(p, gs, lpr, rpw, n_reds, grid, s, rag);
    }
  }
}

hipError_t launch_intg_dump(const IntgDumpParams &p, int64_t *temp_next, hipStream_t s, bool *temp_written, int *path) {
  *temp_written = true;
  *path = 0;
  static const bool no_stream = getenv("ACDSP_NO_INTG_STREAM") != nullptr;   // A/B knob
  if (!no_stream && try_stream(p, s)) {
    *path = 2;
    *temp_written = false;   // every block dumped and nothing was carried in (tile_ok): temp[] was zero and is zero
    return hipGetLastError();
  }
  if (try_mfma(p, s)) {
    *path = 3;
    *temp_written = false;
    return hipGetLastError();
  }
  if (p.tile_ok && p.chn <= 256) {
    const int bpw = 256 / p.chn;
    const int lds_elems = (47 * 1024) / p.in_eb;   // + one pad dword per 512 bytes stays inside the 48 KB
    dim3 grid((unsigned)((p.n_blocks + bpw - 1) / bpw), (unsigned)p.n_obj);
    switch (p.in_eb) {
      case 2: hipLaunchKernelGGL(intg_dump_tile_kernel<int16_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
      case 4: hipLaunchKernelGGL(intg_dump_tile_kernel<int32_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
      default: hipLaunchKernelGGL(intg_dump_tile_kernel<int64_t>, grid, dim3(256), 48 * 1024, s, p, bpw, lds_elems); break;
    }
    *temp_written = false;
    *path = 1;
    return hipGetLastError();
  }
  const int64_t n_work = (int64_t)(p.n_blocks + 1) * p.chn;
  dim3 grid((unsigned)((n_work + 255) / 256), (unsigned)p.n_obj);
  hipLaunchKernelGGL(intg_dump_kernel, grid, dim3(256), 0, s, p, temp_next);
  return hipGetLastError();
}

}  // namespace acdsp
