// cic.hip -- many-channel CIC decimator / interpolator kernels for gfx950.
//
// What they replace: ac_cic_full_core_intg::intStage / decIntgCore / intrIntgCore
// (reference include/ac_dsp/ac_cic_full_core.h:80-160) and
// ac_cic_full_core_diff::comb / diffStage (:228-255), driven by
// ac_cic_dec_full::run (ac_cic_dec_full.h:163-222) and ac_cic_intr_full::run
// (ac_cic_intr_full.h:150-215).
//
// Arithmetic.  The reference computes in INT_TYPE = ac_fixed<outW,..,true>
// with the default AC_TRN/AC_WRAP, i.e. every add/sub is exact modulo 2^outW.
// Reduction mod 2^outW is a ring homomorphism, so the kernels add/subtract in
// uint64 (outW <= 64) and wrap to outW bits once, just before the OUT_TYPE
// conversion.  The integrators are the reference's *pipelined* form (stage i
// adds the previous value of stage i-1, :82-85).  The comb delay line keeps
// the reference's behaviour exactly: its shift loop runs ascending (:249-254),
// which makes the differential delay min(M, 2) -- see `me`.
//
// Parallelisation.  One lane = one channel (the recurrences are serial in
// time, independent across channels); one wave = 64 channels x one time chunk.
// Inputs are [channel][time]: a tile of 64 channels x 64 samples is fetched
// with 16-byte coalesced row loads, transposed through a padded LDS tile
// (conflict-free ds_write_b128 / ds_read_b128), and each lane then walks its
// own row.  A chunk does not need the integrator state of the chunk before it:
// the decimator (and interpolator) is an FIR system overall,
//   H(z) = z^-(N-1) * (1 + z^-1 + ... + z^-(R*me-1))^N      (dec, at the input rate),
// so simulating from zero state `warm_tiles` tiles earlier reproduces every
// output of the chunk exactly (mod 2^outW).  The same argument replaces the
// reference's carried registers by a short input history between run() calls.
#include <type_traits>

#include <cstdlib>
#include "cic_kernels.hpp"

namespace acdsp {

typedef int v4i __attribute__((ext_vector_type(4)));

template <typename TIN> struct Vec16 {
  union { v4i v; TIN e[16 / sizeof(TIN)]; };
};

template <int N, typename TIN, bool INTERP, int ME>
__global__ void __launch_bounds__(64) cic_kernel(CicParams p) {
  constexpr int VE = 16 / (int)sizeof(TIN);             // elements per 16-byte vector
  constexpr int LPR = kCicTile / VE;                    // lanes that cover one tile row = load instructions per tile
  constexpr int RPI = 64 / LPR;                         // rows fetched per load instruction
  // LDS tile [64 rows][64 samples] with a row pitch of 64*sizeof(TIN)+4 bytes (== 1 dword mod 32): the
  // per-sample column read of lane c (row c) is bank-conflict free for every container width
  constexpr int ROWB = kCicTile * (int)sizeof(TIN) + 4;
#ifndef ACDSP_CIC_OB
#define ACDSP_CIC_OB 16
#endif
  constexpr int OB = ACDSP_CIC_OB;                      // decimator: outputs staged per row before a flush (128-byte row segments; 8 and 32 measured slower -- again in round 5 at R = 32 / 64, where the smaller tile buys occupancy: 16 -> 8 -> 4: 3.57 / 3.94 / 4.09 ms and 2.38 / 2.52 / 2.69 ms)
  constexpr int OPITCH = OB + 1;                        // int64 words per staged row (conflict-free ds_write_b64)
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * ROWB];
  __shared__ int64_t obuf[INTERP ? 1 : 64 * OPITCH];

  const int lane = threadIdx.x;
  const int ch0 = blockIdx.y * 64;
  const int ch = ch0 + lane;
  const bool ch_ok = ch < p.n_ch;
  const int64_t c_start = p.t_from + (int64_t)blockIdx.x * p.chunk;
  const int64_t c_end = (c_start + p.chunk < p.n_in) ? c_start + p.chunk : p.n_in;
  const int64_t e_start = (!INTERP && p.emit_from > c_start) ? p.emit_from : c_start;   // first sample whose emission is stored
  const int64_t s0 = c_start - (int64_t)p.warm_tiles * kCicTile;  // >= -hl
  const int R = p.R;

  uint64_t r[N], d0[N], d1[ME == 2 ? N : 1];
#pragma unroll
  for (int i = 0; i < N; i++) { r[i] = 0; d0[i] = 0; }
#pragma unroll
  for (int i = 0; i < (ME == 2 ? N : 1); i++) { d1[i] = 0; }

  // decimator bookkeeping
  int cnt = 0;
  int64_t j = 0;       // output index of the next emission
  int ocnt = 0;        // outputs staged since the last flush (same for every lane)
  int64_t jbase = 0;   // output index of staged slot 0
  if (!INTERP) {
    int64_t m = ((int64_t)p.phase0 + s0) % R;
    cnt = (int)(m < 0 ? m + R : m);
    int64_t te0 = s0 + ((R - cnt) % R);
    j = (te0 - p.first) / R;  // exact division
  }
  // interpolator emission window of this chunk, in global iteration numbers
  int64_t qa = 0, qb = 0, q_base = 0;
  if (INTERP) {
    q_base = p.q_begin > p.q_skip ? p.q_begin : p.q_skip;
    qa = (blockIdx.x == 0) ? p.q_begin : (p.t_prev + c_start) * R;
    if (qa < q_base) { qa = q_base; }
    qb = (p.t_prev + c_end) * R;
    if (qb > p.q_end) { qb = p.q_end; }
  }

  // Row-coalesced write-out of the staged outputs: 64 consecutive lanes cover 64/OB rows x OB outputs.
  auto flush = [&](int n_valid) {
    __syncthreads();
    for (int idx = lane; idx < 64 * OB; idx += 64) {
      const int row = idx / OB, col = idx % OB;
      if (col < n_valid && ch0 + row < p.n_ch) {
        store_raw(p.y, (int64_t)(ch0 + row) * p.out_stride + jbase + col, p.out_eb, obuf[row * OPITCH + col]);
      }
    }
    __syncthreads();
  };

  // The tile of step ts is fetched one tile ahead into registers (16-byte coalesced row loads), so
  // its HBM latency overlaps the serial walk through the previous tile.
  Vec16<TIN> pre[LPR];
  auto fetch = [&](int64_t ts) {
#pragma unroll
    for (int li = 0; li < LPR; li++) {
      const int row = li * RPI + lane / LPR;
      const int col = (lane % LPR) * VE;
      int chr = ch0 + row;
      if (chr >= p.n_ch) { chr = p.n_ch - 1; }
      const int64_t t = ts + col;
      if (ts < 0) {
        pre[li].v = *(const v4i *)((const TIN *)p.hist + (int64_t)chr * p.hl + (p.hl + t));
      } else {
        const TIN *src = (const TIN *)p.x + (int64_t)chr * p.in_stride + t;
        if (p.vec_ok && t + VE <= p.n_in) {
          pre[li].v = *(const v4i *)src;
        } else {
#pragma unroll
          for (int e = 0; e < VE; e++) { pre[li].e[e] = (t + e < p.n_in) ? src[e] : (TIN)0; }
        }
      }
    }
  };

  // (OUT_TYPE) data_in : lossless cast of one staged sample into INT_TYPE, ac_cic_full_core.h:114,147,213
  auto load_x = [&](int k) -> uint64_t {
    const TIN raw = *(const TIN *)(lds + lane * ROWB + k * (int)sizeof(TIN));
    if (sizeof(TIN) == 8) { return (uint64_t)raw; }
    if (p.in.S) { return (uint64_t)(int64_t)raw; }
    if (sizeof(TIN) == 4) { return (uint64_t)(uint32_t)raw; }
    return (uint64_t)(uint16_t)raw;
  };
  auto out_word = [&](uint64_t val) -> int64_t {
    if (p.out_simple == 2) { return wrap64((int64_t)val, p.w_int, 1); }                               // OUT holds INT_TYPE
    if (p.out_simple) { return wrap64(wrap64((int64_t)val, p.w_int, 1), p.out.W, p.out.S); }         // same F, AC_WRAP
    return requant64(wrap64((int64_t)val, p.w_int, 1), p.in.F, p.out);
  };
  // comb / diffStage, ac_cic_full_core.h:228-255, differential delay ME = min(M, 2)
  auto comb = [&](uint64_t val) -> uint64_t {
#pragma unroll
    for (int st = 0; st < N; st++) {
      uint64_t o;
      if (ME == 2) { o = val - d1[st]; d1[st] = d0[st]; }
      else { o = val - d0[st]; }
      d0[st] = val;
      val = o;
    }
    return val;
  };

  fetch(s0);
  for (int64_t ts = s0; ts < c_end; ts += kCicTile) {
    // ---- stage the tile: 64 channel rows x 64 samples (transposition through padded LDS) ----
#pragma unroll
    for (int li = 0; li < LPR; li++) {
      const int row = li * RPI + lane / LPR;
      const int col = (lane % LPR) * VE;
      int *dst = (int *)(lds + row * ROWB + col * (int)sizeof(TIN));
      dst[0] = pre[li].v.x; dst[1] = pre[li].v.y; dst[2] = pre[li].v.z; dst[3] = pre[li].v.w;
    }
    __syncthreads();
    if (ts + kCicTile < c_end) { fetch(ts + kCicTile); }

    // ---- every lane walks its own channel row ----
    // samples [0, nv) of the tile exist; emissions are stored from sample index e0 on (warm-up before)
    const int nv = (c_end - ts < kCicTile) ? (int)(c_end - ts) : kCicTile;
    const int e0 = (e_start > ts) ? ((e_start - ts < kCicTile) ? (int)(e_start - ts) : kCicTile) : 0;
    if (!INTERP) {
      // The walk is split into runs that end at an emitting sample, so the integrator loop body is
      // branch-free: intStage (ac_cic_full_core.h:80-87) on every sample, comb + output only at run ends.
      int k = 0;
      while (k < nv) {
        const int to_emit = (cnt == 0) ? 1 : R - cnt + 1;      // samples up to and including the next emitting one
        const int run = (to_emit < nv - k) ? to_emit : nv - k;
        int i = 0;
        for (; i + 8 <= run; i += 8) {       // 8 LDS reads in flight, then 8 branch-free integrator steps
          uint64_t xs[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { xs[u] = load_x(k + i + u); }
#pragma unroll
          for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int q = N - 1; q > 0; q--) { r[q] += r[q - 1]; }
            r[0] += xs[u];
          }
        }
        for (; i < run; i++) {
          const uint64_t x = load_x(k + i);
#pragma unroll
          for (int q = N - 1; q > 0; q--) { r[q] += r[q - 1]; }
          r[0] += x;
        }
        k += run;
        if (run == to_emit) {                                   // valid = (rate_cnt == 0), :116-120
          const uint64_t val = comb(r[N - 1]);
          if (k - 1 >= e0) {
            if (ocnt == 0) { jbase = j; }
            obuf[lane * OPITCH + ocnt] = out_word(val);
            if (++ocnt == OB) { flush(OB); ocnt = 0; }
          }
          j++;
          cnt = (R == 1) ? 0 : 1;                               // :130-133
        } else {
          cnt += run;
        }
      }
    } else {
      for (int k = 0; k < nv; k++) {
        // intrDiffCore, :211-216
        const uint64_t val = comb(load_x(k));
        // intrIntgCore, :143-160: the sample, then R-1 stuffed zeros
        const int64_t qbase = (p.t_prev + ts + k) * R;
        for (int ph = 0; ph < R; ph++) {
#pragma unroll
          for (int i = N - 1; i > 0; i--) { r[i] += r[i - 1]; }
          r[0] += (ph == 0) ? val : (uint64_t)0;
          const int64_t q = qbase + ph;
          if (q >= qa && q < qb && ch_ok) {
            store_raw(p.y, (int64_t)ch * p.out_stride + (q - q_base), p.out_eb, out_word(r[N - 1]));
          }
        }
      }
    }
    __syncthreads();
  }
  if (!INTERP && ocnt > 0) { flush(ocnt); }
}

template <int N, typename TIN>
static hipError_t launch_n_t(const CicParams &p, dim3 grid, hipStream_t s) {
  if (p.interp) {
    if (p.me == 2) { hipLaunchKernelGGL((cic_kernel<N, TIN, true, 2>), grid, dim3(64), 0, s, p); }
    else { hipLaunchKernelGGL((cic_kernel<N, TIN, true, 1>), grid, dim3(64), 0, s, p); }
  } else {
    if (p.me == 2) { hipLaunchKernelGGL((cic_kernel<N, TIN, false, 2>), grid, dim3(64), 0, s, p); }
    else { hipLaunchKernelGGL((cic_kernel<N, TIN, false, 1>), grid, dim3(64), 0, s, p); }
  }
  return hipGetLastError();
}

template <int N>
static hipError_t launch_n(const CicParams &p, dim3 grid, hipStream_t s) {
  switch (p.in_eb) {
    case 2: return launch_n_t<N, int16_t>(p, grid, s);
    case 4: return launch_n_t<N, int32_t>(p, grid, s);
    default: return launch_n_t<N, int64_t>(p, grid, s);
  }
}

// ---------------------------------------------------------------------------------------------------
// Interpolator through its FIR identity.  Comb chain at the low rate, zero stuffing and N pipelined integrators
// (reference ac_cic_full_core.h:80-87, 143-160, 211-255) are, end to end,
//     out[q] = sum_n h[q - R n] x[n]  (mod 2^W_int),   h = z^-(N-1) (1 + ... + z^-(R me - 1))^N,
// q = global iteration (the sample x[n] enters the integrators at q = R n), i.e. output phase r = q mod R is an
// (N me)-tap FIR on the input: a handful of MACs per output instead of N serial wide adds per output on one lane.
// One thread per output, consecutive threads = consecutive q: the 8-byte stores of a wave are one contiguous 512-byte
// run (the recurrence kernel wrote 128-byte row segments from 64 different rows: 0.7 TB/s).
constexpr int kIntrTile = 4096;   // outputs per workgroup (one block per 256 outputs was bound by the workgroup dispatch rate)

template <typename TIN>
__global__ void __launch_bounds__(256) cic_intr_fir_kernel(CicParams p, const int64_t *__restrict__ taps, int n_taps, uint32_t rcp,
                                                           int kmax) {
  // LDS: the taps as int32 (every tap < 2^31: (R M)^N < 2^31 is a limit of the reference itself) and the input window
  // of this block's 256 outputs, so that a MAC is two LDS reads (broadcast / conflict-free) and one v_mad_i64_i32.
  typedef typename std::conditional<(sizeof(TIN) > 4), int64_t, int32_t>::type XW;
  extern __shared__ __attribute__((aligned(8))) unsigned char lds_raw[];
  int32_t *lt = (int32_t *)lds_raw;
  XW *xw = (XW *)(lds_raw + (size_t)((n_taps + 1) / 2 * 2) * 4);
  const int ch = blockIdx.y;
  const int64_t lo = p.q_begin > p.q_skip ? p.q_begin : p.q_skip;    // first emitted iteration of this call
  const int64_t q_stop = p.q_to > p.q_from ? p.q_to : p.q_end;       // this launch: [q_from, q_to) (or the whole call)
  const int64_t q0 = (p.q_to > p.q_from ? p.q_from : lo) + (int64_t)blockIdx.x * kIntrTile;   // a block owns kIntrTile consecutive outputs
  const int R = p.R;
  const int64_t n_base = q0 / R;                                      // wave-uniform 64-bit division, once
  const int r0 = (int)(q0 - n_base * R);
  for (int i = threadIdx.x; i < n_taps; i += 256) { lt[i] = (int32_t)taps[i]; }
  // window: global inputs n_base - (kmax-1) .. n_base + (r0 + kIntrTile - 1) / R
  const int n_win = kmax + (r0 + kIntrTile - 1) / R;
  const TIN *xrow = (const TIN *)p.x + (int64_t)ch * p.in_stride;
  const TIN *hrow = (const TIN *)p.hist + (int64_t)ch * p.hl + p.hl;  // hrow[t], t < 0
  for (int j = threadIdx.x; j < n_win; j += 256) {
    const int64_t n = n_base - (kmax - 1) + j - p.t_prev;             // local input index
    int64_t x = 0;
    if (n >= 0) { if (n < p.n_in) { x = (int64_t)xrow[n]; } }
    else if (n >= -(int64_t)p.hl) { x = (int64_t)hrow[n]; }
    if (!p.in.S) { x &= (int64_t)((uint64_t)(-1) >> (64 - 8 * (int)sizeof(TIN))); }   // unsigned containers: zero-extend
    xw[j] = (XW)x;
  }
  __syncthreads();
#pragma unroll 2
  for (int i = 0; i < kIntrTile / 256; i++) {
    const int64_t q = q0 + threadIdx.x + 256 * i;                     // consecutive lanes = consecutive outputs
    if (q >= q_stop) { return; }
    const unsigned t = (unsigned)r0 + threadIdx.x + 256u * i;         // < R + kIntrTile < 2^16
    const unsigned dn = __umulhi(t, rcp);                             // t / R (exact for t < 2^16)
    const int r = (int)(t - dn * (unsigned)R);
    const XW *xp = xw + (kmax - 1) + dn;                              // xp[-k] = x[n_hi - k]
    int64_t acc = 0;
    for (int d = r, k = 0; d < n_taps; d += R, k++) {
      if (sizeof(TIN) > 4) { acc = (int64_t)((uint64_t)acc + (uint64_t)(int64_t)lt[d] * (uint64_t)xp[-k]); }
      else { acc += (int64_t)lt[d] * (int64_t)(int32_t)xp[-k]; }      // v_mad_i64_i32
    }
    int64_t o;
    if (p.out_simple == 2) { o = wrap64(acc, p.w_int, 1); }
    else if (p.out_simple == 1) { o = wrap64(wrap64(acc, p.w_int, 1), p.out.W, p.out.S); }
    else { o = requant64(wrap64(acc, p.w_int, 1), p.in.F, p.out); }
    store_raw(p.y, (int64_t)ch * p.out_stride + (q - lo), p.out_eb, o);
  }
}

hipError_t launch_cic_intr_fir(const CicParams &p, const int64_t *d_taps, int n_taps, hipStream_t s) {
  const int64_t lo = p.q_to > p.q_from ? p.q_from : (p.q_begin > p.q_skip ? p.q_begin : p.q_skip);
  const int64_t hi = p.q_to > p.q_from ? p.q_to : p.q_end;
  if (hi <= lo) { return hipSuccess; }
  dim3 grid((unsigned)((hi - lo + kIntrTile - 1) / kIntrTile), (unsigned)p.n_ch);
  const uint32_t rcp = (uint32_t)((0x100000000ull + p.R - 1) / p.R);
  const int kmax = (n_taps + p.R - 1) / p.R;
  const size_t lds = (size_t)((n_taps + 1) / 2 * 2) * 4 + (size_t)(kmax + (kIntrTile - 1) / p.R + 2) * 8;
  switch (p.in_eb) {
    case 2: hipLaunchKernelGGL(cic_intr_fir_kernel<int16_t>, grid, dim3(256), lds, s, p, d_taps, n_taps, rcp, kmax); break;
    case 4: hipLaunchKernelGGL(cic_intr_fir_kernel<int32_t>, grid, dim3(256), lds, s, p, d_taps, n_taps, rcp, kmax); break;
    default: hipLaunchKernelGGL(cic_intr_fir_kernel<int64_t>, grid, dim3(256), lds, s, p, d_taps, n_taps, rcp, kmax); break;
  }
  return hipGetLastError();
}

hipError_t launch_cic(const CicParams &p, hipStream_t s) {
  if (p.n_in <= 0) { return hipSuccess; }
  if (p.n_in <= p.t_from) { return hipSuccess; }
  dim3 grid((unsigned)((p.n_in - p.t_from + p.chunk - 1) / p.chunk), (unsigned)((p.n_ch + 63) / 64));
  switch (p.N) {
    case 1: return launch_n<1>(p, grid, s);
    case 2: return launch_n<2>(p, grid, s);
    case 3: return launch_n<3>(p, grid, s);
    case 4: return launch_n<4>(p, grid, s);
    case 5: return launch_n<5>(p, grid, s);
    case 6: return launch_n<6>(p, grid, s);
    case 7: return launch_n<7>(p, grid, s);
    case 8: return launch_n<8>(p, grid, s);
    default: return hipErrorInvalidValue;
  }
}

__global__ void cic_hist_update_kernel(CicParams p, void *hist_next) {
  const int ch = blockIdx.y;
  for (int jj = blockIdx.x * blockDim.x + threadIdx.x; jj < p.hl; jj += gridDim.x * blockDim.x) {
    int64_t g = p.n_in - p.hl + jj;
    int64_t v = (g >= 0) ? load_raw(p.x, (int64_t)ch * p.in_stride + g, p.in_eb, 1)
                         : load_raw(p.hist, (int64_t)ch * p.hl + p.hl + g, p.in_eb, 1);
    store_raw(hist_next, (int64_t)ch * p.hl + jj, p.in_eb, v);
  }
}

// The same for calls of at least hl samples: the tail of the call as 16-byte pieces (gfx950 serves vector loads at any element-aligned address;
// the two-stage decimator keeps 4 - 8 K samples per channel, and the element-wise copy of 36 MB took 52 us behind a 1.7 ms kernel: 15 us).
__global__ void cic_hist_update_vec_kernel(CicParams p, void *hist_next) {
  const int ch = blockIdx.y, per = 16 / p.in_eb;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j * per < p.hl) {
    const unsigned char *src = (const unsigned char *)p.x + ((int64_t)ch * p.in_stride + p.n_in - p.hl + j * per) * p.in_eb;
    unsigned char *dst = (unsigned char *)hist_next + ((int64_t)ch * p.hl + j * per) * p.in_eb;
    *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
  }
}

hipError_t launch_cic_hist_update(const CicParams &p, void *hist_next, hipStream_t s) {
  if (p.n_in <= 0) { return hipSuccess; }
  static const bool no_vec = getenv("ACDSP_NO_HIST_VEC") != nullptr;   // A/B knob
  if (!no_vec && p.n_in >= p.hl && p.hl >= 1024 && ((int64_t)p.hl * p.in_eb) % 16 == 0 && ((uintptr_t)hist_next % 16) == 0) {
    const int per = 16 / p.in_eb;
    dim3 grid((unsigned)((p.hl / per + 255) / 256), (unsigned)p.n_ch);
    hipLaunchKernelGGL(cic_hist_update_vec_kernel, grid, dim3(256), 0, s, p, hist_next);
    return hipGetLastError();
  }
  dim3 grid((unsigned)((p.hl + 255) / 256), (unsigned)p.n_ch);
  hipLaunchKernelGGL(cic_hist_update_kernel, grid, dim3(256), 0, s, p, hist_next);
  return hipGetLastError();
}

}  // namespace acdsp
