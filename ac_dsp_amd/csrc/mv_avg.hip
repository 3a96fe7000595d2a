// mv_avg.hip -- ac_mv_avg on many objects (second half).
//
// Replaces the per-sample loop of ac_mv_avg_core::mvAvgCore (reference include/ac_dsp/ac_mv_avg.h:111-123) driven by
// ac_mv_avg::run (:146-190): for every valid window position of a frame
//     acc = 0;  for j = -TAPS/2 .. TAPS/2:  acc = ACC_TYPE(acc + ACC_TYPE(w[j]) * coeffs[j + TAPS/2]);  out = OUT_TYPE(acc)
// with w the ac_window_1d_flag over the frame (AC_WIN: interior positions only; AC_CLIP / AC_MIRROR: every position,
// samples outside the frame replaced by the edge sample / the sample mirrored about it -- include/ac_types/ac_window.h).
//
// Mapping: one thread per output, a 256-thread block per tile of one (object, frame); the block stages the
// 256 + TAPS - 1 window samples -- boundary rule applied, already cast to ACC_TYPE (the cast depends on the sample only) --
// and the coefficients in LDS with coalesced loads, then every thread walks its taps in the reference's order.
// Two arithmetic classes: per-tap requantisation in 128 bits (any Q / O), and, for AC_TRN / AC_RND + AC_WRAP accumulators
// whose products fit 63 bits, the order-free form acc = sum_j ((xq_j * c_j + rnd) >> F_c) mod 2^W_acc in int64.
// Streaming op: 2 + 2 bytes per sample at 16-bit containers; bound by HBM for short windows.
#include <cstdlib>
#include <cstring>

#include "fir_kernels.hpp"

namespace acdsp {

namespace {

constexpr int kTile = 256;

__device__ inline i128 shl128_(i128 v, int s) { return (i128)((u128)v << s); }

// line position of window element `pos` after the boundary rule (n samples per frame)
__device__ inline int64_t fold_pos(int64_t pos, int64_t n, int mode) {
  if (mode == 2) { return pos < 0 ? 0 : (pos > n - 1 ? n - 1 : pos); }
  if (n == 1) { return 0; }
  while (pos < 0 || pos > n - 1) {
    if (pos < 0) { pos = -pos; }
    if (pos > n - 1) { pos = 2 * (n - 1) - pos; }
  }
  return pos;
}

template <bool FAST>
__global__ void __launch_bounds__(kTile) mv_avg_kernel(MvAvgParams p) {
  extern __shared__ int64_t smem[];
  int64_t *win = smem;                       // [kTile + taps - 1]: ACC raw words of line positions m0 - h + j
  int64_t *cf = smem + kTile + p.taps - 1;   // [taps]
  const int tid = threadIdx.x, h = p.taps / 2;
  for (int i = tid; i < p.taps; i += kTile) { cf[i] = p.coeffs[i]; }
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  const int64_t first = p.win_mode == 0 ? h : 0;   // line position of a frame's first output
  for (int64_t pr = blockIdx.y; pr < pairs; pr += gridDim.y) {
    const int64_t obj = pr / p.n_frames, fr = pr % p.n_frames;
    const int64_t xbase = obj * p.in_stride + fr * p.n_sample, ybase = obj * p.out_stride + fr * p.out_per_frame;
    const int64_t m0 = first + (int64_t)blockIdx.x * kTile;   // line position of this tile's first output
    __syncthreads();   // previous pair's readers are done with win[]
    for (int j = tid; j < kTile + p.taps - 1; j += kTile) {
      int64_t pos = m0 - h + j;
      int64_t v = 0;
      if (p.win_mode != 0) { pos = fold_pos(pos, p.n_sample, p.win_mode); }
      if (pos >= 0 && pos < p.n_sample) { v = requant64(load_raw(p.x, xbase + pos, p.in_eb, p.in.S), p.in.F, p.acc); }   // (ACC_TYPE) w[j]
      win[j] = v;
    }
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * kTile + tid;   // output index within the frame
    if (k < p.out_per_frame) {
      const int64_t *w = win + tid;   // w[j + h] = element j of the window centred on this output
      int64_t acc = 0;
      if (FAST) {
        uint64_t s = 0;
        const int sh = p.cf.F;   // >= 0 in this class
        const int64_t rnd = (p.acc.Q == ACDSP_RND && sh > 0) ? (int64_t(1) << (sh - 1)) : 0;
        for (int j = 0; j < p.taps; j++) { s += (uint64_t)((w[j] * cf[j] + rnd) >> sh); }
        acc = wrap64((int64_t)s, p.acc.W, p.acc.S);
      } else {
        const int fp = p.acc.F + p.cf.F, f = fp > p.acc.F ? fp : p.acc.F;
        for (int j = 0; j < p.taps; j++) {
          const i128 sum = shl128_((i128)acc, f - p.acc.F) + shl128_((i128)w[j] * (i128)cf[j], f - fp);
          acc = requant128(sum, f, p.acc);
        }
      }
      store_raw(p.y, ybase + k, p.out_eb, requant64(acc, p.acc.F, p.out));
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Streaming kernel for the common class: 16-bit samples and 16-bit coefficients whose absolute values sum to < 2^15, a
// wrapping AC_TRN / AC_RND accumulator that holds the cast of a sample exactly (F_acc >= F_in, enough integer bits).  Then
//     acc = wrap( sum_j floor((x_j * 2^d * c_j + rnd) / 2^sh) ),   d = F_acc - F_in,  sh = F_coeff
// and every term fits int32:   d >= sh ("linear"):  x_j * c_j * 2^(d - sh)      -> v_dot2_i32_i16 on sample / coefficient pairs
//                              d <  sh ("per tap"): (x_j * c_j + r) >> (sh - d) -> mad / shift / add per tap.
// A wave owns runs of 512-output tiles of the (object, frame) rows: one coalesced 16-byte load per lane (+ a partial one
// for the window reach) goes to a wave-private LDS image of the tile, the frame-edge positions of AC_CLIP / AC_MIRROR are
// patched in from their source samples, every lane reads its 8 + TAPS - 1 window samples back with (unaligned) 16-byte
// reads, computes 8 consecutive outputs -- coefficients sit in SGPRs -- converts them (32-bit shift / clamp when nothing
// can wrap, else the 64-bit branch-free form) and stores 16 bytes per instruction.  The loads of the next tile are issued
// before the arithmetic of the current one.  No workgroup barrier: waves never share data.
struct MvStreamArgs {
  int16_t c16[66];            // coefficients, zero padded (pairs (c[2q], c[2q+1]) are the v_dot2 operands of even outputs)
  int16_t c16o[66];           // the same behind one zero: pairs (c[2q-1], c[2q]) for odd outputs, whose windows start in a high half
  int32_t taps, h, hb, off, mode, nxg;
  int32_t linear, ls, e, rnd_e;
  int32_t pk16;               // 32-bit epilogue into a signed 16-bit AC_SAT output at aligned frames: saturating packs
  int32_t cv32, s32, r32, lo32, hi32, ko32;   // 32-bit epilogue: q = (S + r32) >> s32, clamp, wrap (ko32 = 32 - W_out or 0)
  int32_t ka, rs, ls2, ko;    // 64-bit epilogue (see IdConv in intg_dump.hip)
  uint64_t am, om;
  int64_t rnd, lo, hi;
  int32_t out_eb, vec_ok, run_ok;   // vec_ok: every lane's eight outputs are an aligned 16 / 32 / 64 bytes; else run_ok: tiles leave through the image as aligned pieces
  int64_t n_sample, n_frames, opf, in_stride, out_stride, tpf, n_tiles, tiles_per_wave;
  const int16_t *x;
  void *y;
  int32_t xcd_map;   // XCD-affine block order (acdsp_dev.hpp: xcd_remap)
  // packed short frames (round 5, PK instantiations): pk_n = samples per frame (64 / 128 / 256; 0 = off), pk_pitch = image samples per frame
  // (n + 2 hb); the tile loads then start at the tile, not hb samples in front of it: every halo is a patch
  int32_t pk_n, pk_pitch, pk_sh;
  int32_t row_n, row_opf;   // ROW instantiations: samples / outputs per frame (the kernel's n_sample / opf are then the ROW's lengths, n_frames = 1); row_n 0: off
  // matrix-core instantiations (MF = K blocks of 64 samples): Toeplitz fragments [m][b][plane][lane] x 16 bytes (mv_avg_build_frags), and
  // 128 sum(c), what the re-biased low sample plane leaves out
  const uint32_t *frag;
  int32_t kbias;
};

typedef short v2s_t __attribute__((ext_vector_type(2)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef int v4i_mv __attribute__((ext_vector_type(4)));
// (tile loads keep the plain policy: non-temporal +3 % in time, profiles/r2_copy_probe_ldsdma.txt / DESIGN 5.1)
// 2-byte outputs (the bench row: one 16-byte store per lane) leave with the non-temporal policy; -DACDSP_MV_ST_PLAIN: plain (A/B)
#ifdef ACDSP_MV_ST_PLAIN
#define ACDSP_MV_ST(v, ptr) (*(ptr) = (v))
#else
#define ACDSP_MV_ST(v, ptr) __builtin_nontemporal_store(*reinterpret_cast<const v4u_t *>(&(v)), reinterpret_cast<v4u_t *>(ptr))
#endif

#ifndef ACDSP_MV_AHEAD
#define ACDSP_MV_AHEAD 2   // 4: -3 %, 8: -15 %, 16: -80 % on the bench row (profiles/r4_ab_up_store.txt, last block): more tiles in flight cost occupancy
#endif
// MF > 0 (round 6; windows of 17 taps and more, where the v_dot2 form is bound by VALU issue: 0.46 / 0.29 of the roofline at 33 / 65 taps): the
// window sums of a tile on the matrix cores.  The wave's image is the B operand as it lies -- column j' of v_mfma_i32_16x16x64_i8 is the 64 samples
// from image position 32 j' (+ 64 per further K block), a lane reads 32 aligned bytes and splits them into a low (re-biased by -128) and a high
// byte plane with v_perm_b32 -- and A is the Toeplitz matrix of the coefficients (balanced signed byte digits, built on the host), in TWO row
// orders: row i of fragment set m is output 32 j' + 8 (i >> 2) + (i & 3) + 4 m.  A lane's accumulator rows are then outputs 8 v .. 8 v + 7 of
// the tile, v = 4 (lane & 15) + (lane >> 4): eight consecutive outputs, the same epilogue and 16-byte stores as the v_dot2 form, a whole KB
// per store instruction.  8 MF MFMAs per 512 outputs (4 plane products x 2 row orders), 2 MF 16-byte LDS reads per lane instead of NR.
template <int NR, bool LINEAR, bool CV32, bool EDGE, bool PK = false, int MF = 0, bool RUN = false, int ROW = 0>   // ROW: tiles walk the ROW, frame edges fall inside them (a.row_n; 1: AC_CLIP / AC_MIRROR, 2: AC_WIN; see try_stream); RUN: output tiles that start off a 16-byte boundary leave as aligned runs (a.run_ok; as a run-time branch it cost the bench row 12 VGPRs and two waves per SIMD: -8 %); PK: packed short frames (MvStreamArgs::pk_*; as run-time branches they cost the bench row 1.4 %)
__global__ void __launch_bounds__(256) mv_avg_stream_kernel(MvStreamArgs a) {
  static_assert(MF == 0 || (LINEAR && !PK), "matrix-core form: linear class, whole-frame tiles");
  static_assert(ROW == 0 || (!PK && MF == 0 && !RUN && EDGE == (ROW == 1)), "row walk: v_dot2 form, aligned outputs");
  constexpr int REGION = 512 + 8 * NR + 8;      // samples of one wave's LDS image (+ 8: the odd-offset window reads one dword further)
  constexpr int NQ = 4 * NR - 3;                // coefficient pairs covering 8 NR - 7 taps (+ a zero)
  // (4- and 8-byte outputs turn their tile around in the same image -- 2 / 4 KB -- before it leaves: IMG)
  constexpr int IMG = REGION > 2048 ? REGION : 2048;
  __shared__ __attribute__((aligned(16))) int16_t sm[4][IMG];   // (exactly 16 KB per workgroup: 16 bytes more -- tried for run_ok tiles of 8-byte outputs -- cost the bench row 9 %, one workgroup per CU less)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int16_t *img = sm[wave];
  int bx, by_;
  xcd_remap(a.xcd_map, bx, by_);
  const int64_t t0 = ((int64_t)bx * 4 + wave) * a.tiles_per_wave;
  const int64_t t_end = (t0 + a.tiles_per_wave < a.n_tiles) ? t0 + a.tiles_per_wave : a.n_tiles;
  if (t0 >= t_end) { return; }
  const int n_my = (int)(t_end - t0);
  const uint32_t *cp = reinterpret_cast<const uint32_t *>(a.c16), *cpo = reinterpret_cast<const uint32_t *>(a.c16o);
  const int vlane = MF > 0 ? 4 * (lane & 15) + (lane >> 4) : lane;   // whose eight outputs this lane converts and stores
  v4i_mv A[MF > 0 ? 4 * MF : 1];                                        // [m][b][plane]
  if constexpr (MF > 0) {
#pragma unroll
    for (int f = 0; f < 4 * MF; f++) { A[f] = reinterpret_cast<const v4i_mv *>(a.frag)[f * 64 + lane]; }
  }

  // Tiles are fetched two ahead into alternating register sets.  (object, frame, tile) of the tile being fetched: one
  // division per wave, then counted up; past the wave's last tile the fetch repeats that tile (branch-free loop body, so
  // the waits in front of the LDS writes are counted vmcnt's on the older set only).
  struct Tile { uint4 mv, ev; int16_t fxl, fxr; int64_t ti, fr, obj; int bnd; };   // bnd (ROW): image position of the frame edge inside this tile's image, 2^20: none
  const int64_t pair0 = t0 / a.tpf;
  int64_t f_ti = t0 - pair0 * a.tpf, f_obj = pair0 / a.n_frames, f_fr = pair0 - f_obj * a.n_frames;
  int f_left = n_my;   // tiles not fetched yet
  auto fetch = [&](Tile &T) {   // aligned 8-sample groups: whole groups lie inside or outside the frame
    const int16_t *row = a.x + f_obj * a.in_stride + f_fr * a.n_sample;
    // ROW 2 (AC_WIN): the tile's first output o0 = 512 f_ti of the output row belongs to frame fr0 and reads the input row from in0 on; the outputs
    // from the next frame's first one (tile-relative index bnd) read TAPS - 1 samples further on -- the image is the contiguous input
    int64_t in0 = f_ti * 512 - (PK ? 0 : a.hb);
    if constexpr (ROW == 2) {
      const int64_t o0 = f_ti * 512;
      const uint32_t fr0 = (uint32_t)o0 / (uint32_t)a.row_opf;
      in0 = (int64_t)fr0 * a.row_n + (o0 - (int64_t)fr0 * a.row_opf);
      T.bnd = (int)((int64_t)(fr0 + 1) * a.row_opf - o0);
    }
    const int64_t n = a.n_sample, g = in0 + 8 * lane;
    T.ti = f_ti; T.fr = f_fr; T.obj = f_obj;
    // every load is unconditional (addresses clamped into the frame; what the clamped lanes fetch is never used), so
    // the loop body has no load under a branch and the waits stay counted
    const int64_t gc = g < 0 ? 0 : (g < n ? g : n - 8);
    { const v4u_t t_ = *reinterpret_cast<const v4u_t *>(row + gc); T.mv = make_uint4(t_.x, t_.y, t_.z, t_.w); }
    const int64_t g2 = in0 + 512 + 8 * (lane < a.nxg ? lane : 0);
    T.ev = *reinterpret_cast<const uint4 *>(row + (g2 < n ? g2 : n - 8));
#ifndef ACDSP_MV_LDS_EDGE
    if constexpr (ROW == 1) {
      // the one frame edge (a multiple of the frame length, 0 and the row's end included) that can lie inside the image (P0 - hb, P0 + 512 + hb):
      // frames are at least 512 + 2 hb samples long.  Its two halos take the mirrored / clipped samples of the frame they belong to.
      const int64_t P0 = f_ti * 512, c = P0 - a.hb + 1;
      const int64_t B = c <= 0 ? 0 : (int64_t)(((uint32_t)c + (uint32_t)a.row_n - 1u) / (uint32_t)a.row_n) * a.row_n;
      T.bnd = B < P0 + 512 + a.hb ? (int)(B - (P0 - a.hb)) : (1 << 20);
      const int l = lane < a.h ? lane : 0;
      int64_t sl = a.mode == 2 ? B : B + 1 + l, sr = a.mode == 2 ? B - 1 : B - 2 - l;
      sl = sl > n - 1 ? n - 1 : sl;                  // (the row's end: no frame behind it; the row's start: none in front)
      sr = sr < 0 ? 0 : (sr > n - 1 ? n - 1 : sr);
      T.fxl = row[sl];
      T.fxr = row[sr];
    } else if constexpr (EDGE) {   // frame-edge patches of AC_CLIP / AC_MIRROR: the source samples of positions -1 - lane and n + lane
      const int l = lane < a.h ? lane : 0;
      T.fxl = row[a.mode == 2 ? 0 : 1 + l];
      T.fxr = row[a.mode == 2 ? n - 1 : n - 2 - l];
    }
#endif
    if (f_left > 1) {
      f_left--;
      if (++f_ti == a.tpf) {
        f_ti = 0;
        if (++f_fr == a.n_frames) { f_fr = 0; f_obj++; }
      }
    }
  };
  auto process = [&](Tile &T) {
    const int64_t ti = T.ti, fr = T.fr, obj = T.obj, p0 = ti * 512 - a.hb;
    const int bnd = ROW != 0 ? T.bnd : 0;
    // image position of this lane's eight samples: 8 lane, or -- packed short frames -- frame f' = 8 lane / n of the tile at hb + f' (n + 2 hb)
    // (ROW: the samples behind the frame edge sit 2 hb further on -- the gap holds the right halo of the frame that ends and the left halo of the one that begins)
    const int gap = 2 * a.hb;
    const int lb = PK ? a.hb + ((8 * lane) >> a.pk_sh) * a.pk_pitch + ((8 * lane) & (a.pk_n - 1)) : (ROW == 1 ? 8 * lane + (8 * lane >= bnd ? gap : 0) : 8 * lane);
    *reinterpret_cast<uint4 *>(img + lb) = T.mv;
    if (lane < a.nxg) { *reinterpret_cast<uint4 *>(img + 512 + 8 * lane + (ROW == 1 && 512 + 8 * lane >= bnd ? gap : 0)) = T.ev; }
    if constexpr (ROW == 1) {
      if (lane < a.h && bnd < (1 << 20)) {
        img[bnd + lane] = T.fxr;
        img[bnd + gap - 1 - lane] = T.fxl;
      }
    } else if constexpr (EDGE) {   // positions outside [0, n) take the clipped / mirrored source sample
      if constexpr (PK) {
        // every frame of the tile has both its halos inside the image: filled from the staged samples themselves
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n_patch = (512 >> a.pk_sh) * 2 * a.h;
        for (int e = lane; e < n_patch; e += 64) {
          const int fq = e / (2 * a.h), k2 = e - fq * 2 * a.h, right = k2 >= a.h, k = right ? k2 - a.h : k2;
          const int base = a.hb + fq * a.pk_pitch;
          const int dst = right ? base + a.pk_n + k : base - 1 - k;
          const int src = a.mode == 2 ? (right ? base + a.pk_n - 1 : base) : (right ? base + a.pk_n - 2 - k : base + 1 + k);
          img[dst] = img[src];
        }
      } else {
#ifdef ACDSP_MV_LDS_EDGE
        // A/B variant, off: the sources of a frame's out-of-frame positions lie inside the tile's own image (the first h + 1 / last h + 1 samples of
        // the frame), so the two scattered 2-byte loads per tile (fxl / fxr) could go.  Same box, three passes: 0.7625 ms with them, 0.7638 without.
        const bool left = ti == 0, right = a.n_sample - p0 < REGION;
        if (left || right) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (lane < a.h) {
            if (left) { img[a.hb - 1 - lane] = img[a.mode == 2 ? a.hb : a.hb + 1 + lane]; }
            const int64_t idx = a.n_sample + lane - p0;
            if (right && idx < REGION) { img[idx] = img[a.mode == 2 ? a.n_sample - 1 - p0 : a.n_sample - 2 - lane - p0]; }
          }
        }
#else
        if (lane < a.h) {
          if (ti == 0) { img[a.hb - 1 - lane] = T.fxl; }
          const int64_t idx = a.n_sample + lane - p0;
          if (idx < REGION) { img[idx] = T.fxr; }
        }
#endif
      }
    }
    fetch(T);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // The window starts a.off samples into the image: any even byte offset.  A DS read off its natural alignment is replayed lane by lane
    // (~85 instead of ~8 cycles per wave read, profiles/r5_lds_align.txt; rounds 2 - 4 read 16 bytes at 2-byte alignment here -- 8 bytes into
    // a 16-byte granule on the bench row), so the widest reads the offset is aligned for (the offset is wave-uniform: uniform branches).
    int S[8];
    const int s_init = CV32 ? a.r32 : 0;   // the 32-bit epilogue's rounding constant rides in the sum (|sum| < 2^30, r32 <= 2^29)
    if constexpr (MF > 0) {
      const unsigned char *ib = reinterpret_cast<const unsigned char *>(img) + 64 * (lane & 15) + 32 * (lane >> 4);
      const v4i_mv z4 = {0, 0, 0, 0};
      v4i_mv acc[2][3] = {{z4, z4, z4}, {z4, z4, z4}};   // [row order m][weight 1, 2^8, 2^16]
#pragma unroll
      for (int b = 0; b < MF; b++) {
        const uint4 d0 = *reinterpret_cast<const uint4 *>(ib + 128 * b), d1 = *reinterpret_cast<const uint4 *>(ib + 128 * b + 16);
        v4i_mv lo, hi;
        lo.x = (int)(__builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u) ^ 0x80808080u); lo.y = (int)(__builtin_amdgcn_perm(d0.w, d0.z, 0x06040200u) ^ 0x80808080u);
        lo.z = (int)(__builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u) ^ 0x80808080u); lo.w = (int)(__builtin_amdgcn_perm(d1.w, d1.z, 0x06040200u) ^ 0x80808080u);
        hi.x = (int)__builtin_amdgcn_perm(d0.y, d0.x, 0x07050301u); hi.y = (int)__builtin_amdgcn_perm(d0.w, d0.z, 0x07050301u);
        hi.z = (int)__builtin_amdgcn_perm(d1.y, d1.x, 0x07050301u); hi.w = (int)__builtin_amdgcn_perm(d1.w, d1.z, 0x07050301u);
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const v4i_mv cl = A[(m * MF + b) * 2], ch = A[(m * MF + b) * 2 + 1];
          acc[m][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cl, lo, acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cl, hi, acc[m][1], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ch, lo, acc[m][1], 0, 0, 0);
          acc[m][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ch, hi, acc[m][2], 0, 0, 0);
        }
      }
      // every partial stays far inside int32; the recombination is modular and the sum itself is inside 2^30 (host-checked)
#pragma unroll
      for (int m = 0; m < 2; m++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          S[4 * m + r] = (int)((uint32_t)acc[m][0][r] + ((uint32_t)acc[m][1][r] << 8) + ((uint32_t)acc[m][2][r] << 16) + (uint32_t)(s_init + a.kbias));
        }
      }
    } else {
    uint32_t R[4 * NR + 1];
    {
      const unsigned char *wb = reinterpret_cast<const unsigned char *>(img + (PK ? lb - a.hb : (ROW == 1 ? 8 * lane + (8 * lane + a.hb >= bnd ? gap : 0) : (ROW == 2 ? 8 * lane + (8 * lane >= bnd ? 2 * a.h : 0) : lb)))) + 2 * a.off;
      const int bo = 2 * a.off;
      if ((bo & 15) == 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) { const uint4 v = *reinterpret_cast<const uint4 *>(wb + 16 * r); R[4 * r] = v.x; R[4 * r + 1] = v.y; R[4 * r + 2] = v.z; R[4 * r + 3] = v.w; }
      } else if ((bo & 7) == 0) {
#pragma unroll
        for (int r = 0; r < 2 * NR; r++) { const uint2 v = *reinterpret_cast<const uint2 *>(wb + 8 * r); R[2 * r] = v.x; R[2 * r + 1] = v.y; }
      } else if ((bo & 3) == 0) {
#pragma unroll
        for (int r = 0; r < 4 * NR; r++) { R[r] = *reinterpret_cast<const uint32_t *>(wb + 4 * r); }
      } else {   // an odd sample offset: aligned dwords one sample early, shifted down by 16 bits
        uint32_t prev = *reinterpret_cast<const uint32_t *>(wb - 2);
#pragma unroll
        for (int r = 0; r < 4 * NR; r++) {
          const uint32_t nxt = *reinterpret_cast<const uint32_t *>(wb + 2 + 4 * r);
          R[r] = __builtin_amdgcn_alignbit(nxt, prev, 16);
          prev = nxt;
        }
      }
    }
    R[4 * NR] = 0;
    if constexpr (LINEAR) {
      // output i of the lane starts at sample i of its dwords: even outputs pair (x[i+2q], x[i+2q+1]) = dword i/2 + q with
      // (c[2q], c[2q+1]); odd outputs pair the same aligned dwords with the coefficient pairs shifted by one tap,
      // (c[2q-1], c[2q]) with c[-1] = 0 -- no realignment of the samples
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int sum = s_init;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s_t, R[i / 2 + q]), __builtin_bit_cast(v2s_t, (i & 1) ? cpo[q] : cp[q]), sum, false);
        }
        S[i] = sum;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int sum = s_init;
#pragma unroll
        for (int j = 0; j < 8 * NR - 7; j++) {
          const int k = i + j;
          const int xs = (k & 1) ? ((int)R[k / 2] >> 16) : (int)(int16_t)R[k / 2];
          sum += (xs * (int)a.c16[j] + a.rnd_e) >> a.e;
        }
        S[i] = sum;
      }
    }
    }
    __builtin_amdgcn_wave_barrier();   // every lane has its window: the image may be overwritten
    const int64_t k0 = ti * 512 + 8 * vlane;
    bool done16 = false;
    if constexpr (CV32 && !RUN) {
      // signed 16-bit AC_SAT outputs (the bench row's OUT_TYPE): shift, then v_cvt_pk_i16_i32 saturates and packs two outputs per instruction --
      // 12 instead of 28 VALU instructions per lane and tile on a kernel that is bound by VALU issue
      if (a.pk16) {
        if (k0 < a.opf) {
          typedef short v2s16_ __attribute__((ext_vector_type(2)));
          uint4 v;
          v.x = __builtin_bit_cast(uint32_t, (v2s16_)__builtin_amdgcn_cvt_pk_i16(S[0] >> a.s32, S[1] >> a.s32));
          v.y = __builtin_bit_cast(uint32_t, (v2s16_)__builtin_amdgcn_cvt_pk_i16(S[2] >> a.s32, S[3] >> a.s32));
          v.z = __builtin_bit_cast(uint32_t, (v2s16_)__builtin_amdgcn_cvt_pk_i16(S[4] >> a.s32, S[5] >> a.s32));
          v.w = __builtin_bit_cast(uint32_t, (v2s16_)__builtin_amdgcn_cvt_pk_i16(S[6] >> a.s32, S[7] >> a.s32));
          ACDSP_MV_ST(v, reinterpret_cast<uint4 *>((int16_t *)a.y + (obj * a.out_stride + fr * a.opf + k0)));
        }
        done16 = true;
      }
    }
    if (!done16 && k0 < a.opf) {
      const int64_t yb = obj * a.out_stride + fr * a.opf + k0;
      int64_t ov[8];                                  // OUT raw words (the low out_eb bytes are what leaves)
      if constexpr (CV32) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int q = S[i] >> a.s32;
          q = q < a.lo32 ? a.lo32 : (q > a.hi32 ? a.hi32 : q);
          if (a.ko32) { q = (int)((uint32_t)q << a.ko32) >> a.ko32; }
          if (a.om != ~uint64_t(0)) { q = (int)((uint32_t)q & (uint32_t)a.om); }
          ov[i] = a.om != ~uint64_t(0) ? (int64_t)(uint32_t)q : (int64_t)q;   // unsigned OUT: the masked word zero-extends; signed: sign-extends
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int64_t acc = (int64_t)(((uint64_t)((int64_t)((uint64_t)(int64_t)S[i] << (a.ls + a.ka)) >> a.ka)) & a.am);   // wrap to ACC_TYPE
          int64_t qv = (int64_t)((uint64_t)((acc + a.rnd) >> a.rs) << a.ls2);
          qv = qv < a.lo ? a.lo : (qv > a.hi ? a.hi : qv);
          ov[i] = (int64_t)(((uint64_t)((int64_t)((uint64_t)qv << a.ko) >> a.ko)) & a.om);
        }
      }
      if (a.vec_ok) {
        if (a.out_eb == 2) {
          uint4 v;
          v.x = __builtin_amdgcn_perm((uint32_t)ov[1], (uint32_t)ov[0], 0x05040100u); v.y = __builtin_amdgcn_perm((uint32_t)ov[3], (uint32_t)ov[2], 0x05040100u);
          v.z = __builtin_amdgcn_perm((uint32_t)ov[5], (uint32_t)ov[4], 0x05040100u); v.w = __builtin_amdgcn_perm((uint32_t)ov[7], (uint32_t)ov[6], 0x05040100u);
          ACDSP_MV_ST(v, reinterpret_cast<uint4 *>((int16_t *)a.y + yb));
        } else {
          // 4- / 8-byte containers: a lane's eight outputs are 32 / 64 contiguous bytes, and stored from the registers every instruction
          // covered its 2 / 4 KB with 16-byte pieces 32 / 64 bytes apart (0.48 / 0.38 of the roofline where the 2-byte row runs 0.69).
          // The tile goes through the wave's LDS image once -- rows of 16 bytes per lane out -- and leaves in whole 1 KB runs.
          unsigned char *ib = reinterpret_cast<unsigned char *>(img);
          if (a.out_eb == 4) {
            uint4 *w = reinterpret_cast<uint4 *>(ib + 32 * vlane);
            w[0] = make_uint4((uint32_t)ov[0], (uint32_t)ov[1], (uint32_t)ov[2], (uint32_t)ov[3]);
            w[1] = make_uint4((uint32_t)ov[4], (uint32_t)ov[5], (uint32_t)ov[6], (uint32_t)ov[7]);
          } else {
            ulonglong2 *w = reinterpret_cast<ulonglong2 *>(ib + 64 * vlane);
#pragma unroll
            for (int i = 0; i < 8; i += 2) { w[i / 2] = make_ulonglong2((uint64_t)ov[i], (uint64_t)ov[i + 1]); }
          }
        }
      } else if (RUN) {
        // output frames that start off a 16-byte boundary (round 6: AC_WIN with TAPS - 1 no multiple of 8 -- 1012 outputs per frame -- ran eight
        // element stores per lane, 0.25 - 0.30 of the roofline): the tile's outputs go to the image at the byte offset their first one has
        // inside its 16-byte granule, and leave below as aligned 16-byte pieces
        // (no guards: what a lane writes past the tile's last output stays in the image)
        const int mis = (int)(((uintptr_t)a.y + (uint64_t)(yb - 8 * vlane) * a.out_eb) & 15);
        unsigned char *ib = reinterpret_cast<unsigned char *>(img) + mis + 8 * vlane * a.out_eb;
        if (a.out_eb == 2) {
          const uint32_t p0_ = __builtin_amdgcn_perm((uint32_t)ov[1], (uint32_t)ov[0], 0x05040100u), p1_ = __builtin_amdgcn_perm((uint32_t)ov[3], (uint32_t)ov[2], 0x05040100u);
          const uint32_t p2_ = __builtin_amdgcn_perm((uint32_t)ov[5], (uint32_t)ov[4], 0x05040100u), p3_ = __builtin_amdgcn_perm((uint32_t)ov[7], (uint32_t)ov[6], 0x05040100u);
          if (!(mis & 2)) {         // wave-uniform: dword-aligned pairs
            uint32_t *w = reinterpret_cast<uint32_t *>(ib);
            w[0] = p0_; w[1] = p1_; w[2] = p2_; w[3] = p3_;
          } else {                  // one element, three straddling pairs, one element
            reinterpret_cast<int16_t *>(ib)[0] = (int16_t)ov[0];
            uint32_t *w = reinterpret_cast<uint32_t *>(ib + 2);
            w[0] = __builtin_amdgcn_alignbit(p1_, p0_, 16); w[1] = __builtin_amdgcn_alignbit(p2_, p1_, 16); w[2] = __builtin_amdgcn_alignbit(p3_, p2_, 16);
            reinterpret_cast<int16_t *>(ib)[7] = (int16_t)ov[7];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) { reinterpret_cast<int32_t *>(ib)[i] = (int32_t)ov[i]; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (k0 + i < a.opf) { store_raw(a.y, yb + i, a.out_eb, ov[i]); }
        }
      }
    }
    if (RUN && !a.vec_ok) {   // wave-uniform: second half of the unaligned turn-around
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned char *ib = reinterpret_cast<const unsigned char *>(img);
      const uintptr_t b0 = (uintptr_t)a.y + (uint64_t)(obj * a.out_stride + fr * a.opf + ti * 512) * a.out_eb;
      const int mis = (int)(b0 & 15);
      const int64_t left = a.opf - ti * 512;
      const int total = mis + (int)(left < 512 ? left : 512) * a.out_eb;      // image bytes [mis, total) are this tile's outputs
      unsigned char *g = reinterpret_cast<unsigned char *>(b0 - mis);
#pragma unroll
      for (int k = 0; k < 3; k++) {                                             // 512 x 4 bytes + 15 = 129 pieces at most
        const int o = 16 * (lane + 64 * k);
        if (o >= mis && o + 16 <= total) {
          const uint4 v = *reinterpret_cast<const uint4 *>(ib + o);
          ACDSP_MV_ST(v, reinterpret_cast<uint4 *>(g + o));
        }
      }
      {
        // the first and the last piece of the run hold up to seven 2-byte units each: ONE store instruction for both, a unit per lane (lanes 0 - 7
        // the head, 8 - 15 the tail; as 2-byte stores per piece they were 16 more memory instructions per tile than the aligned form's one)
        const int hf = (mis + 15) & ~15, tf = total & ~15, t0b = tf > hf ? tf : hf;
        const int bb = (lane < 8 ? mis : t0b) + 2 * (lane & 7);
        if (lane < 16 && bb < total && (lane >= 8 || bb < hf)) { *reinterpret_cast<int16_t *>(g + bb) = *reinterpret_cast<const int16_t *>(ib + bb); }
      }
      __builtin_amdgcn_wave_barrier();   // read back before the next tile is staged over it
    }
    {
      if (a.vec_ok && a.out_eb != 2) {   // wave-uniform: the second half of the turn-around above
        __builtin_amdgcn_wave_barrier();
        const unsigned char *ib = reinterpret_cast<const unsigned char *>(img);
        const int64_t rem_b = (a.opf - ti * 512) * a.out_eb;                 // bytes of this tile that exist (a multiple of 32)
        unsigned char *yt = reinterpret_cast<unsigned char *>(a.y) + (obj * a.out_stride + fr * a.opf + ti * 512) * a.out_eb;
        const int nk = a.out_eb == 4 ? 2 : 4;
        for (int k = 0; k < nk; k++) {
          const int off = 1024 * k + 16 * lane;
          if (off < rem_b) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ib + off);
            ACDSP_MV_ST(v, reinterpret_cast<uint4 *>(yt + off));
          }
        }
        __builtin_amdgcn_wave_barrier();   // read back before the next tile is staged over it
      }
    }
  };
  // ACDSP_MV_AHEAD register sets: tiles in flight per wave (A/B knob; 2 = rounds 2 - 3)
  Tile T[ACDSP_MV_AHEAD];
#pragma unroll
  for (int j = 0; j < ACDSP_MV_AHEAD; j++) {
    T[j].mv = T[j].ev = make_uint4(0, 0, 0, 0);
    T[j].fxl = T[j].fxr = 0;
  }
#pragma unroll
  for (int j = 0; j < ACDSP_MV_AHEAD; j++) { fetch(T[j]); }
  for (int it = 0; it < n_my; it += ACDSP_MV_AHEAD) {
#pragma unroll
    for (int j = 0; j < ACDSP_MV_AHEAD; j++) {
      if (j == 0 || it + j < n_my) { process(T[j]); }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 32-bit samples (round 5; the header's usage example declares IN_TYPE ac_fixed<32,16>, ac_mv_avg.h:47): the order-free class on int32
// containers whose cast to ACC_TYPE is exact -- xq = x << d, d = F_acc - F_in >= 0, enough integer bits -- and coefficients inside int32:
//     acc = wrap( sum_j floor((x_j c_j 2^d + rnd) / 2^sh) ),  sh = F_coeff;  x_j c_j is ONE v_mad_i64_i32, host-checked to stay inside 2^62.
// A 256-thread block takes a tile of 1024 outputs of one (object, frame): the 1024 + TAPS - 1 window samples go to LDS as int32 with
// the boundary rule applied (coalesced loads), a thread owns four consecutive outputs and slides an eight-register window over the taps
// -- one aligned 16-byte LDS read per four taps; coefficients are uniform reads -- then converts (requant64: any OUT_TYPE) and
// stores its four outputs as one or two 16-byte vectors.  Before: 128-bit per-tap kernel, 0.034 of the roofline on <32,16> samples.
struct MvW32Args {
  int32_t d, sh, linear, ls, e; int64_t rnd_e; int32_t vec_ok;
  // branch-free ACC -> OUT (CONV instantiations; IdConv's form in intg_dump.hip): wrap to ACC_TYPE, rounding shift, clamp, wrap to OUT_TYPE
  int32_t ka, rs, ls2, ko; uint64_t am, om; int64_t rnd, lo, hi;
};

template <bool LINEAR, bool CONV>   // LINEAR: d >= sh, the products add up unshifted (one v_mad_i64_i32 per tap and output); CONV: TRN / RND into WRAP / SAT
__global__ void __launch_bounds__(256) mv_avg_w32_kernel(MvAvgParams p, MvW32Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char w32_lds[];
  const int h = p.taps / 2, tid = threadIdx.x;
  const int nwin = (1024 + p.taps - 1 + 3 + 4) & ~3;            // window dwords, padded: the last 16-byte read of a thread may reach one group further
  int32_t *win = (int32_t *)w32_lds;
  int32_t *cf = win + nwin;                                       // [taps] coefficients (int32 by the host check)
  for (int i = tid; i < p.taps; i += 256) { cf[i] = (int32_t)p.coeffs[i]; }
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  const int64_t first = p.win_mode == 0 ? h : 0;
  for (int64_t pr = blockIdx.y; pr < pairs; pr += gridDim.y) {
    const int64_t obj = pr / p.n_frames, fr = pr % p.n_frames;
    const int32_t *xrow = (const int32_t *)p.x + obj * p.in_stride + fr * p.n_sample;
    const int64_t ybase = obj * p.out_stride + fr * p.out_per_frame;
    const int64_t m0 = first + (int64_t)blockIdx.x * 1024;
    __syncthreads();
    for (int j = tid; j < nwin; j += 256) {
      int64_t pos = m0 - h + j;
      if (p.win_mode != 0) { pos = fold_pos(pos, p.n_sample, p.win_mode); }
      // (round 6: 16-bit containers as well -- frames that are no multiple of 8 samples, weights beyond the int32 class of the streaming kernel: 0.03 on the int64 kernel)
      win[j] = (pos >= 0 && pos < p.n_sample) ? (p.in_eb == 4 ? xrow[pos] : (int32_t)load_raw(p.x, obj * p.in_stride + fr * p.n_sample + pos, 2, p.in.S)) : 0;
    }
    __syncthreads();
    const int64_t k0 = (int64_t)blockIdx.x * 1024 + 4 * tid;    // first of this thread's four outputs within the frame
    if (k0 < p.out_per_frame) {
      typedef int v4i_ __attribute__((ext_vector_type(4)));
      const v4i_ *wv = (const v4i_ *)(win + 4 * tid);
      v4i_ cur = wv[0];
      uint64_t s[4] = {0, 0, 0, 0};
      for (int jg = 0; 4 * jg < p.taps; jg++) {
        const v4i_ nxt = wv[jg + 1];
        const int w8[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = 4 * jg + u;
          if (j < p.taps) {
            const int c = cf[j];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              if (LINEAR) { s[i] = (uint64_t)((int64_t)w8[u + i] * (int64_t)c + (int64_t)s[i]); }
              else { s[i] += (uint64_t)(((int64_t)w8[u + i] * (int64_t)c + a.rnd_e) >> a.e); }
            }
          }
        }
        cur = nxt;
      }
      int64_t o[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (CONV) {
          const int64_t acc = (int64_t)(((uint64_t)((int64_t)(s[i] << (a.ls + a.ka)) >> a.ka)) & a.am);   // (sum << ls) wrapped to ACC_TYPE
          int64_t qv = (int64_t)((uint64_t)((acc + a.rnd) >> a.rs) << a.ls2);
          qv = qv < a.lo ? a.lo : (qv > a.hi ? a.hi : qv);
          o[i] = (int64_t)(((uint64_t)((int64_t)((uint64_t)qv << a.ko) >> a.ko)) & a.om);
        } else {
          o[i] = requant64(wrap64((int64_t)(s[i] << a.ls), p.acc.W, p.acc.S), p.acc.F, p.out);
        }
      }
      if (a.vec_ok && k0 + 4 <= p.out_per_frame) {
        if (p.out_eb == 8) {
          typedef long v2l_ __attribute__((ext_vector_type(2)));
          v2l_ *yp = (v2l_ *)((int64_t *)p.y + ybase + k0);
          yp[0] = (v2l_){o[0], o[1]}; yp[1] = (v2l_){o[2], o[3]};
        } else if (p.out_eb == 4) {
          *(v4i_ *)((int32_t *)p.y + ybase + k0) = (v4i_){(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
        } else {
          typedef short v4s_ __attribute__((ext_vector_type(4)));
          *(v4s_ *)((int16_t *)p.y + ybase + k0) = (v4s_){(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
        }
      } else {
        for (int i = 0; i < 4 && k0 + i < p.out_per_frame; i++) { store_raw(p.y, ybase + k0 + i, p.out_eb, o[i]); }
      }
    }
  }
}

}  // namespace

// true: launched (mv_avg_w32_kernel's class and shape conditions)
static bool try_w32(const MvAvgParams &p, hipStream_t s) {
  static const bool off = getenv("ACDSP_NO_MVAVG_W32") != nullptr;   // A/B knob
  if (off || p.force_generic || !p.h_coeffs || (p.in_eb != 4 && p.in_eb != 2) || !(p.in.S || p.in.W <= 31) || p.taps > 1025) { return false; }
  if (p.acc.O != ACDSP_WRAP || (p.acc.Q != ACDSP_TRN && p.acc.Q != ACDSP_RND) || p.cf.F >= 62) { return false; }   // order-free class
  const int d = p.acc.F - p.in.F, sh = p.cf.F;
  const int i_in = p.in.W - p.in.F, i_acc = p.acc.W - p.acc.F;
  if (d < 0 || sh < 0) { return false; }
  if (p.in.S ? (!p.acc.S || i_acc < i_in) : (i_acc < i_in + (p.acc.S ? 1 : 0))) { return false; }   // the cast (ACC_TYPE) w[j] is exact
  int cbits = 1;
  for (int i = 0; i < p.taps; i++) {
    const int64_t c = p.h_coeffs[i];
    if (c < INT32_MIN || c > INT32_MAX) { return false; }
    const uint64_t m = (uint64_t)(c < 0 ? ~c : c);
    int b = 1;
    while (b < 64 && (m >> (b - 1)) != 0) { b++; }
    if (b > cbits) { cbits = b; }
  }
  MvW32Args a;
  memset(&a, 0, sizeof a);
  a.d = d; a.sh = sh; a.linear = d >= sh; a.ls = a.linear ? d - sh : 0; a.e = a.linear ? 0 : sh - d;
  // linear: the 32 x 32-bit products add up mod 2^64 and the shift comes once, after the sum; per tap: |x c| + rnd inside int64
  if (a.ls > 62 || a.e > 62) { return false; }
  a.rnd_e = (!a.linear && p.acc.Q == ACDSP_RND) ? (int64_t(1) << (a.e - 1)) : 0;
  if (a.e > 0 && p.in.W + cbits > 61) { return false; }
  a.vec_ok = p.out_per_frame % 4 == 0 && p.out_stride % 4 == 0 && ((uintptr_t)p.y % 16) == 0 && (p.out_eb == 2 || p.out_eb == 4 || p.out_eb == 8);
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  dim3 grid((unsigned)((p.out_per_frame + 1023) / 1024), (unsigned)(pairs < 65535 ? pairs : 65535));
  const int nwin = (1024 + p.taps - 1 + 3 + 4) & ~3;
  const size_t lds = (size_t)(nwin + p.taps) * sizeof(int32_t);
  // ACC -> OUT without branches where the modes and the 64-bit arithmetic allow it (the conditions of intg_dump.hip: make_conv)
  const int rs = p.acc.F - p.out.F;
  const bool wrap_o = p.out.O == ACDSP_WRAP;
  a.ka = 64 - p.acc.W; a.am = p.acc.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.acc.W));
  a.rs = rs > 0 ? rs : 0; a.ls2 = rs < 0 ? -rs : 0;
  a.rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (int64_t(1) << (rs - 1)) : 0;
  if (p.out.O == ACDSP_SAT) { a.lo = p.out.lo; a.hi = p.out.hi; a.ko = 0; a.om = ~uint64_t(0); }
  else { a.lo = INT64_MIN; a.hi = INT64_MAX; a.ko = 64 - p.out.W; a.om = p.out.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.out.W)); }
  const bool conv = (p.out.Q == ACDSP_TRN || p.out.Q == ACDSP_RND) && (wrap_o || p.out.O == ACDSP_SAT) && a.rs <= 60 && a.ls + a.ka < 64 &&
                    (a.rnd == 0 || p.acc.W <= 61) && (a.ls2 == 0 || p.acc.W + a.ls2 <= 61 || wrap_o) &&
                    (p.acc.S || p.acc.W <= 62 || (rs == 0 && wrap_o)) && (wrap_o || p.out.S ? p.out.W <= 64 : p.out.W <= 62);
  if (a.linear) {
    if (conv) { hipLaunchKernelGGL((mv_avg_w32_kernel<true, true>), grid, dim3(256), lds, s, p, a); }
    else { hipLaunchKernelGGL((mv_avg_w32_kernel<true, false>), grid, dim3(256), lds, s, p, a); }
  } else {
    if (conv) { hipLaunchKernelGGL((mv_avg_w32_kernel<false, true>), grid, dim3(256), lds, s, p, a); }
    else { hipLaunchKernelGGL((mv_avg_w32_kernel<false, false>), grid, dim3(256), lds, s, p, a); }
  }
  return true;
}

// true: launched.  Class and shape conditions of the streaming kernel (see above).
static bool try_stream(const MvAvgParams &p, hipStream_t s, bool *mfma) {
  *mfma = false;
  static const bool off = getenv("ACDSP_NO_MVAVG_STREAM") != nullptr;   // A/B knob
  if (off || p.force_generic || !p.h_coeffs || p.taps > 65 || p.in_eb != 2 || !(p.in.S || p.in.W <= 15)) { return false; }
  if (p.n_sample < p.taps || p.n_sample % 8 != 0 || p.in_stride % 8 != 0 || ((uintptr_t)p.x % 16) != 0) { return false; }
  // the cast (ACC_TYPE) w[j] is exact and the accumulator wraps
  const int d = p.acc.F - p.in.F, sh = p.cf.F;
  const int i_in = p.in.W - p.in.F, i_acc = p.acc.W - p.acc.F;
  if (d < 0 || sh < 0 || p.acc.O != ACDSP_WRAP || (p.acc.Q != ACDSP_TRN && p.acc.Q != ACDSP_RND)) { return false; }
  if (p.in.S ? (!p.acc.S || i_acc < i_in) : (i_acc < i_in + (p.acc.S ? 1 : 0))) { return false; }
  int64_t sum_abs = 0;
  for (int i = 0; i < p.taps; i++) {
    if (p.h_coeffs[i] < -32768 || p.h_coeffs[i] > 32767) { return false; }
    sum_abs += p.h_coeffs[i] < 0 ? -p.h_coeffs[i] : p.h_coeffs[i];
  }
  if (sum_abs > 32767) { return false; }            // |sum of products| < 2^30: every intermediate stays inside int32
  MvStreamArgs a;
  memset(&a, 0, sizeof a);
  for (int i = 0; i < p.taps; i++) { a.c16[i] = (int16_t)p.h_coeffs[i]; a.c16o[i + 1] = (int16_t)p.h_coeffs[i]; }
  a.taps = p.taps; a.h = p.taps / 2; a.mode = p.win_mode;
  a.hb = p.win_mode == 0 ? 0 : 8 * ((a.h + 7) / 8);
  a.off = p.win_mode == 0 ? 0 : a.hb - a.h;
  a.nxg = (a.off + p.taps - 1 + 7) / 8;
  a.linear = d >= sh;
  a.ls = a.linear ? d - sh : 0;
  a.e = a.linear ? 0 : sh - d;
  if (a.e > 30 || a.ls > 62) { return false; }
  a.rnd_e = (!a.linear && p.acc.Q == ACDSP_RND) ? (1 << (a.e - 1)) : 0;
  if (p.acc.W > 61 || a.ls >= p.acc.W) { return false; }
  // ACC -> OUT
  if ((p.out.Q != ACDSP_TRN && p.out.Q != ACDSP_RND) || (p.out.O != ACDSP_WRAP && p.out.O != ACDSP_SAT) || p.out.W > 62) { return false; }
  const int rs = p.acc.F - p.out.F;
  a.ka = 64 - p.acc.W; a.am = p.acc.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.acc.W));
  a.rs = rs > 0 ? rs : 0; a.ls2 = rs < 0 ? -rs : 0;
  if (a.rs > 60 || p.acc.W + a.ls2 > 61) { return false; }
  a.rnd = (p.out.Q == ACDSP_RND && rs > 0) ? (int64_t(1) << (rs - 1)) : 0;
  if (p.out.O == ACDSP_SAT) { a.lo = p.out.lo; a.hi = p.out.hi; a.ko = 0; a.om = ~uint64_t(0); }
  else { a.lo = INT64_MIN; a.hi = INT64_MAX; a.ko = 64 - p.out.W; a.om = p.out.S ? ~uint64_t(0) : (~uint64_t(0) >> (64 - p.out.W)); }
  // 32-bit form: the accumulator cannot wrap (signed, W_acc >= 31 + ls) and the output shift swallows ls
  // (|sum| <= sum|c| * 2^(W_in - 1): samples narrower than 16 bits need a narrower accumulator)
  int sum_bits = 0;
  while (sum_bits < 62 && (int64_t(1) << sum_bits) <= sum_abs * (int64_t(1) << (p.in.W - (p.in.S ? 1 : 0)))) { sum_bits++; }   // |sum| < 2^sum_bits
  a.cv32 = p.acc.S && p.acc.W >= sum_bits + 1 + a.ls && rs >= a.ls && rs - a.ls <= 30 && (p.out.S || p.out.W <= 31);
  if (a.cv32) {
    a.s32 = rs - a.ls;
    a.r32 = (p.out.Q == ACDSP_RND && a.s32 > 0) ? (1 << (a.s32 - 1)) : 0;
    a.lo32 = INT32_MIN; a.hi32 = INT32_MAX; a.ko32 = 0;
    if (p.out.O == ACDSP_SAT) {
      if (p.out.lo > INT32_MIN) { a.lo32 = (int32_t)p.out.lo; }
      if (p.out.hi < INT32_MAX) { a.hi32 = (int32_t)p.out.hi; }
    } else if (p.out.W < 32) { a.ko32 = 32 - p.out.W; }
  }
  a.out_eb = p.out_eb;
  a.vec_ok = p.out_per_frame % 8 == 0 && p.out_stride % 8 == 0 && ((uintptr_t)p.y % 16) == 0;
  static const bool no_pk16 = getenv("ACDSP_NO_MVAVG_PK16") != nullptr;   // A/B knob
  a.pk16 = !no_pk16 && a.cv32 && a.vec_ok && p.out_eb == 2 && a.lo32 == -32768 && a.hi32 == 32767 && a.ko32 == 0 && a.om == ~uint64_t(0);
  static const bool no_run = getenv("ACDSP_NO_MVAVG_RUN") != nullptr;   // A/B knob: element stores for unaligned output frames
  a.run_ok = !a.vec_ok && !no_run && ((uintptr_t)p.y % p.out_eb) == 0 && p.out_eb <= 4;   // (8-byte outputs: 4 KB + the offset would not fit the image; they keep element stores, 64 contiguous bytes per lane)
  a.n_sample = p.n_sample; a.n_frames = p.n_frames; a.opf = p.out_per_frame; a.in_stride = p.in_stride; a.out_stride = p.out_stride;
  a.tpf = (p.out_per_frame + 511) / 512;
  a.n_tiles = (int64_t)p.n_obj * p.n_frames * a.tpf;
  // Frames shorter than a 512-output tile (round 5: 128-sample frames used 16 of a wave's 64 lanes, 0.21 of the roofline): with AC_CLIP /
  // AC_MIRROR every output of a frame depends on that frame's samples only, frames lie back to back in the row, so a tile takes 512 / n whole
  // frames -- each in its own slot of the LDS image, both halos patched from the staged samples -- and the row is walked as ONE frame.
  a.pk_n = 0; a.pk_pitch = 0; a.pk_sh = 0;
  static const bool no_pk = getenv("ACDSP_NO_MVAVG_PACK") != nullptr;   // A/B knob
  if (!no_pk && p.win_mode != 0 && (p.n_sample == 64 || p.n_sample == 128 || p.n_sample == 256) && p.n_frames % (512 / p.n_sample) == 0) {
    a.pk_n = (int32_t)p.n_sample; a.pk_pitch = a.pk_n + 2 * a.hb; a.pk_sh = p.n_sample == 64 ? 6 : (p.n_sample == 128 ? 7 : 8);
    a.nxg = 0;
    a.n_sample = p.n_sample * p.n_frames; a.n_frames = 1; a.opf = a.n_sample; a.tpf = a.opf / 512;
    a.n_tiles = (int64_t)p.n_obj * a.tpf;
    a.run_ok = 0;   // (packed tiles keep element stores when the output row is off the boundary)
  }
  // Frames that are no multiple of the 512-output tile (round 6: frames of 1000 samples at 0.57 where 1024 run 0.68).  Frame-relative tiles start wherever
  // their frame starts: every 1 KB store run and load straddles cache lines, and the frame's last tile is partly empty.  With AC_CLIP / AC_MIRROR input
  // and output positions of a row coincide, so the tiles can walk the ROW in aligned 512-output steps and take the frame edges where they fall: at most
  // one edge per tile image (frames of at least 512 + 2 hb samples), the samples behind it shifted by a gap of 2 hb that holds both halos (ROW instantiations).
  a.row_n = 0;
  static const bool no_row = getenv("ACDSP_NO_MVAVG_ROW") != nullptr;   // A/B knob
  const bool mf_wanted = [&] {
    ACDSP_TUNE_ENV(mf_env0, "ACDSP_MVAVG_MFMA_MIN");
    const int mn = mf_env0 ? atoi(mf_env0) : (p.win_mode == 0 ? 9 : 11);
    return p.frag && p.frag_nb > 0 && a.linear && mn > 0 && p.taps >= mn;
  }();
  if (!no_row && !a.pk_n && !mf_wanted && p.win_mode != 0 && p.n_sample % 512 != 0 && p.n_sample >= 512 + 2 * a.hb && a.vec_ok &&
      p.n_sample * p.n_frames < (int64_t(1) << 31) - 1024) {
    a.row_n = (int32_t)p.n_sample;
    a.n_sample = p.n_sample * p.n_frames; a.n_frames = 1; a.opf = a.n_sample; a.tpf = (a.opf + 511) / 512;
    a.n_tiles = (int64_t)p.n_obj * a.tpf;
  }
  // ... and AC_WIN (ROW 2): N - TAPS + 1 outputs per frame lie back to back in the output row while their windows jump TAPS - 1 samples at every frame
  // edge.  Where both are multiples of 8 (TAPS = 9, 17, 25 ...: the windows stay aligned) the tiles walk the OUTPUT row; the image is the contiguous input
  // and the lanes behind the edge start TAPS - 1 samples further on.
  a.row_opf = 0;
  {
    ACDSP_TUNE_ENV(roww_env, "ACDSP_MVAVG_ROWW_MAX");   // A/B knob: the longest window that walks the output row (longer ones: matrix cores, frame-relative)
    // (same box, 512 objects x 2^20 in frames of 1024: 9 taps 0.54 frame-relative v_dot2 / 0.59 matrix cores / 0.69 row walk; 17 taps 0.62 / 0.61 / 0.62; 25 taps
    // 0.55 / 0.59 / 0.55; 33 taps 0.50 / 0.61 / 0.49 -- the v_dot2 form is VALU-bound from 17 taps on, where the walk buys nothing)
    const int roww_max = roww_env ? atoi(roww_env) : 9;
    if (!no_row && p.win_mode == 0 && p.taps <= roww_max && p.taps > 1 && (p.taps - 1) % 8 == 0 && p.out_per_frame >= 512 && p.out_per_frame % 512 != 0 && a.vec_ok &&
        p.n_sample * p.n_frames < (int64_t(1) << 31) - 1024) {
      a.row_n = (int32_t)p.n_sample; a.row_opf = (int32_t)p.out_per_frame;
      a.n_sample = p.n_sample * p.n_frames; a.n_frames = 1; a.opf = p.out_per_frame * p.n_frames; a.tpf = (a.opf + 511) / 512;
      a.n_tiles = (int64_t)p.n_obj * a.tpf;
      a.nxg += (p.taps - 1) / 8;
    }
  }
  a.tiles_per_wave = 16;   // 16 KB spans (8 / 16 tiles alike, 32: -5 %, 64: -9 %: profiles/r3_span_sweep.txt)
  while (a.tiles_per_wave > 1 && a.n_tiles / a.tiles_per_wave < 16384) { a.tiles_per_wave /= 2; }
  ACDSP_TUNE_ENV(tpw_env, "ACDSP_MVAVG_TPW");   // tuning knob: tiles per wave
  if (tpw_env && atoi(tpw_env) > 0) { a.tiles_per_wave = atoi(tpw_env); }
  a.x = (const int16_t *)p.x; a.y = p.y;
  const int64_t waves = (a.n_tiles + a.tiles_per_wave - 1) / a.tiles_per_wave;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7FFFFFFF) { return false; }
  dim3 grid((unsigned)blocks);
  a.xcd_map = (xcd_map_wanted(false) && blocks % 8 == 0) ? 1 : 0;
  // windows of 17 taps and more: the sums on the matrix cores (fragments built by the handle at set_coeffs: mv_avg_build_frags)
  ACDSP_TUNE_ENV(mf_env, "ACDSP_MVAVG_MFMA_MIN");   // A/B knob: the smallest window that takes them (0: never)
  // (same box, 512 objects x 2^20, frames of 1024: v_dot2 0.68 / MFMA 0.62 of the roofline at 9 taps, 0.57 / 0.62 at 11, 0.46 / 0.61 at 33, 0.29 / 0.53 at
  // 65; AC_WIN 0.54 / 0.58 at 9 taps -- profiles/r6_mvavg_mfma.txt)
  const int mf_min = mf_env ? atoi(mf_env) : (p.win_mode == 0 ? 9 : 11);
  if (p.frag && p.frag_nb > 0 && a.linear && !a.pk_n && !a.row_n && mf_min > 0 && p.taps >= mf_min) {
    a.frag = p.frag;
    a.kbias = (int32_t)(128 * p.frag_csum);
// (NR = 9: only REGION -- how far the right-edge patches reach into the image -- depends on it here)
#define ACDSP_MV_MF(CV_, EDGE_, MF_)                                                                                      \
  do {                                                                                                                      \
    if (a.run_ok) { hipLaunchKernelGGL((mv_avg_stream_kernel<9, true, CV_, EDGE_, false, MF_, true>), grid, dim3(256), 0, s, a); } \
    else { hipLaunchKernelGGL((mv_avg_stream_kernel<9, true, CV_, EDGE_, false, MF_, false>), grid, dim3(256), 0, s, a); }   \
  } while (0)
    if (p.frag_nb == 1) {
      if (a.cv32) { if (a.mode != 0) { ACDSP_MV_MF(true, true, 1); } else { ACDSP_MV_MF(true, false, 1); } }
      else { if (a.mode != 0) { ACDSP_MV_MF(false, true, 1); } else { ACDSP_MV_MF(false, false, 1); } }
    } else {
      if (a.cv32) { if (a.mode != 0) { ACDSP_MV_MF(true, true, 2); } else { ACDSP_MV_MF(true, false, 2); } }
      else { if (a.mode != 0) { ACDSP_MV_MF(false, true, 2); } else { ACDSP_MV_MF(false, false, 2); } }
    }
#undef ACDSP_MV_MF
    *mfma = true;
    return true;
  }
  const int nr = p.taps <= 9 ? 2 : (p.taps <= 17 ? 3 : (p.taps <= 25 ? 4 : (p.taps <= 33 ? 5 : (p.taps <= 49 ? 7 : 9))));   // taps <= 8 NR - 7
#define ACDSP_MV_LAUNCH2(NR_, LIN_, CV_)                                                                                           \
  if (a.pk_n) { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, true, true>), grid, dim3(256), 0, s, a); }               \
  else if (a.row_n && a.row_opf) { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, false, false, 0, false, 2>), grid, dim3(256), 0, s, a); } \
  else if (a.row_n) { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, true, false, 0, false, 1>), grid, dim3(256), 0, s, a); } \
  else if (a.run_ok) {                                                                                                             \
    if (a.mode != 0) { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, true, false, 0, true>), grid, dim3(256), 0, s, a); } \
    else { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, false, false, 0, true>), grid, dim3(256), 0, s, a); }          \
  }                                                                                                                                \
  else if (a.mode != 0) { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, true>), grid, dim3(256), 0, s, a); }             \
  else { hipLaunchKernelGGL((mv_avg_stream_kernel<NR_, LIN_, CV_, false>), grid, dim3(256), 0, s, a); }
#define ACDSP_MV_LAUNCH(NR_)                                                                                                       \
  if (a.linear) {                                                                                                                   \
    if (a.cv32) { ACDSP_MV_LAUNCH2(NR_, true, true) } else { ACDSP_MV_LAUNCH2(NR_, true, false) }                                  \
  } else {                                                                                                                          \
    if (a.cv32) { ACDSP_MV_LAUNCH2(NR_, false, true) } else { ACDSP_MV_LAUNCH2(NR_, false, false) }                                \
  }
  switch (nr) {
    case 2: ACDSP_MV_LAUNCH(2) break;
    case 3: ACDSP_MV_LAUNCH(3) break;
    case 4: ACDSP_MV_LAUNCH(4) break;
    case 5: ACDSP_MV_LAUNCH(5) break;
    case 7: ACDSP_MV_LAUNCH(7) break;
    default: ACDSP_MV_LAUNCH(9) break;
  }
#undef ACDSP_MV_LAUNCH2
#undef ACDSP_MV_LAUNCH
  return true;
}

// Toeplitz fragments of the matrix-core form (mv_avg_stream_kernel, MF instantiations).  Returns the K blocks (1 / 2; 0: the coefficient set
// or the window does not fit) and fills out[((m nb + b) 2 + plane) 64 + lane][4 dwords]: lane (row i = lane & 15, kg = lane >> 4) holds
// A[i][16 kg .. 16 kg + 15] of K block b, A[i][kk] = digit_plane(c[64 b + kk - sigma_m(i) - off]), sigma_m(i) = 8 (i >> 2) + (i & 3) + 4 m;
// c = 256 ch + cl with both digits signed bytes (balanced: cl = int8(c & 255)), so that every product is a signed i8 x i8 one.
int mv_avg_build_frags(const int64_t *c, int taps, int win_mode, uint32_t *out, int64_t *csum) {
  const int h = taps / 2, hb = win_mode == 0 ? 0 : 8 * ((h + 7) / 8), off = win_mode == 0 ? 0 : hb - h;
  if (taps > 65 || 31 + off + taps - 1 >= 128) { return 0; }
  const int nb = (31 + off + taps - 1 < 64) ? 1 : 2;
  int64_t sum = 0;
  for (int t = 0; t < taps; t++) {
    if (c[t] < -32768 || c[t] > 32639) { return 0; }   // the balanced high digit of 32640 .. 32767 is 128
    sum += c[t];
  }
  *csum = sum;
  for (int m = 0; m < 2; m++) {
    for (int b = 0; b < nb; b++) {
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, kg = lane >> 4, sig = 8 * (i >> 2) + (i & 3) + 4 * m;
        for (int dw = 0; dw < 4; dw++) {
          uint32_t wl = 0, wh = 0;
          for (int bj = 0; bj < 4; bj++) {
            const int t = 64 * b + 16 * kg + 4 * dw + bj - sig - off;
            const int64_t cv = (t >= 0 && t < taps) ? c[t] : 0;
            const int cl = (int)(int8_t)(uint8_t)(cv & 0xFF), ch = (int)((cv - cl) >> 8);
            wl |= (uint32_t)(uint8_t)cl << (8 * bj);
            wh |= (uint32_t)(uint8_t)ch << (8 * bj);
          }
          out[(((size_t)(m * nb + b) * 2 + 0) * 64 + lane) * 4 + dw] = wl;
          out[(((size_t)(m * nb + b) * 2 + 1) * 64 + lane) * 4 + dw] = wh;
        }
      }
    }
  }
  return nb;
}

hipError_t launch_mv_avg(const MvAvgParams &p, hipStream_t s, int *path) {
  if (p.out_per_frame <= 0 || p.n_frames <= 0) { return hipSuccess; }
  bool mfma = false;
  if (try_stream(p, s, &mfma)) { *path = mfma ? 4 : 2; return hipGetLastError(); }
  if (try_w32(p, s)) { *path = 3; return hipGetLastError(); }
  *path = p.fast ? 1 : 0;
  const int64_t pairs = (int64_t)p.n_obj * p.n_frames;
  dim3 grid((unsigned)((p.out_per_frame + kTile - 1) / kTile), (unsigned)(pairs < 65535 ? pairs : 65535));
  const size_t lds = (size_t)(kTile + 2 * p.taps - 1) * sizeof(int64_t);
  if (p.fast) { hipLaunchKernelGGL(mv_avg_kernel<true>, grid, dim3(kTile), lds, s, p); }
  else { hipLaunchKernelGGL(mv_avg_kernel<false>, grid, dim3(kTile), lds, s, p); }
  return hipGetLastError();
}

}  // namespace acdsp
