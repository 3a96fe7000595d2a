// fir_kernels.hpp -- launch interface between the C-ABI layer (engine.hip) and the FIR kernels.
#pragma once
#include <cstdlib>
#include <vector>

#include "acdsp_dev.hpp"

namespace acdsp {

// Everything a FIR launch needs; passed by value as the kernel argument.
struct FirParams {
  int32_t n_taps, ftype, n_ch, coeffs_per_channel;
  DFmt in, cf, acc, out;
  int32_t in_eb, out_eb;       // container bytes of IN / OUT
  int32_t hl;                  // history samples kept per channel (>= n_taps-1, multiple of 32)
  int32_t use_rt;              // TRANSPOSED with carried partial sums (reg_trans state)
  int32_t lossless_shift;      // F_acc - F_in - F_coeff (>= 0 on the lossless paths)
  int64_t in_stride, out_stride, n;
  const void *x;               // input samples  [n_ch][in_stride]  (IN containers)
  void *y;                     // output samples [n_ch][out_stride] (OUT containers)
  const void *hist;            // [n_ch][hl] IN containers: the hl samples before t = 0
  const int64_t *coeffs;       // [n_sets][n_taps] raw words
  const int64_t *rt;           // [n_ch][n_taps] ACC raw words (use_rt)
  void *hist_next;             // small calls: the exact-order kernels write the next history themselves (one launch per call); else null
  int64_t t_begin;             // exact-order kernels: outputs [t_begin, n) only (the ragged rest behind a matrix-core launch); normally 0
  int32_t in_flip;             // int8 matrix-core kernels: the rows hold UNSIGNED 16-bit samples, to be read as x - 32768 (MfmaArgs::hi_xor); else 0
};

// Exact per-tap emulation in the reference's loop order (any Q/O, any widths <= 64).
hipError_t launch_fir_generic(const FirParams &p, hipStream_t s);
// Lossless accumulator + AC_WRAP: integer dot product mod 2^64, one requant at the end.
hipError_t launch_fir_lossless64(const FirParams &p, hipStream_t s);
// class B on 16-bit types (lossy accumulator, AC_TRN / AC_RND into AC_WRAP, order-free ftypes): fir_generic.hip
bool fir_lossy_fast_ok(const FirParams &p);
hipError_t launch_fir_lossy(const FirParams &p, hipStream_t s);
// class C on 16-bit types with a saturating accumulator of up to 32 bits, in the reference's tap order: fir_generic.hip
bool fir_satacc_fast_ok(const FirParams &p);
hipError_t launch_fir_satacc(const FirParams &p, hipStream_t s);
// State carry: history ring (all ftypes but TRANSPOSED-with-rt) and reg_trans partial sums.
hipError_t launch_fir_hist_update(const FirParams &p, void *hist_next, hipStream_t s);
hipError_t launch_fir_rt_update(const FirParams &p, int64_t *rt_next, hipStream_t s);

// int8-split MFMA Toeplitz path (fir_mfma.hip)
struct FirMfmaPlan {
  int32_t nb;                  // 32-sample K blocks per output block = ceil((n_taps-1)/32) + 1
  uint64_t hi_mask, lo_mask;   // bit b set: Toeplitz block b of the hi / lo coefficient plane is non-zero
  int64_t corr;                // 128 * sum(c): undoes the signed re-bias of the low input byte
  int64_t sum_abs;             // sum |c|            (bounds |y|  <= 32768 * sum_abs)
  int64_t sum_abs_hi;          // sum |high byte|    (bounds |S(ch,.)| <= 128 * sum_abs_hi)
  int64_t sum_abs_lo;          // sum |low byte|
};
// Builds the per-lane A fragments (host side) into frag[2][nb][64][4] dwords; returns false if the
// coefficient set cannot be split into two signed bytes per tap.
bool fir_mfma_build_fragments(const int64_t *coeffs, int n_taps, FirMfmaPlan *plan, uint32_t *frag /* host */);
int fir_mfma_plan_blocks_padded(int n_taps);   // the same with the padding applied regardless of the ACDSP_NO_MID knob (state geometry)
int fir_mfma_plan_blocks(int n_taps);   // K-blocks of the plan (258 - 513 taps: padded to an odd count, fir_mfma.hip)
int fir_mfma_max_blocks();       // K-blocks the MFMA path can take at all (A fragments in LDS): 33 -> 1025 taps
int fir_mfma_max_reg_blocks();   // ... with the A fragments register-resident (needed for per-channel coefficient sets): 9
// One wave = one channel x a time chunk; fragments and corr are per coefficient set:
// d_frag[n_sets][2][nb][64][4], d_corr[n_sets]; `plan` carries the worst-case bounds over all sets.
int fir_mfma_issued_per_step(const FirParams &p, const FirMfmaPlan &plan);
int fir_mfma_epilogue_class(const FirParams &p, const FirMfmaPlan &plan);   // 0 generic, 1 / 2 the 32-bit classes, 3 wide, 4 the 64-bit branch-free one
bool fir_mfma_register_resident(const FirParams &p, const FirMfmaPlan &plan);   // fragments in registers: per-channel coefficient sets allowed
hipError_t launch_fir_mfma(const FirParams &p, const FirMfmaPlan &plan, int frag_per_channel, const uint32_t *d_frag,
                           const int64_t *d_corr, hipStream_t s);

// Generalised exact FIR on the matrix cores (fir_gen.hip): wide inputs / coefficients, decimation.
// FirParams::ftype values of the ac_fir_reg_share cores (ascending MAC order, anti-symmetric folds:
// reference ac_fir_reg_share.h:136-260); the const/load/prog cores use the public ACDSP_* values.
enum { kRsShiftReg = 16, kRsFoldEven = 17, kRsFoldEvenAnti = 18, kRsFoldOdd = 19, kRsFoldOddAnti = 20 };

struct FirGenPlan {
  int32_t pc, nb, off, R;   // coefficient byte planes, 64-sample K blocks, T_n - W_n, decimation
  int64_t sum_h;            // sum of the taps mod 2^64 (re-bias correction)
  int64_t sum_abs_h;        // sum of |taps|, saturating at 2^62 (bounds of the 32-bit limb epilogues)
  int64_t dig_abs[3];       // sum over the taps of |digit q| (bounds of the int32 plane accumulators)
};
// y[m] = sum_k h[k] x[first + m R - k] mod 2^64.  Builds the A fragments for first % 16 == first_mod16;
// false if the taps need more than 3 byte planes or more than 8 K blocks.
bool fir_gen_plan(const int64_t *h, int n_taps, int R, int first_mod16, FirGenPlan *pl, std::vector<uint32_t> *frag);
// out_mode 0: FIR class A epilogue (p.lossless_shift, p.acc wrap, requant to p.out);
// out_mode 1: CIC epilogue (wrap to w_int, requant from p.in.F to p.out).  p.n = inputs, p.hist holds >= off samples.
// Class B on the ring kernel (lossy accumulator, AC_TRN / AC_RND into AC_WRAP; fir_gen.hip: LZ instantiations).  pl / d_frag are the
// plan of the EFFECTIVE tap vector (the exact sum), this struct the per-tap residues.
struct FirLossyPlan {
  int32_t s;                           // F_in + F_c - F_acc, 1 .. 15
  int32_t flush;                       // iterations after which the packed 16-bit sums are emptied into the 32-bit totals
  int32_t neg, var_y;                  // folded pairs: difference instead of sum (ac_fir_reg_share's anti-symmetric cores); even tap count
  int32_t p_n, p_woff, p_voff;         // pair loop: iterations (two pairs each), window offsets in samples relative to the output's own sample
  int32_t s_n, s_woff;                 // single-tap loop: iterations (two taps each), window offset
  uint32_t h2, m2;                     // rounding constant / mask of the dropped bits in both 16-bit fields
  int64_t k;                           // slots * h
  int32_t acc_bits;                    // bits (sign included) the accumulator VALUE can reach, <= W_acc: bounds the OUT_TYPE rounding add
  const uint32_t *d_tab;               // device [kLossyTabWords]: per iteration (c_S, c_D) = c mod 2^s in both 16-bit fields; pair loop first
};
constexpr int kLossyTabWords = 256;
// Slot tables of the residue loops (fir_gen.hip): taps [0, n_pair) fold with their mirror N-1-i, taps [single0, single0 + n_single) stand alone;
// pl = plan of the effective taps (its window offset fixes the sample parities).  false: more slots than the table / the 16-bit sums hold.
bool fir_gen_lossy_table(const FirGenPlan &pl, const int64_t *coeffs, int n_taps, int n_pair, int n_single, int single0, int neg, int s, bool rnd,
                         FirLossyPlan *out, std::vector<uint32_t> *tab);
// are this handle's (formats, plan) compiled as a class-B ring shape?  (decided at set_coeffs time: acdsp_fir_path reports it)
bool fir_gen_lossy_shape_ok(const FirParams &p, const FirGenPlan &pl, int acc_bits);   // acc_bits: FirLossyPlan::acc_bits
// lz != nullptr: class B -- only complete chunks run here; *covered = outputs written (the caller runs the exact-order kernel on the rest)
hipError_t launch_fir_gen(const FirParams &p, const FirGenPlan &pl, const uint32_t *d_frag, int out_mode, int w_int,
                          int64_t first, int64_t n_out, hipStream_t s, const FirLossyPlan *lz = nullptr, int64_t *covered = nullptr);

// Fused decimator -> FIR cascade on the matrix cores (fir_gen.hip).  pa = stage A as for launch_fir_gen with
// out_mode 1 (history of >= 256*R + off + 16 samples), pb = stage B formats + output buffer (int32 containers).
// hipErrorNotSupported: shapes outside the compiled ones (run the two kernels instead).
// A second stream of a handle (round 5) for the small kernels of a call that neither feed nor follow its main kernel -- the head / tail of an
// ac_poly_intr call: a few thousand waves, latency-bound, and on the caller's stream they queued behind the main kernel (4.7 % of the bench row).
// fork() makes the side stream wait for everything enqueued on the caller's stream so far, join() makes the caller's stream wait for the side
// stream: an ordinary fork / join, legal under stream capture.  ACDSP_NO_SIDE_STREAM=1: everything on the caller's stream (A/B knob).
// (Same-box A/B, two passes: poly_intr 0.899 -> 0.846 ms with the state kernel on the side stream too; the fused DDC's edge-chunk kernel, 143 us behind a 1.73 ms main kernel, gained nothing
// from the same treatment -- 1.751 / 1.766 against 1.752 / 1.759 ms -- and the head / tail of a CIC interpolator call (2 x 11 us beside
// 3.2 ms) lost 0.3 - 0.8 %: both stay on the caller's stream.)
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipError_t ensure() {
    hipError_t e = hipSuccess;
    if (!s) { e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
    if (e == hipSuccess && !ev_fork) { e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming); }
    if (e == hipSuccess && !ev_join) { e = hipEventCreateWithFlags(&ev_join, hipEventDisableTiming); }
    return e;
  }
  static bool enabled() { static const bool off = getenv("ACDSP_NO_SIDE_STREAM") != nullptr; return !off; }
  hipError_t fork(hipStream_t main) {
    hipError_t e = ensure();
    if (e == hipSuccess) { e = hipEventRecord(ev_fork, main); }
    if (e == hipSuccess) { e = hipStreamWaitEvent(s, ev_fork, 0); }
    return e;
  }
  hipError_t join(hipStream_t main) {
    hipError_t e = hipEventRecord(ev_join, s);
    if (e == hipSuccess) { e = hipStreamWaitEvent(main, ev_join, 0); }
    return e;
  }
  void destroy() {
    if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
    if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
    if (s) { (void)hipStreamDestroy(s); s = nullptr; }
  }
};

hipError_t launch_cascade(const FirParams &pa, const FirGenPlan &pla, const uint32_t *d_fragA, int w_int, int64_t first,
                          const FirParams &pb, const FirGenPlan &plb, const uint32_t *d_fragB, int64_t n_out, hipStream_t s);

// Interpolating polyphase FIR on the matrix cores (fir_up.hip): z[n L + j] = sum_k E_j[k] x[n - k] mod 2^64.
struct FirUpPlan {
  int32_t L, nt, pc, nb, hs;   // up factor, taps per phase, coefficient digit planes, 32-sample K blocks, history slots (2 nb - 1)
};
// E = [L][nt] taps; builds the Toeplitz fragments [RG][nb][3][64][4] and the per-phase re-bias correction [L] for px input
// byte planes.  false: L does not divide 32, more than 3 digits per tap, or a window of more than 2 K blocks (nt > 49).
bool fir_up_plan(const int64_t *E, int L, int nt, int px, FirUpPlan *pl, std::vector<uint32_t> *frag, std::vector<int64_t> *corr);
bool fir_up_shape_ok(int in_eb, int px, int nb, int L, int out_eb);   // is this shape compiled in?
// mode 0: poly_intr epilogue ((V << p.lossless_shift) >> sh_j, p.acc.F -> p.out);  mode 1: CIC epilogue (wrap to w_int, p.in.F -> p.out).
// Covers input slots [slot0, slot0 + 32 n_steps) (16 samples each, slot0 >= pl.hs, rows 16-byte aligned); output element of
// (input n, phase j) = n L + j + out_off (8-byte aligned runs).  hipErrorNotSupported: shape not compiled in.
hipError_t launch_fir_up(const FirParams &p, const FirUpPlan &pl, int px, const uint32_t *d_frag, const int64_t *d_corr, int mode, int w_int,
                         int out_simple, uint32_t sh_mask, int64_t max_abs_v, int64_t slot0, int64_t n_steps, int64_t out_off, hipStream_t s);

// Polyphase interpolator (polyintr.hip): ftype 0 FOLD_EVEN, 1 FOLD_ODD, 2 FOLD_ANTI (ac_poly_intr.h:71)
struct PolyIntrParams {
  int32_t n_taps, coeff_sz, ifac, ftype, n_ch;
  DFmt in, cf, acc, out;
  int32_t in_eb, out_eb, hl;
  int32_t skip;               // 1: the stream's very first sample is in this call and emits nothing (folded cores)
  int32_t lossless, lossless_shift;   // exact-accumulation class: int64 dot products, shift = F_acc - F_in - F_coeff
  int64_t in_stride, out_stride, n, n_out;   // n inputs -> n_out outputs per channel
  int64_t o_begin, o_end;     // outputs [o_begin, o_end) of the call are produced by this launch (the rest: fir_up.hip)
  const void *x; void *y; const void *hist;
  const int64_t *coeffs;      // [coeff_sz]
  const uint8_t *sign, *corr; // [ifac]
  const int64_t *saved;       // [n_ch][ifac] sums of the previous call's last sample (folded cores)
};
hipError_t launch_polyintr(const PolyIntrParams &p, int64_t *saved_next, hipStream_t s);

// Integrate-and-dump (intg_dump.hip).  Block b covers rounds [blk_off[b], blk_off[b] + blk_rounds[b]) of the interleaved
// stream; blk_out[b] = index of its output group or -1 (no dump); blk_chain[b] = first block of its carry chain.
struct IntgDumpParams {
  int32_t chn, n_obj, n_blocks;
  int32_t lossless;           // ACC_TYPE is AC_WRAP with F_acc >= F_in: integer sums mod 2^W
  int32_t tile_ok;            // lossless, every block of the call dumps and the handle carries no undumped sum in
  int64_t uni_rounds;         // > 0: tile_ok and every block has this many rounds (streaming kernel)
  DFmt in, acc, out;
  int32_t in_eb, out_eb;
  int64_t in_stride, out_stride;
  const void *x; void *y;
  const int64_t *temp;        // [n_obj][chn] ACC raw words carried in
  const int64_t *blk_off, *blk_rounds, *blk_out;
  const int32_t *blk_chain;
};
// *temp_written = false: the call took a tile_ok kernel -- nothing carried in, every block dumped -- and left temp_next alone: the handle's
// current temp[] is all zero and stays the state (round 5: a zeroing kernel per call was 1.4 % of the bench row)
// *path: 0 exact-order kernel, 1 LDS-tiled kernel, 2 streaming kernels, 3 matrix-core kernel (channel counts that do not divide a 16-byte load)
hipError_t launch_intg_dump(const IntgDumpParams &p, int64_t *temp_next, hipStream_t s, bool *temp_written, int *path);

// Moving average (mv_avg.hip).  win_mode 0 AC_WIN, 1 AC_MIRROR, 2 AC_CLIP; rows = objects, frames of n_sample inputs back to back
struct MvAvgParams {
  int32_t taps, win_mode, n_obj;
  int32_t force_generic;      // ACDSP_FLAG_FORCE_GENERIC: exact-order kernel only
  int32_t fast;               // AC_TRN / AC_RND + AC_WRAP accumulator, F_coeff >= 0, products within 63 bits: order-free int64 sums
  DFmt in, cf, acc, out;
  int32_t in_eb, out_eb;
  int64_t n_sample, n_frames, out_per_frame, in_stride, out_stride;
  const void *x; void *y;
  const int64_t *coeffs;      // [taps]
  const int64_t *h_coeffs;    // host copy (the streaming kernel takes its coefficients as kernel arguments)
  const uint32_t *frag;       // device: Toeplitz fragments of the matrix-core form (mv_avg_build_frags), frag_nb K blocks (0: none), frag_csum = sum of the coefficients
  int32_t frag_nb;
  int64_t frag_csum;
};
hipError_t launch_mv_avg(const MvAvgParams &p, hipStream_t s, int *path);   // *path: 0 exact order, 1 int64 sums, 2 streaming kernel, 3 32-bit samples, 4 streaming kernel on the matrix cores
constexpr int kMvAvgFragWords = 2 * 2 * 2 * 64 * 4;   // [m][b][plane][lane][4]
int mv_avg_build_frags(const int64_t *c, int taps, int win_mode, uint32_t *out, int64_t *csum);

// Sets the calling thread's acdsp_last_error() message and returns `code` (engine.hip); for the layers above the engine (node.hip).
int set_error(int code, const char *msg);

// Measurement kernels of acdsp_diag_* (diag.hip): a plain 16-byte-per-thread copy, and the stream + issued-MFMA envelope of a FIR row.
hipError_t launch_diag_copy(const void *src, void *dst, int64_t bytes, hipStream_t s);
// placement probe: a bare stream reading one block and writing the other at their byte ratio (diag.hip)
hipError_t launch_diag_mix(const void *src, int64_t src_bytes, void *dst, int64_t dst_bytes, hipStream_t s);
// d_frag: six Toeplitz fragments [4 low-plane blocks][2 high-plane blocks] x 64 lanes x 16 bytes; hipErrorInvalidValue: count not compiled
bool diag_envelope_compiled(int mfma, int mfma_hi);
hipError_t launch_diag_clock(float *d_mhz, int n_blocks, hipStream_t s);   // shader clock in MHz per block (diag.hip)
hipError_t launch_diag_envelope(const uint32_t *d_frag, const void *x, void *y, int64_t bytes, int mfma, int mfma_hi, hipStream_t s);
// ... in the copy kernel's geometry (one 16-byte element per thread; stand-in A operands made of the loaded bytes)
// d_frag != nullptr: four elements per thread and the real fragments (loaded once per wave)
hipError_t launch_diag_envelope_copygeom(const uint32_t *d_frag, const void *x, void *y, int64_t bytes, int mfma, int mfma_hi, hipStream_t s);

// Polyphase decimator, exact per-MAC order (polydec.hip); p.coeffs = STR_COEFF_TYPE array [ntaps*df], p.n = inputs used
hipError_t launch_polydec_generic(const FirParams &p, int ntaps, int df, int64_t n_out, hipStream_t s);

}  // namespace acdsp
