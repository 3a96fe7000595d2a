/* acdsp_oracle.h -- CPU restatement of the hlslibs/ac_dsp FIR/CIC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ac_dsp_amd/ or include/ may call,
 * link or import this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported CPU baseline.
 *
 * It follows the reference's per-sample algorithm line by line on plain
 * integers (raw two's-complement words, value = raw * 2^-(W-I)); every
 * function cites the reference file:line it restates.  The ac_fixed
 * arithmetic itself lives in hlslibs/ac_types, an un-vendored dependency that
 * is absent from /root/reference (no version pin in the tree; ac_dsp release
 * v2026.1.1 pairs with the ac_types of the same Catapult 2026.1 train), so the
 * quantisation/overflow rules are restated from the published AC Datatypes
 * semantics.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - CIC decimator / interpolator: pinned EXACTLY by the reference's own
 *     vectors tests/ac_cic_{dec,intr}_full_{input,ref}.txt (tests/golden/).
 *   - FIR: the reference's three *_ref.txt are MATLAB doubles compared at
 *     SQNR >= 60 dB (tests/rtest_ac_fir_const_coeffs.cpp:168-194); the oracle
 *     reproduces them at 84-90 dB.  Bit-level FIR behaviour with default
 *     AC_TRN/AC_WRAP types is pinned only through that and through the CIC
 *     vectors (same wrap add / cast rules).
 *   - Non-default Q/O modes, saturating accumulators, ac_poly_dec,
 *     ac_poly_intr, ac_intg_dump and ac_fir_reg_share (no reference test or
 *     vector exists for any of them):
 *     PARITY UNPINNED by any reference vector; cross-checked against the
 *     independent template implementation in include/ac_types/ac_fixed.h only.
 */
#ifndef ACDSP_ORACLE_H
#define ACDSP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t W, I, S, Q, O; } orc_fmt_t;

enum { ORC_TRN = 0, ORC_RND, ORC_TRN_ZERO, ORC_RND_ZERO, ORC_RND_INF, ORC_RND_MIN_INF, ORC_RND_CONV, ORC_RND_CONV_ODD };
enum { ORC_WRAP = 0, ORC_SAT, ORC_SAT_ZERO, ORC_SAT_SYM };
/* FTYPE enum: reference include/ac_dsp/ac_fir_const_coeffs.h:96 */
enum { ORC_SHIFT_REG = 0, ORC_ROTATE_SHIFT, ORC_C_BUFF, ORC_FOLD_EVEN, ORC_FOLD_ODD, ORC_TRANSPOSED, ORC_FOLD_EVEN_ANTI, ORC_FOLD_ODD_ANTI };

/* Convert the exact value x * 2^-f_src (x given as hi:lo of a 128-bit two's
 * complement integer) into format *dst; returns the raw word. */
int64_t orc_requant128(int64_t x_hi, uint64_t x_lo, int32_t f_src, const orc_fmt_t *dst);
int64_t orc_requant(int64_t x, int32_t f_src, const orc_fmt_t *dst);
/* double -> fixed, as `ac_fixed<...> v = d;` */
int64_t orc_from_double(double d, const orc_fmt_t *dst);

/* ---- FIR: one object per channel, like one ac_fir_* instance ---- */
typedef struct orc_fir orc_fir_t;
orc_fir_t *orc_fir_new(int32_t n_taps, int32_t ftype, const orc_fmt_t *in, const orc_fmt_t *coeff,
                       const orc_fmt_t *acc, const orc_fmt_t *out);
void orc_fir_free(orc_fir_t *f);
void orc_fir_reset(orc_fir_t *f);
/* One sample through the core selected by ftype; returns 0, or -1 for the
 * FOLD_*_ANTI values that the reference's run() does not handle. */
int32_t orc_fir_step(orc_fir_t *f, const int64_t *coeffs, int64_t x, int64_t *y);
/* Drain n samples (the while(available) loop of run()). */
int32_t orc_fir_run(orc_fir_t *f, const int64_t *coeffs, const int64_t *x, int64_t n, int64_t *y);
/* Many independent channels, [channel][time] layout with the given strides
 * (in elements); coeffs is [n_taps] shared or [n_ch][n_taps]. State is kept in
 * the array of objects fs[n_ch]. */
int32_t orc_fir_run_many(orc_fir_t **fs, int64_t n_ch, const int64_t *coeffs, int32_t coeffs_per_channel,
                         const int64_t *x, int64_t x_stride, int64_t n, int64_t *y, int64_t y_stride);

/* ac_fir_reg_share<N_TAPS, IN, OUT, COEFF, ACC, MEM_WORD_WIDTH, BLK_SZ, BLK_OFFSET, ftype>::run (row f1 of SURVEY 8:
 * reference include/ac_dsp/ac_fir_reg_share.h:120-306).  Uses an orc_fir_t made with the FTYPE value; valid ftypes
 * SHIFT_REG, FOLD_EVEN, FOLD_EVEN_ANTI, FOLD_ODD, FOLD_ODD_ANTI.  `c` = the coeffs[N_TAPS] array as run() receives it. */
int32_t orc_fir_reg_share_step(orc_fir_t *f, const int64_t *c, int32_t mem_word_width, int32_t blk_sz, int32_t blk_offset,
                               int64_t x, int64_t *y);
int32_t orc_fir_reg_share_run(orc_fir_t *f, const int64_t *c, int32_t mem_word_width, int32_t blk_sz, int32_t blk_offset,
                              const int64_t *x, int64_t n, int64_t *y);
int64_t orc_fir_reg_share_delay_line(const orc_fir_t *f);

/* ---- CIC ---- */
typedef struct orc_cic orc_cic_t;
/* interp = 0: ac_cic_dec_full, 1: ac_cic_intr_full */
orc_cic_t *orc_cic_new(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, const orc_fmt_t *out);
void orc_cic_free(orc_cic_t *c);
/* Intermediate (lossless) type the reference derives; W,I,S filled in. */
int32_t orc_cic_int_type(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, orc_fmt_t *it);
/* One run() call: consumes n_in inputs, writes up to cap outputs, returns the
 * number produced (or -1 if cap is too small). */
int64_t orc_cic_run(orc_cic_t *c, const int64_t *x, int64_t n_in, int64_t *y, int64_t cap);

/* ---- polyphase decimator (row f2 of SURVEY 8: reference include/ac_dsp/ac_poly_dec.h:82-138) ---- */
typedef struct orc_polydec orc_polydec_t;
orc_polydec_t *orc_polydec_new(int32_t ntaps, int32_t df, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                               const orc_fmt_t *out);
void orc_polydec_free(orc_polydec_t *f);
/* One run() call: consumes floor(n_in / DF) * DF inputs (the reference loops `while (available(DF))`), writes
 * one output per group; coeffs is the STR_COEFF_TYPE array [NTAPS*DF].  Returns the number of outputs. */
int64_t orc_polydec_run(orc_polydec_t *f, const int64_t *coeffs, const int64_t *x, int64_t n_in, int64_t *y);

/* ---- polyphase interpolator (row f2 of SURVEY 8: reference include/ac_dsp/ac_poly_intr.h:104-320) ----
 * ftype 0 FOLD_EVEN, 1 FOLD_ODD, 2 FOLD_ANTI (the enum of ac_poly_intr.h:71).  One step = one input sample = one call of
 * the selected core; coeffs[COEFFSZ], sign[IF], corr[IF] are the members of the coefficient / control structs. */
typedef struct orc_polyintr orc_polyintr_t;
orc_polyintr_t *orc_polyintr_new(int32_t ntaps, int32_t coeffsz, int32_t ifac, int32_t ftype, const orc_fmt_t *in,
                                 const orc_fmt_t *coeff, const orc_fmt_t *acc, const orc_fmt_t *out);
void orc_polyintr_free(orc_polyintr_t *f);
int64_t orc_polyintr_step(orc_polyintr_t *f, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr, int64_t x, int64_t *y);
int64_t orc_polyintr_run(orc_polyintr_t *f, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr, const int64_t *x,
                         int64_t n_in, int64_t *y);

/* ---- integrate and dump (row f4 of SURVEY 8: reference include/ac_dsp/ac_intg_dump.h:77-151) ----
 * temp[CHN] is the object's state (zeros after construction); n_sample[n_blocks] are the N_TYPE words read at the start
 * of each block; x = the interleaved input stream.  Returns the outputs written; *used = input samples consumed. */
int64_t orc_intg_dump_run(int64_t *temp, int32_t ns, int32_t chn, const orc_fmt_t *in, const orc_fmt_t *acc, const orc_fmt_t *out,
                          const int64_t *n_sample, int64_t n_blocks, const int64_t *x, int64_t *y, int64_t *used);

/* ---- moving average (row f4 of SURVEY 8: reference include/ac_dsp/ac_mv_avg.h:93-196) ----
 * PARITY UNPINNED for the window class: ac_mv_avg is built on ac_window_1d_flag (ac_types' <ac_window.h>), which is absent
 * from the image and from the reference tree; its boundary behaviour is restated from the documented meaning of the modes
 * (AC_WIN none, AC_CLIP replicate the edge sample, AC_MIRROR reflect about it) and from how ac_mv_avg.h drives it.  The
 * MAC loop, its cast and its order are the reference's (:113-121). */
#define ORC_MVAVG_MAX_TAPS 1025
enum { ORC_WIN_PLAIN = 0, ORC_WIN_MIRROR = 1, ORC_WIN_CLIP = 2 };
int64_t orc_mv_avg_run(int32_t taps, int32_t win_mode, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                       const orc_fmt_t *out, const int64_t *c, const int64_t *x, int64_t n_sample, int64_t n_frames, int64_t *y);

/* ---- types wider than 64 bits (acdsp_oracle_wide.cpp): the FIR cores and both CIC directions again, on 128-bit raw words ----
 * IN / COEFF <= 64 bits, ACC / OUT / INT_TYPE <= 128 bits, 256-bit exact intermediates.  Words are little-endian pairs.
 * These functions also accept formats of <= 64 bits; there they must agree with orc_fir_run / orc_cic_run word for word
 * (tests/test_wide_cpu.py). */
typedef struct { uint64_t lo; int64_t hi; } orcw_word_t;
typedef struct orcw_fir orcw_fir_t;
typedef struct orcw_cic orcw_cic_t;
orcw_fir_t *orcw_fir_new(int32_t n_taps, int32_t ftype, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                         const orc_fmt_t *out);
void orcw_fir_free(orcw_fir_t *f);
int32_t orcw_fir_run(orcw_fir_t *f, const int64_t *coeffs, const int64_t *x, int64_t n, orcw_word_t *y);
orcw_cic_t *orcw_cic_new(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, const orc_fmt_t *out);
void orcw_cic_free(orcw_cic_t *c);
int64_t orcw_cic_run(orcw_cic_t *c, const int64_t *x, int64_t n_in, orcw_word_t *y, int64_t cap);
/* exact x * 2^-f_src (x: 256-bit little-endian two's-complement integer) -> raw word of *dst; -1 for an unsupported format */
int32_t orcw_requant(const uint64_t x[4], int32_t f_src, const orc_fmt_t *dst, orcw_word_t *out);

/* ---- synthetic stimulus shared with the GPU generator ---- */
uint64_t orc_splitmix64(uint64_t seed, uint64_t index);
/* raw sample for (channel, t): low `bits` bits of the hash, sign-extended */
int64_t orc_stimulus(uint64_t seed, uint64_t ch, uint64_t t, int32_t bits);

#ifdef __cplusplus
}
#endif
#endif
