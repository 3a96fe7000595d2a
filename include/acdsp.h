/* acdsp.h -- C ABI of the MI355X fixed-point streaming-filter engine (libacdsp.so).
 *
 * hlslibs/ac_dsp has no FFI/plugin layer: its boundary is the C++ class-template
 * API (`ac_fir_*_coeffs::run`, `ac_cic_*_full::run`).  The drop-in headers in
 * include/ac_dsp/ keep those templates and call the entry points below, which
 * replace the reference's per-sample core loops:
 *
 *   acdsp_fir_*   <- fir_{const,load,prog}_coeffs_core member functions
 *                    reference include/ac_dsp/ac_fir_const_coeffs.h:153-296,
 *                    ac_fir_load_coeffs.h:145-278, ac_fir_prog_coeffs.h:110-247
 *                    driven by run(): ac_fir_const_coeffs.h:321-355,
 *                    ac_fir_load_coeffs.h:320-365, ac_fir_prog_coeffs.h:277-303
 *   acdsp_cic_*   <- ac_cic_full_core_intg / _diff, ac_cic_full_core.h:80-160,198-255
 *                    driven by run(): ac_cic_dec_full.h:163-222, ac_cic_intr_full.h:150-215
 *
 * Data model: every sample is the raw W-bit two's-complement word of an
 * ac_fixed<W,I,S,Q,O> value (value = raw * 2^-(W-I)), stored sign-/zero-extended
 * in the smallest of int16/int32/int64 that holds W bits (acdsp_elem_bytes).
 * Streams are laid out [channel][time]: sample t of channel c is element
 * c*stride + t.  Channels are independent filter objects (one reference object
 * each); all filter state lives in the handle and carries across run() calls
 * exactly like the reference's object members.
 *
 * There is no CPU fallback: every entry point runs HIP kernels on gfx950 and
 * fails with ACDSP_ENODEVICE / ACDSP_EHIP otherwise.
 */
#ifndef ACDSP_H
#define ACDSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACDSP_ABI_VERSION 1

/* ac_fixed<W,I,S,Q,O> */
typedef struct { int32_t W, I, S, Q, O; } acdsp_fmt_t;

enum { ACDSP_TRN = 0, ACDSP_RND, ACDSP_TRN_ZERO, ACDSP_RND_ZERO, ACDSP_RND_INF, ACDSP_RND_MIN_INF, ACDSP_RND_CONV, ACDSP_RND_CONV_ODD };
enum { ACDSP_WRAP = 0, ACDSP_SAT, ACDSP_SAT_ZERO, ACDSP_SAT_SYM };
/* FTYPE, reference ac_fir_const_coeffs.h:96.  For the const/load/prog classes the *_ANTI values are rejected:
 * their run() has no branch for them and writes an unassigned value; ac_fir_reg_share gives them a meaning. */
enum { ACDSP_SHIFT_REG = 0, ACDSP_ROTATE_SHIFT, ACDSP_C_BUFF, ACDSP_FOLD_EVEN, ACDSP_FOLD_ODD, ACDSP_TRANSPOSED, ACDSP_FOLD_EVEN_ANTI, ACDSP_FOLD_ODD_ANTI };
enum { ACDSP_FIR_CONST = 0, ACDSP_FIR_LOAD = 1, ACDSP_FIR_PROG = 2,
       /* ac_fir_reg_share (reference ac_fir_reg_share.h:101-310): ftype is one of SHIFT_REG, FOLD_EVEN, FOLD_EVEN_ANTI,
        * FOLD_ODD, FOLD_ODD_ANTI (run() has no branch for the others, :288-306); the MAC loops walk the taps in ascending
        * order (:136-260).  Coefficients are handed over in TAP order: the header resolves the class's coefficient-memory
        * addressing (MEM_WORD_WIDTH / BLK_SZ / BLK_OFFSET) before acdsp_fir_set_coeffs. */
       ACDSP_FIR_REG_SHARE = 3 };

enum {
  ACDSP_OK = 0,
  ACDSP_EINVAL = 1,       /* bad argument / descriptor */
  ACDSP_EUNSUPPORTED = 2, /* valid in the reference but outside this engine's limits */
  ACDSP_EHIP = 3,         /* a HIP call failed */
  ACDSP_ENODEVICE = 4,    /* no gfx950 device */
  ACDSP_ESTATE = 5        /* call sequence error (e.g. run before coefficients are set) */
};

/* flags */
#define ACDSP_FLAG_FORCE_GENERIC 1 /* never pick the MFMA / fast kernels (parity tests) */

/* which kernel family a FIR handle resolved to (acdsp_fir_path) */
enum { ACDSP_PATH_GENERIC = 0, ACDSP_PATH_LOSSLESS64 = 1, ACDSP_PATH_MFMA_I8 = 2, ACDSP_PATH_MFMA_GEN = 3,
       ACDSP_PATH_WIDE = 4 /* a format wider than 64 bits: exact-order kernels on 128-bit words, 256-bit intermediates */,
       ACDSP_PATH_MFMA_LOSSY = 5 /* lossy wrapping accumulator (per-tap AC_TRN / AC_RND): exact sum on the matrix cores minus the dropped bits */,
       ACDSP_PATH_CIC_2STAGE = 6 /* ac_cic_dec_full with R = R1 R2: FIR identity of rate R1 on the matrix cores, integrators / combs of rate R2 behind it, one launch */ };
/* finer: the kernel family inside ACDSP_PATH_GENERIC (acdsp_fir_kernel_class; the other values equal the path) */
enum { ACDSP_KCLASS_LOSSY16 = 6 /* fir_lossy_kernel: class B on 16-bit types, int32 VALU */,
       ACDSP_KCLASS_SATACC16 = 7 /* fir_satacc_kernel: saturating accumulator of <= 32 bits on 16-bit types, reference tap order */ };

typedef struct {
  int32_t kind;               /* ACDSP_FIR_*: which reference class this mirrors (informational) */
  int32_t ftype;              /* ACDSP_SHIFT_REG ... */
  int32_t n_taps;
  int32_t n_channels;
  int32_t coeffs_per_channel; /* 0: one coefficient set shared by all channels, 1: one set per channel */
  acdsp_fmt_t in, coeff, acc, out;
  int32_t device;             /* HIP device ordinal */
  int32_t flags;
} acdsp_fir_desc_t;

typedef struct {
  int32_t interp;             /* 0: ac_cic_dec_full, 1: ac_cic_intr_full */
  int32_t R, M, N;
  int32_t n_channels;
  acdsp_fmt_t in, out;
  int32_t device;
  int32_t flags;
} acdsp_cic_desc_t;

/* ac_poly_dec<IN, COEFF, STR_COEFF, ACC, OUT, NTAPS, DF> (reference include/ac_dsp/ac_poly_dec.h:82) */
typedef struct {
  int32_t n_taps;             /* NTAPS: taps per polyphase branch; the coefficient struct holds NTAPS*DF words */
  int32_t df;                 /* decimation factor */
  int32_t n_channels;
  acdsp_fmt_t in, coeff, acc, out;
  int32_t device;
  int32_t flags;
} acdsp_polydec_desc_t;

/* ac_poly_intr<IN, COEFF, ACC, OUT, STR_CTRL, STR_COEFF, NTAPS, COEFFSZ, IF, ftype> (reference include/ac_dsp/ac_poly_intr.h:275).
 * ftype is THAT header's enum (ac_poly_intr.h:71), not the FIR FTYPE. */
enum { ACDSP_POLY_FOLD_EVEN = 0, ACDSP_POLY_FOLD_ODD = 1, ACDSP_POLY_FOLD_ANTI = 2 };
typedef struct {
  int32_t n_taps;             /* NTAPS: length of the shift register */
  int32_t coeff_sz;           /* COEFFSZ: words in the coefficient struct */
  int32_t ifac;               /* IF: interpolation factor = outputs per input sample */
  int32_t ftype;              /* ACDSP_POLY_* */
  int32_t n_channels;
  acdsp_fmt_t in, coeff, acc, out;
  int32_t device;
  int32_t flags;
} acdsp_polyintr_desc_t;

/* ac_intg_dump<IN, ACC, OUT, N_TYPE, NS, CHN> (reference include/ac_dsp/ac_intg_dump.h:113) */
typedef struct {
  int32_t ns;                 /* NS: most rounds a block may accumulate */
  int32_t chn;                /* CHN: channels interleaved in one object's stream */
  int32_t n_objects;          /* independent ac_intg_dump objects (rows of the buffers) */
  acdsp_fmt_t in, acc, out;
  int32_t device;
  int32_t flags;
} acdsp_intgdump_desc_t;

/* ac_mv_avg<MAX_SAMPLE, TAPS, WIN_TYPE, IN, OUT, ACC, COEFF, S_TYPE> (reference include/ac_dsp/ac_mv_avg.h:135) */
enum { ACDSP_WIN_PLAIN = 0 /* AC_WIN */, ACDSP_WIN_MIRROR = 1 /* AC_MIRROR */, ACDSP_WIN_CLIP = 2 /* AC_CLIP */ };
typedef struct {
  int32_t max_sample;         /* MAX_SAMPLE: longest frame the object accepts */
  int32_t taps;               /* TAPS: window length, odd (the reference's MAC loop reads coeffs[TAPS] on even values) */
  int32_t win_mode;           /* ACDSP_WIN_* */
  int32_t n_objects;          /* independent ac_mv_avg objects (rows of the buffers) */
  acdsp_fmt_t in, coeff, acc, out;
  int32_t device;
  int32_t flags;
} acdsp_mvavg_desc_t;

/* Raw-integer stream file / wire format: this 64-byte little-endian header, then n_channels rows of `stride` containers
 * (two's-complement raw words, acdsp_elem_bytes(W) bytes each) -- the [channel][time] layout the kernels take. */
typedef struct {
  char magic[8];              /* "ACDSPRAW" (filled in by acdsp_stream_write) */
  uint32_t version;           /* 1 */
  uint32_t elem_bytes;        /* 2, 4 or 8 = acdsp_elem_bytes(fmt.W) */
  acdsp_fmt_t fmt;            /* the ac_fixed<W,I,S,Q,O> the words are in */
  uint32_t reserved;          /* 0 */
  uint64_t n_channels, n_samples, stride;   /* stride >= n_samples, in elements */
} acdsp_stream_hdr_t;

typedef struct acdsp_fir *acdsp_fir_t;
typedef struct acdsp_polydec *acdsp_polydec_t;
typedef struct acdsp_cic *acdsp_cic_t;
typedef struct acdsp_ddc *acdsp_ddc_t;
typedef struct acdsp_polyintr *acdsp_polyintr_t;
typedef struct acdsp_intgdump *acdsp_intgdump_t;
typedef struct acdsp_mvavg *acdsp_mvavg_t;

/* ---- general ---- */
int32_t acdsp_abi_version(void);
const char *acdsp_last_error(void);            /* message for the last failing call on this thread */
int32_t acdsp_device_count(void);              /* number of gfx950 devices, 0 if none */
int32_t acdsp_elem_bytes(int32_t W);           /* 2, 4 or 8 */

/* device-memory helpers so that pure C/C++ callers need no HIP headers */
int32_t acdsp_dev_alloc(int32_t device, uint64_t bytes, void **d_ptr);
int32_t acdsp_dev_free(int32_t device, void *d_ptr);
/* Allocation with a placement probe: the HBM-bound operators run 3 - 8 % apart on different (input, output) allocation pairs
 * (profiles/r3_placement_modes.txt).  n_candidates blocks of `bytes` are allocated, each is timed with a bare mixed stream against the partner block
 * (partner_reads != 0: the partner is read and the new block written -- an output allocated beside its input; 0: the other way round), the
 * fastest is kept, the others freed.  probe_ms: optional [n_candidates] probe times.  Without a partner (or n_candidates <= 1): plain allocation. */
int32_t acdsp_dev_alloc_paired(int32_t device, uint64_t bytes, const void *d_partner, uint64_t partner_bytes, int32_t partner_reads,
                               int32_t n_candidates, void **d_ptr, float *probe_ms);
/* ... with the caller's own call as the probe (the bare stream ranks pairs like the operator only for thin streams: 16 : 1 and beyond):
 * trial(ctx, d_candidate) enqueues ONE call of the operator on the NULL stream with the candidate as its output (or input) and returns 0.  Every
 * candidate is timed over `reps` calls behind one untimed call, twice round; the fastest is kept, the others freed.  The trial runs the
 * operator for real: point it at a clone of the handle (acdsp_*_clone) when the stream's state must not advance.  trial_ms: optional [n_candidates]. */
int32_t acdsp_dev_alloc_shop(int32_t device, uint64_t bytes, int32_t n_candidates, int32_t (*trial)(void *ctx, void *d_candidate), void *ctx, int32_t reps,
                             void **d_ptr, float *trial_ms);
/* the probe itself: average ms of a bare stream reading [d_src, +src_bytes) and writing [d_dst, +dst_bytes) at their byte ratio */
int32_t acdsp_diag_mix_ms(int32_t device, const void *d_src, uint64_t src_bytes, void *d_dst, uint64_t dst_bytes, int32_t warmup, int32_t reps, void *stream,
                          float *ms_avg);
int32_t acdsp_copy_h2d(int32_t device, void *d_dst, const void *h_src, uint64_t bytes);
int32_t acdsp_copy_d2h(int32_t device, void *h_dst, const void *d_src, uint64_t bytes);
int32_t acdsp_sync(int32_t device, void *stream);

/* Counter-hash stimulus written straight into HBM (bench/test inputs too big to
 * ship over PCIe).  Element (c,t) = top `bits` bits of splitmix64(seed, ((ch0+c)<<32)|(t0+t)),
 * sign-extended, stored in an elem_bytes-wide integer. */
int32_t acdsp_fill_stimulus(int32_t device, void *d_ptr, int32_t elem_bytes, int64_t n_ch, int64_t n, int64_t stride,
                            uint64_t seed, int32_t bits, uint64_t ch0, uint64_t t0, void *stream);

/* ---- diagnostics: the two reference speeds bench.py prints beside a FIR row's roofline ("also report vs a measured
 * device-copy bandwidth"; DESIGN 5.2: the envelope of the int8-split formulation).  Measurement kernels only -- nothing a filter
 * caller needs; they run on the caller's buffers and stream, `warmup` untimed launches then `reps` launches between two events.
 *   copy:      dst[i] = src[i], one 16-byte element per thread, 256-thread workgroups in memory order.  GB/s = 2 * bytes / ms.
 *   envelope:  streams `bytes` from d_x to d_y (d_y receives meaningless words) in 32 KB spans with 8-load / 8-store non-temporal
 *              bursts while issuing `mfma_per_step` v_mfma_i32_32x32x32_i8 per 1024 int16 samples, `mfma_hi_per_step` of them on
 *              high-byte-plane Toeplitz fragments of `coeffs` (n_taps raw 16-bit words) -- no byte-plane split, no LDS, no
 *              epilogue.  Compiled (per_step, hi): (0,0) (26,8) (36,18) (76,10) (132,66); anything else is ACDSP_EUNSUPPORTED. */
int32_t acdsp_diag_copy_ms(int32_t device, const void *d_src, void *d_dst, uint64_t bytes, int32_t warmup, int32_t reps, void *stream,
                           float *ms_avg);
/* shader clock (MHz) measured by a short spin kernel on `stream`, i.e. right behind whatever the stream ran last: the clock state of a bench row */
int32_t acdsp_diag_shader_clock_mhz(int32_t device, void *stream, float *mhz);
int32_t acdsp_diag_fir_envelope_ms(int32_t device, const int64_t *coeffs, int32_t n_taps, int32_t mfma_per_step, int32_t mfma_hi_per_step,
                                   const void *d_x, void *d_y, uint64_t bytes, int32_t warmup, int32_t reps, void *stream, float *ms_avg);

/* the same MFMA counts in the COPY kernel's geometry (256-thread workgroups in memory order, every wave lives for its loads, their MFMAs and
 * stores).  coeffs == NULL: one 16-byte element per thread, A operands are stand-ins made of the loaded bytes (such a wave cannot keep
 * fragments resident); coeffs != NULL: four elements per thread, the real Toeplitz fragments of the set loaded once per wave */
int32_t acdsp_diag_fir_envelope_copygeom_ms(int32_t device, const int64_t *coeffs, int32_t n_taps, int32_t mfma_per_step, int32_t mfma_hi_per_step,
                                            const void *d_x, void *d_y, uint64_t bytes, int32_t warmup, int32_t reps, void *stream, float *ms_avg);

/* ---- FIR ---- */
int32_t acdsp_fir_create(const acdsp_fir_desc_t *desc, acdsp_fir_t *out);
int32_t acdsp_fir_destroy(acdsp_fir_t h);
/* Deep copy, state included (the reference objects are plain aggregates: copying one copies its
 * shift register / reg_trans / coefficients). */
int32_t acdsp_fir_clone(acdsp_fir_t h, acdsp_fir_t *out);
/* Raw coefficient words from host memory: [n_taps] or [n_channels][n_taps].
 * const_coeffs: call once (the reference borrows the pointer for the object's
 * life); load_coeffs: the coefficient-load phase of run(); prog_coeffs: before
 * every run().  Takes effect for samples processed by later run() calls. */
int32_t acdsp_fir_set_coeffs(acdsp_fir_t h, const int64_t *coeffs);
/* Filter n_samples per channel.  d_in / d_out are device pointers to IN / OUT
 * containers, strides in elements.  Asynchronous on `stream` (a hipStream_t, or
 * NULL for the default stream). */
int32_t acdsp_fir_run(acdsp_fir_t h, const void *d_in, int64_t in_stride, int64_t n_samples, void *d_out,
                      int64_t out_stride, void *stream);
/* Same, host buffers, dense [n_channels][n_samples]; synchronous. */
int32_t acdsp_fir_run_host(acdsp_fir_t h, const void *h_in, int64_t n_samples, void *h_out);
int32_t acdsp_fir_reset(acdsp_fir_t h);        /* back to the freshly constructed state (coefficients kept) */
int32_t acdsp_fir_path(acdsp_fir_t h);         /* ACDSP_PATH_* chosen for the current coefficients */
/* int8 matrix-core path only (else -1): epilogue class (0 generic, 1 / 2 the 32-bit classes, 3 wide, 4 64-bit branch-free) | coefficient pre-shift << 8 |
 * sign-flipped unsigned samples << 16 -- diagnostic, recorded by tests/test_pathmap_gpu.py */
int32_t acdsp_fir_mfma_epilogue(acdsp_fir_t h);
int32_t acdsp_fir_kernel_class(acdsp_fir_t h); /* the same, with ACDSP_KCLASS_* inside ACDSP_PATH_GENERIC; -1 before set_coeffs */
/* Duration of the main kernel of the most recent TIMED run(), from HIP events
 * recorded on the launch stream (blocks until that kernel has finished).  Small host-side
 * calls (run_host of at most 64 KB: the drop-in classes' one-channel path) are launch-bound
 * and record no events: after such a call these two report the last timed run (ACDSP_ESTATE
 * when there has been none). */
int32_t acdsp_fir_last_kernel_ms(acdsp_fir_t h, float *ms);
/* Average / minimum main-kernel duration over the last `last_k` (<= 64) run() calls. */
int32_t acdsp_fir_kernel_stats(acdsp_fir_t h, int32_t last_k, float *avg_ms, float *min_ms);
/* 32x32x32 int8 MFMA instructions the selected kernel issues per 1024 samples of one channel (0 when the current
 * coefficients do not run on the int8 matrix-core path): the numerator of an MFMA-utilisation figure.  All-zero
 * high-byte blocks of the coefficient set are not issued, so this is <= 4 * (ceil((n_taps - 1) / 32) + 1). */
int32_t acdsp_fir_mfma_issued(acdsp_fir_t h, int32_t *per_1024_samples);

/* ---- CIC ---- */
int32_t acdsp_cic_create(const acdsp_cic_desc_t *desc, acdsp_cic_t *out);
int32_t acdsp_cic_destroy(acdsp_cic_t h);
int32_t acdsp_cic_clone(acdsp_cic_t h, acdsp_cic_t *out);
int32_t acdsp_cic_int_type(const acdsp_cic_desc_t *desc, acdsp_fmt_t *it); /* the reference's lossless INT_TYPE */
int64_t acdsp_cic_out_count(acdsp_cic_t h, int64_t n_in); /* outputs per channel the next run(n_in) produces */
int32_t acdsp_cic_run(acdsp_cic_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                      int64_t *n_out, void *stream);
int32_t acdsp_cic_run_host(acdsp_cic_t h, const void *h_in, int64_t n_in, void *h_out, int64_t out_cap, int64_t *n_out);
int32_t acdsp_cic_reset(acdsp_cic_t h);
int32_t acdsp_cic_path(acdsp_cic_t h);         /* last run(): 0 recurrence kernel, 3 FIR-identity MFMA kernel (ACDSP_PATH_MFMA_GEN), 6 two-stage kernel (ACDSP_PATH_CIC_2STAGE; decimation
                                                  factors from 32 on that a compiled stage-1 rate divides -- others keep the recurrence kernel), 4 wide */
int32_t acdsp_cic_last_kernel_ms(acdsp_cic_t h, float *ms);
int32_t acdsp_cic_kernel_stats(acdsp_cic_t h, int32_t last_k, float *avg_ms, float *min_ms);

/* ---- polyphase decimator (replaces the loop nest of ac_poly_dec::run, ac_poly_dec.h:109-128) ---- */
int32_t acdsp_polydec_create(const acdsp_polydec_desc_t *desc, acdsp_polydec_t *out);
int32_t acdsp_polydec_destroy(acdsp_polydec_t h);
int32_t acdsp_polydec_set_coeffs(acdsp_polydec_t h, const int64_t *coeffs); /* [NTAPS*DF], the STR_COEFF_TYPE array */
/* n_in must be a multiple of DF (run() consumes whole groups: `while (data_in.available(DF))`); n_in/DF outputs */
int32_t acdsp_polydec_run(acdsp_polydec_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out,
                          int64_t out_stride, void *stream);
int32_t acdsp_polydec_run_host(acdsp_polydec_t h, const void *h_in, int64_t n_in, void *h_out);
int32_t acdsp_polydec_reset(acdsp_polydec_t h);
int32_t acdsp_polydec_path(acdsp_polydec_t h);  /* ACDSP_PATH_GENERIC or ACDSP_PATH_MFMA_GEN */

/* ---- DDC cascade: ac_cic_dec_full -> FIR on the decimator's lossless INT_TYPE words ----
 * Replaces the pair of run() calls `cic.run(in, mid); fir.run(mid, out);` on many channels (the reference couples the
 * two blocks through an ac_channel of INT_TYPE words, cf. ac_cic_dec_full.h:164-177).  cic->out and fir->in must both be
 * the decimator's INT_TYPE (acdsp_cic_int_type).  For the BASELINE config-5 shape class (int16 input, 36-bit words,
 * <= 127 taps, int32 output containers) both stages run in ONE kernel and the INT_TYPE stream never reaches HBM
 * (acdsp_ddc_path = 1); otherwise the two stage kernels run back to back through an internal buffer (path 0).
 * Results are identical to running the two handles separately; state carries across calls like theirs. */
int32_t acdsp_ddc_create(const acdsp_cic_desc_t *cic, const acdsp_fir_desc_t *fir, acdsp_ddc_t *out);
int32_t acdsp_ddc_destroy(acdsp_ddc_t h);
int32_t acdsp_ddc_set_coeffs(acdsp_ddc_t h, const int64_t *coeffs);   /* FIR coefficients, raw COEFF_TYPE words [n_taps] */
int64_t acdsp_ddc_out_count(acdsp_ddc_t h, int64_t n_in);
int32_t acdsp_ddc_run(acdsp_ddc_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                      int64_t *n_out, void *stream);
int32_t acdsp_ddc_reset(acdsp_ddc_t h);
int32_t acdsp_ddc_path(acdsp_ddc_t h);                                  /* 1: fused kernel, 0: two kernels */
int32_t acdsp_ddc_kernel_stats(acdsp_ddc_t h, int32_t last_k, float *avg_ms, float *min_ms);

/* ---- polyphase interpolator (reference ac_poly_intr.h:104-320) ----
 * One input sample -> IF outputs.  The folded cores emit the sums of a sample one sample later (accumulator banks
 * acc_a / acc_b, :153-175), so the stream's very first sample produces nothing.  set_ctrl = the `read_ctrl` call of run()
 * (:291-294): coeffs[COEFFSZ] from the coefficient struct, sign[IF] / corr[IF] from the control struct. */
int32_t acdsp_polyintr_create(const acdsp_polyintr_desc_t *desc, acdsp_polyintr_t *out);
int32_t acdsp_polyintr_destroy(acdsp_polyintr_t h);
int32_t acdsp_polyintr_set_ctrl(acdsp_polyintr_t h, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr);
int64_t acdsp_polyintr_out_count(acdsp_polyintr_t h, int64_t n_in);
int32_t acdsp_polyintr_run(acdsp_polyintr_t h, const void *d_in, int64_t in_stride, int64_t n_in, void *d_out, int64_t out_stride,
                           int64_t *n_out, void *stream);
int32_t acdsp_polyintr_run_host(acdsp_polyintr_t h, const void *h_in, int64_t n_in, void *h_out, int64_t out_cap, int64_t *n_out);
int32_t acdsp_polyintr_reset(acdsp_polyintr_t h);
int32_t acdsp_polyintr_path(acdsp_polyintr_t h);   /* kernel family of the last run(): ACDSP_PATH_GENERIC, _LOSSLESS64 or _MFMA_GEN */

/* ---- integrate and dump (reference ac_intg_dump.h:93-147) ----
 * Rows = objects; a row is the interleaved stream the reference reads from data_in (round-major, channel-minor).
 * n_sample[n_blocks] (host array, shared by the objects) are the N_TYPE words read at the start of each block: a block
 * with 1 <= n_sample <= NS takes n_sample rounds and yields CHN outputs, any other value takes NS rounds and yields none
 * (its sums carry on, :138-146).  The call must hold whole blocks (the reference would read an empty channel otherwise). */
int32_t acdsp_intgdump_create(const acdsp_intgdump_desc_t *desc, acdsp_intgdump_t *out);
int32_t acdsp_intgdump_destroy(acdsp_intgdump_t h);
int32_t acdsp_intgdump_counts(acdsp_intgdump_t h, const int64_t *n_sample, int64_t n_blocks, int64_t *n_in, int64_t *n_out);
int32_t acdsp_intgdump_run(acdsp_intgdump_t h, const void *d_in, int64_t in_stride, const int64_t *n_sample, int64_t n_blocks,
                           void *d_out, int64_t out_stride, int64_t *n_out, void *stream);
int32_t acdsp_intgdump_run_host(acdsp_intgdump_t h, const void *h_in, const int64_t *n_sample, int64_t n_blocks, void *h_out,
                                int64_t out_cap, int64_t *n_out);
int32_t acdsp_intgdump_reset(acdsp_intgdump_t h);
int32_t acdsp_intgdump_path(acdsp_intgdump_t h);   /* kernel family of the last run(): 0 exact order (ragged blocks / carried sums / saturating ACC), 1 LDS-tiled, 2 streaming, 3 matrix cores (selection-matrix product: CHN that does not divide a 16-byte load) */

/* ---- moving average (reference ac_mv_avg.h:93-196) ----
 * One run() = one run() call of every object: n_frames frames of n_sample input samples each, back to back in the row (the
 * reference reads n_sample once per call, :155, and loops `while (data_in.available(1))` over frames).  A frame yields
 * n_sample outputs in the boundary modes and n_sample - TAPS + 1 (or none) under AC_WIN.  The window class behind it,
 * ac_window_1d_flag, is not part of ac_dsp: see include/ac_types/ac_window.h for the semantics assumed.  No state crosses
 * calls (the reference constructs its core object inside run(), :154); coefficients are bound like ac_fir_const_coeffs'. */
int32_t acdsp_mvavg_create(const acdsp_mvavg_desc_t *desc, acdsp_mvavg_t *out);
int32_t acdsp_mvavg_destroy(acdsp_mvavg_t h);
int32_t acdsp_mvavg_set_coeffs(acdsp_mvavg_t h, const int64_t *coeffs);   /* raw COEFF_TYPE words [TAPS] */
int64_t acdsp_mvavg_out_per_frame(acdsp_mvavg_t h, int64_t n_sample);     /* -1: n_sample outside 1..MAX_SAMPLE */
int32_t acdsp_mvavg_path(acdsp_mvavg_t h);   /* kernel family of the last run(): 0 exact per-tap order, 1 order-free int64 sums, 2 streaming (16-bit samples, int32 sums), 3 sliding window on 32-bit samples (int64 sums), 4 streaming with the window sums on the matrix cores (17 taps and more) */
int32_t acdsp_mvavg_run(acdsp_mvavg_t h, const void *d_in, int64_t in_stride, int64_t n_sample, int64_t n_frames, void *d_out,
                        int64_t out_stride, int64_t *n_out, void *stream);
int32_t acdsp_mvavg_run_host(acdsp_mvavg_t h, const void *h_in, int64_t n_sample, int64_t n_frames, void *h_out, int64_t out_cap,
                             int64_t *n_out);

/* ---- state save / restore (checkpoint / resume, moving a stream to another GPU) ----
 * The reference's filter state is plain object members (shift register / reg_trans: ac_fir_const_coeffs.h:124-127;
 * integrator / comb registers and rate counters: ac_cic_full_core.h:71-74,219) that a caller can copy with the object.
 * Here it is a versioned little-endian blob: a 64-byte header (magic "ACDSPST1", kind, channel count, word size, words per
 * channel, the CIC input count, the filter parameters) followed by the raw state words [channel][word]:
 *   FIR   -- the last `per_channel` input samples of every channel, oldest first (word per_channel-1-k is reg[k] of the
 *            reference's shift register); TRANSPOSED load/prog filters: reg_trans[0..N_TAPS-1] as int64 ACC raw words;
 *   CIC   -- the last `per_channel` input samples (both directions are FIR systems of the input)
 *            plus the number of inputs consumed so far (= rate_cnt phase).
 * state_get waits for the handle's pending run() calls.  state_set accepts only a blob from a handle of the same class,
 * parameters and channel count (ACDSP_EINVAL otherwise); coefficients are not part of the state. */
int64_t acdsp_fir_state_size(acdsp_fir_t h);                                   /* bytes state_get writes; -1 on a null handle */
int32_t acdsp_fir_state_get(acdsp_fir_t h, void *h_buf, uint64_t cap_bytes);
int32_t acdsp_fir_state_set(acdsp_fir_t h, const void *h_buf, uint64_t bytes);
int64_t acdsp_cic_state_size(acdsp_cic_t h);
int32_t acdsp_cic_state_get(acdsp_cic_t h, void *h_buf, uint64_t cap_bytes);
int32_t acdsp_cic_state_set(acdsp_cic_t h, const void *h_buf, uint64_t bytes);
/* cascade (acdsp_ddc_*): the input history and input count of the fused kernel (its FIR window is recomputed from them), or
 * the two stage blobs behind one header when the handle runs the stages as two kernels */
int64_t acdsp_ddc_state_size(acdsp_ddc_t h);
int32_t acdsp_ddc_state_get(acdsp_ddc_t h, void *h_buf, uint64_t cap_bytes);
int32_t acdsp_ddc_state_set(acdsp_ddc_t h, const void *h_buf, uint64_t bytes);

/* ---- node level: one filter bank sharded over the GPUs of a node ----
 * Channels are independent filter objects (reference ac_fir_const_coeffs.h:124-127, ac_cic_full_core.h:71-74,219), so a bank of
 * desc->n_channels filters is cut into n_devices CONTIGUOUS channel slices (acdsp_node_shard: the first n_channels % n_devices
 * slices hold one channel more), one per entry of `devices` (NULL: devices 0 .. n_devices-1; a device may be listed more than once:
 * its shards are then concurrent streams of that GPU).  Per shard the node handle owns an ordinary engine handle (`desc` with that
 * slice's channel count and device), a non-blocking stream and a host thread bound to the device.  Coefficients are replicated
 * (or, with coeffs_per_channel, sliced); there is NO collective and no cross-device traffic.
 * run():      d_in[s] / d_out[s] = device pointers ON shard s's device to its [ch_hi - ch_lo][stride] block; every shard's thread
 *             launches its slice on its stream and waits for it; the call returns when all have.  Aggregate rate of a call =
 *             n_channels * n / acdsp_node_last_ms's maximum (per-shard times: the engine handles' own HIP events).
 *             PRECONDITION: the shard streams are non-blocking and order with no other stream -- every d_in block must be complete
 *             and every d_out block unused by pending work of other streams when run() is called (synchronise the producer first).
 * run_host(): dense host block [n_channels][n]: every thread moves and filters its rows (acdsp_fir_run_host of the slice).
 * The shard handles (acdsp_node_shard_info) take every per-handle call of this header: state get / set, reset, path, kernel_stats. */
typedef struct acdsp_node *acdsp_node_t;
int32_t acdsp_node_shard(int64_t n_total, int32_t n_shards, int32_t shard, int64_t *lo, int64_t *hi);
int32_t acdsp_node_n_shards(acdsp_node_t h);   /* -1 on a null handle */
int32_t acdsp_node_shard_info(acdsp_node_t h, int32_t shard, int32_t *device, int64_t *ch_lo, int64_t *ch_hi, void **handle, void **stream);
int32_t acdsp_node_last_ms(acdsp_node_t h, float *per_shard_ms /* [n_shards] or NULL */, float *max_ms);
int32_t acdsp_node_destroy(acdsp_node_t h);
int32_t acdsp_node_fir_create(const acdsp_fir_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int32_t acdsp_node_fir_set_coeffs(acdsp_node_t h, const int64_t *coeffs);   /* [n_taps], or [n_channels][n_taps] with coeffs_per_channel */
int32_t acdsp_node_fir_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_samples, void *const *d_out, int64_t out_stride);
int32_t acdsp_node_fir_run_host(acdsp_node_t h, const void *h_in, int64_t n_samples, void *h_out);
int32_t acdsp_node_cic_create(const acdsp_cic_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int64_t acdsp_node_cic_out_count(acdsp_node_t h, int64_t n_in);
int32_t acdsp_node_cic_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride,
                           int64_t *n_out);
int32_t acdsp_node_ddc_create(const acdsp_cic_desc_t *cic, const acdsp_fir_desc_t *fir, int32_t n_devices, const int32_t *devices,
                              acdsp_node_t *out);
int32_t acdsp_node_ddc_set_coeffs(acdsp_node_t h, const int64_t *coeffs);
int64_t acdsp_node_ddc_out_count(acdsp_node_t h, int64_t n_in);
int32_t acdsp_node_ddc_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride,
                           int64_t *n_out);
/* the f-row classes: rows = channels (poly_dec, poly_intr) or objects (intg_dump, mv_avg), sliced the same way; coefficients /
 * control words / block counts are replicated.  acdsp_node_last_ms: the shard's whole call between two events of its stream. */
int32_t acdsp_node_polydec_create(const acdsp_polydec_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int32_t acdsp_node_polydec_set_coeffs(acdsp_node_t h, const int64_t *coeffs);
int32_t acdsp_node_polydec_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride);
int32_t acdsp_node_polyintr_create(const acdsp_polyintr_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int32_t acdsp_node_polyintr_set_ctrl(acdsp_node_t h, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr);
int64_t acdsp_node_polyintr_out_count(acdsp_node_t h, int64_t n_in);
int32_t acdsp_node_polyintr_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_in, void *const *d_out, int64_t out_stride,
                                int64_t *n_out);
int32_t acdsp_node_intgdump_create(const acdsp_intgdump_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int32_t acdsp_node_intgdump_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, const int64_t *n_sample, int64_t n_blocks,
                                void *const *d_out, int64_t out_stride, int64_t *n_out);
int32_t acdsp_node_mvavg_create(const acdsp_mvavg_desc_t *desc, int32_t n_devices, const int32_t *devices, acdsp_node_t *out);
int32_t acdsp_node_mvavg_set_coeffs(acdsp_node_t h, const int64_t *coeffs);
int32_t acdsp_node_mvavg_run(acdsp_node_t h, const void *const *d_in, int64_t in_stride, int64_t n_sample, int64_t n_frames, void *const *d_out,
                             int64_t out_stride, int64_t *n_out);

/* ---- raw-integer stream files (host side; no device needed) ---- */
int32_t acdsp_stream_write(const char *path, const acdsp_stream_hdr_t *hdr, const void *data);
int32_t acdsp_stream_read_header(const char *path, acdsp_stream_hdr_t *hdr);
int32_t acdsp_stream_read(const char *path, void *data, uint64_t cap_bytes);

#ifdef __cplusplus
}
#endif
#endif
