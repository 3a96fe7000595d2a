/* acdsp_oracle.c -- CPU restatement of the hlslibs/ac_dsp FIR/CIC hot path.
 * TEST INFRASTRUCTURE ONLY -- see acdsp_oracle.h for the rules and the
 * pinning status.  Plain C, gcc, __int128 for exact intermediates.
 *
 * Reference files restated here (all under /root/reference/include/ac_dsp/):
 *   ac_fir_const_coeffs.h:153-296   (identical cores: ac_fir_load_coeffs.h:145-278,
 *                                    ac_fir_prog_coeffs.h:110-247)
 *   ac_cic_full_core.h:80-160,198-255
 *   ac_cic_dec_full.h:116-137,187-222   ac_cic_intr_full.h:107-127,173-215
 */
#include "acdsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* ac_fixed conversion rules (AC Datatypes semantics, restated)        */
/* ------------------------------------------------------------------ */

static i128 fmt_min(const orc_fmt_t *f) { return f->S ? -(((i128)1) << (f->W - 1)) : (i128)0; }
static i128 fmt_max(const orc_fmt_t *f) {
  return f->S ? (i128)((((u128)1) << (f->W - 1)) - 1) : (i128)((((u128)1) << f->W) - 1);
}

static i128 wrap_w(i128 x, int W, int S) {
  u128 m = (((u128)1) << W) - 1; /* W <= 64 here */
  u128 u = ((u128)x) & m;
  if (S && ((u >> (W - 1)) & 1)) { u |= ~m; }
  return (i128)u;
}

/* Quantise x*2^-f_src to f_dst fractional bits.  The increment rules are the
 * eight ac_q_mode behaviours: qb = most significant dropped bit, r = OR of the
 * remaining dropped bits, neg = sign of the source, lsb = LSB of the kept part. */
static i128 quantize(i128 x, int f_src, int f_dst, int Q) {
  int sh = f_src - f_dst;
  if (sh <= 0) { return (-sh >= 128) ? (i128)0 : (i128)((u128)x << (-sh)); }
  int neg = x < 0, sticky = 0;
  if (sh > 126) {
    int k = sh - 126;
    i128 xs = (k >= 128) ? (neg ? (i128)-1 : (i128)0) : (x >> k);
    sticky = (k >= 128) ? (x != 0) : ((x - (i128)((u128)xs << k)) != 0);
    x = xs;
    sh = 126;
  }
  i128 q = x >> sh;
  i128 rem = x - (i128)((u128)q << sh);
  i128 half = ((i128)1) << (sh - 1);
  int qb = rem >= half;
  int r = ((rem & (half - 1)) != 0) || sticky;
  int lsb = (int)(q & 1);
  int inc = 0;
  switch (Q) {
    case ORC_TRN: inc = 0; break;                       /* toward -inf */
    case ORC_RND: inc = qb; break;                      /* nearest, ties toward +inf */
    case ORC_TRN_ZERO: inc = neg && (qb || r); break;   /* toward zero */
    case ORC_RND_ZERO: inc = qb && (r || neg); break;   /* nearest, ties toward zero */
    case ORC_RND_INF: inc = qb && (r || !neg); break;   /* nearest, ties away from zero */
    case ORC_RND_MIN_INF: inc = qb && r; break;         /* nearest, ties toward -inf */
    case ORC_RND_CONV: inc = qb && (r || lsb); break;   /* nearest, ties to even */
    case ORC_RND_CONV_ODD: inc = qb && (r || !lsb); break; /* nearest, ties to odd */
    default: break;
  }
  return q + inc;
}

static i128 overflow(i128 q, const orc_fmt_t *f) {
  i128 lo = fmt_min(f), hi = fmt_max(f);
  int ovf = (q < lo) || (q > hi);
  switch (f->O) {
    case ORC_WRAP: return wrap_w(q, f->W, f->S);
    case ORC_SAT: return ovf ? ((q < lo) ? lo : hi) : q;
    case ORC_SAT_ZERO: return ovf ? (i128)0 : q;
    case ORC_SAT_SYM:
      if (f->S) {
        if (ovf) { return (q < 0) ? lo + 1 : hi; }
        return (q == lo && f->W > 1) ? lo + 1 : q;
      }
      return ovf ? ((q < lo) ? lo : hi) : q;
    default: return q;
  }
}

static int64_t requant(i128 x, int f_src, const orc_fmt_t *dst) {
  return (int64_t)overflow(quantize(x, f_src, dst->W - dst->I, dst->Q), dst);
}

int64_t orc_requant128(int64_t x_hi, uint64_t x_lo, int32_t f_src, const orc_fmt_t *dst) {
  i128 x = (i128)(((u128)(uint64_t)x_hi << 64) | (u128)x_lo);
  return requant(x, f_src, dst);
}
int64_t orc_requant(int64_t x, int32_t f_src, const orc_fmt_t *dst) { return requant((i128)x, f_src, dst); }

int64_t orc_from_double(double d, const orc_fmt_t *dst) {
  if (d == 0.0 || d != d) { return 0; }
  int ex;
  double fr = frexp(d, &ex);
  i128 m = (i128)(long long)ldexp(fr, 53); /* d == m * 2^(ex-53), exact */
  int e = ex - 53;
  if (e > 70) { /* astronomically out of range */
    orc_fmt_t t = *dst;
    i128 big = (d < 0) ? -(((i128)1) << 100) : (((i128)1) << 100);
    return (int64_t)overflow(t.O == ORC_WRAP ? (i128)0 : big, &t);
  }
  return requant(m, -e, dst);
}

/* exact a*2^-fa + b*2^-fb at max(fa,fb) fractional bits */
static i128 add_aligned(i128 a, int fa, i128 b, int fb, int *f_out) {
  int f = fa > fb ? fa : fb;
  *f_out = f;
  return (i128)((u128)a << (f - fa)) + (i128)((u128)b << (f - fb));
}

/* ------------------------------------------------------------------ */
/* FIR cores                                                           */
/* ------------------------------------------------------------------ */

struct orc_fir {
  int32_t n, ftype;
  orc_fmt_t in, cf, acc, out;
  int fi, fc, fa;      /* fractional bits of IN, COEFF, ACC */
  int64_t *reg;        /* IN_TYPE reg[N_TAPS]       (ac_fir_const_coeffs.h:124) */
  int64_t *reg_trans;  /* ACC_TYPE reg_trans[N_TAPS] (ac_fir_const_coeffs.h:125) */
  int32_t wptr;        /* circular-buffer write pointer (ac_fir_const_coeffs.h:126) */
};

orc_fir_t *orc_fir_new(int32_t n_taps, int32_t ftype, const orc_fmt_t *in, const orc_fmt_t *coeff,
                       const orc_fmt_t *acc, const orc_fmt_t *out) {
  if (n_taps < 1) { return NULL; }
  const orc_fmt_t *fs[4] = {in, coeff, acc, out};
  for (int i = 0; i < 4; i++) {
    if (fs[i]->W < 1 || fs[i]->W > 64) { return NULL; }
  }
  orc_fir_t *f = (orc_fir_t *)calloc(1, sizeof *f);
  f->n = n_taps; f->ftype = ftype;
  f->in = *in; f->cf = *coeff; f->acc = *acc; f->out = *out;
  f->fi = in->W - in->I; f->fc = coeff->W - coeff->I; f->fa = acc->W - acc->I;
  f->reg = (int64_t *)calloc((size_t)n_taps, sizeof(int64_t));       /* init_array<AC_VAL_0>, :145 */
  f->reg_trans = (int64_t *)calloc((size_t)n_taps, sizeof(int64_t)); /* :146 */
  f->wptr = 0;
  return f;
}
void orc_fir_free(orc_fir_t *f) {
  if (!f) { return; }
  free(f->reg); free(f->reg_trans); free(f);
}
void orc_fir_reset(orc_fir_t *f) {
  memset(f->reg, 0, sizeof(int64_t) * (size_t)f->n);
  memset(f->reg_trans, 0, sizeof(int64_t) * (size_t)f->n);
  f->wptr = 0;
}

/* firShiftReg -- ac_fir_const_coeffs.h:153-159 */
static void fir_shift_reg(orc_fir_t *f, int64_t din) {
  for (int i = f->n - 1; i >= 0; i--) { f->reg[i] = (i == 0) ? din : f->reg[i - 1]; }
}

/* `acc += a * b` : exact product, exact sum with acc, then ACC_TYPE quantise/overflow */
static int64_t mac(const orc_fir_t *f, int64_t acc, i128 prod, int f_prod) {
  int fs;
  i128 s = add_aligned((i128)acc, f->fa, prod, f_prod, &fs);
  return requant(s, fs, &f->acc);
}

int32_t orc_fir_step(orc_fir_t *f, const int64_t *c, int64_t x, int64_t *y) {
  const int N = f->n;
  int64_t acc = 0; /* ACC_TYPE acc = 0.0 */
  switch (f->ftype) {
    case ORC_SHIFT_REG: { /* :190-199 */
      fir_shift_reg(f, x);
      for (int i = N - 1; i >= 0; i--) { acc = mac(f, acc, (i128)f->reg[i] * c[i], f->fi + f->fc); }
      break;
    }
    case ORC_ROTATE_SHIFT: { /* :205-220 */
      int64_t temp_rotate;
      for (int i = N; i >= 0; i--) {
        if (i == N) {
          temp_rotate = x;
        } else {
          temp_rotate = f->reg[N - 1];
          acc = mac(f, acc, (i128)f->reg[N - 1] * c[i], f->fi + f->fc);
        }
        fir_shift_reg(f, temp_rotate);
      }
      break;
    }
    case ORC_C_BUFF: { /* :226-237 with firCircularBuffWrite/Read :165-184 */
      for (int i = 0; i <= N - 1; i++) {
        if (i == 0) {
          f->reg[f->wptr] = x;
          if (f->wptr == N - 1) { f->wptr = 0; } else { f->wptr++; }
        }
        int rptr = f->wptr - 1 - i;
        if (rptr < 0) { rptr += N; }
        acc = mac(f, acc, (i128)f->reg[rptr] * c[i], f->fi + f->fc);
      }
      break;
    }
    case ORC_FOLD_EVEN: { /* :244-253 -- pre-add is exact (full-precision sum type) */
      fir_shift_reg(f, x);
      for (int i = (N / 2) - 1; i >= 0; i--) {
        i128 pre = (i128)f->reg[i] + (i128)f->reg[N - 1 - i];
        acc = mac(f, acc, (i128)c[i] * pre, f->fc + f->fi);
      }
      break;
    }
    case ORC_FOLD_ODD: { /* :260-275 -- `fold` is an ACC_TYPE variable: the pre-add is quantised */
      fir_shift_reg(f, x);
      for (int i = 0; i < ((N - 1) / 2) + 1; i++) {
        int64_t fold;
        if (i == (N - 1) / 2) {
          fold = requant((i128)f->reg[i], f->fi, &f->acc);
        } else {
          fold = requant((i128)f->reg[i] + (i128)f->reg[(N - 1) - i], f->fi, &f->acc);
        }
        acc = mac(f, acc, (i128)c[i] * (i128)fold, f->fc + f->fa);
      }
      break;
    }
    case ORC_TRANSPOSED: { /* :281-296 */
      for (int i = N - 1; i >= 0; i--) {
        int64_t temp = (i == 0) ? 0 : f->reg_trans[i - 1];
        int fs;
        i128 s = add_aligned((i128)x * c[(N - 1) - i], f->fi + f->fc, (i128)temp, f->fa, &fs);
        f->reg_trans[i] = requant(s, fs, &f->acc);
      }
      acc = f->reg_trans[N - 1];
      break;
    }
    default: /* FOLD_EVEN_ANTI / FOLD_ODD_ANTI: run() has no branch for them (:330-352) */
      return -1;
  }
  *y = requant((i128)acc, f->fa, &f->out); /* data_out = acc */
  return 0;
}

/* ---- ac_fir_reg_share (row f1 of SURVEY 8): reference include/ac_dsp/ac_fir_reg_share.h ----
 * One run() call: firShiftReg(data_in) on the (externally owned) register array (:120-126,:286), then the
 * MAC core selected by ftype, which walks the taps in ASCENDING order in blocks of blk_sz and reads the
 * coefficient of tap i + index from coeffs[ram_addr + block_count], ram_addr advancing by mem_word_width
 * per block and block_count starting at blk_offset (:136-260).  `c` is the caller's coefficient array
 * [n_taps] exactly as passed to run().  Returns -1 where the reference indexes outside its arrays (tap
 * count of the loop not a multiple of blk_sz, or a coefficient index >= n_taps) and for ftypes run() has
 * no branch for (ROTATE_SHIFT, C_BUFF, TRANSPOSED: core_out stays unassigned, :288-306).              */
int32_t orc_fir_reg_share_step(orc_fir_t *f, const int64_t *c, int32_t mem_word_width, int32_t blk_sz, int32_t blk_offset,
                               int64_t x, int64_t *y) {
  const int N = f->n;
  int count; /* taps visited by the BLK loop */
  switch (f->ftype) {
    case ORC_SHIFT_REG: count = N; break;
    case ORC_FOLD_EVEN: case ORC_FOLD_EVEN_ANTI: count = N / 2; break;
    case ORC_FOLD_ODD: case ORC_FOLD_ODD_ANTI: count = ((N - 1) / 2) + 1; break;
    default: return -1;
  }
  if (blk_sz < 1 || mem_word_width < 0 || blk_offset < 0 || count % blk_sz != 0) { return -1; }
  if (count > 0 && ((count / blk_sz) - 1) * mem_word_width + blk_offset + blk_sz > N) { return -1; }
  fir_shift_reg(f, x); /* filter.firShiftReg(data_in) */
  int64_t acc = 0;     /* ACC_TYPE acc = 0.0 */
  int ram_addr = 0;
  for (int i = 0; i < count; i += blk_sz, ram_addr += mem_word_width) { /* BLK */
    int index = 0;
    for (int block_count = blk_offset; block_count < blk_offset + blk_sz; block_count++) { /* MAC */
      const int64_t cf = c[ram_addr + block_count];
      const int t = i + index;
      switch (f->ftype) {
        case ORC_SHIFT_REG: /* :136-150  acc += reg[i+index] * coeffs[...] */
          acc = mac(f, acc, (i128)f->reg[t] * cf, f->fi + f->fc);
          break;
        case ORC_FOLD_EVEN: /* :157-171  (reg + reg) is the exact sum type */
          acc = mac(f, acc, ((i128)f->reg[t] + (i128)f->reg[N - 1 - t]) * cf, f->fi + f->fc);
          break;
        case ORC_FOLD_EVEN_ANTI: /* :178-192 */
          acc = mac(f, acc, ((i128)f->reg[t] - (i128)f->reg[N - 1 - t]) * cf, f->fi + f->fc);
          break;
        default: { /* FOLD_ODD :199-219, FOLD_ODD_ANTI :226-246: `fold` is an ACC_TYPE variable */
          int64_t fold;
          if (t == (N - 1) / 2) {
            fold = requant((i128)f->reg[t], f->fi, &f->acc);
          } else if (f->ftype == ORC_FOLD_ODD) {
            fold = requant((i128)f->reg[t] + (i128)f->reg[(N - 1) - t], f->fi, &f->acc);
          } else {
            fold = requant((i128)f->reg[t] - (i128)f->reg[(N - 1) - t], f->fi, &f->acc);
          }
          acc = mac(f, acc, (i128)cf * (i128)fold, f->fc + f->fa);
          break;
        }
      }
      index++;
    }
  }
  *y = requant((i128)acc, f->fa, &f->out); /* data_out = acc; data_out = core_out */
  return 0;
}

int32_t orc_fir_reg_share_run(orc_fir_t *f, const int64_t *c, int32_t mem_word_width, int32_t blk_sz, int32_t blk_offset,
                              const int64_t *x, int64_t n, int64_t *y) {
  for (int64_t t = 0; t < n; t++) {
    if (orc_fir_reg_share_step(f, c, mem_word_width, blk_sz, blk_offset, x[t], &y[t])) { return -1; }
  }
  return 0;
}

/* ac_firProgCoeffs_delay_line(): data_out = reg[N_TAPS-1]  (:128-130, :310-313) -- IN_TYPE assigned to OUT_TYPE */
int64_t orc_fir_reg_share_delay_line(const orc_fir_t *f) {
  return requant((i128)f->reg[f->n - 1], f->fi, &f->out);
}

int32_t orc_fir_run(orc_fir_t *f, const int64_t *coeffs, const int64_t *x, int64_t n, int64_t *y) {
  for (int64_t t = 0; t < n; t++) { /* while (data_in.available(1)) -- :325 */
    if (orc_fir_step(f, coeffs, x[t], &y[t])) { return -1; }
  }
  return 0;
}

int32_t orc_fir_run_many(orc_fir_t **fs, int64_t n_ch, const int64_t *coeffs, int32_t coeffs_per_channel,
                         const int64_t *x, int64_t x_stride, int64_t n, int64_t *y, int64_t y_stride) {
  for (int64_t ch = 0; ch < n_ch; ch++) {
    const int64_t *c = coeffs + (coeffs_per_channel ? ch * fs[ch]->n : 0);
    if (orc_fir_run(fs[ch], c, x + ch * x_stride, n, y + ch * y_stride)) { return -1; }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* CIC                                                                 */
/* ------------------------------------------------------------------ */

struct orc_cic {
  int32_t interp, R, M, N;
  orc_fmt_t in, out, it; /* it = INT_TYPE */
  int fi;                /* fractional bits (IN == INT) */
  /* ac_cic_full_core_intg members (ac_cic_full_core.h:70-74,90-91) */
  int valid, dvalid;
  uint8_t rate_cnt1, rate_cnt, cnt; /* ac_int<8,false> */
  int64_t *intg_reg;                /* [N] */
  /* ac_cic_full_core_diff member (ac_cic_full_core.h:219) */
  int64_t *comb_dly_ln;             /* [N][M] */
  /* ac_channel<INT_TYPE> inf (ac_cic_dec_full.h:177) */
  int64_t *inf;
  int64_t inf_head, inf_size, inf_cap;
};

static int log2_ceil_u64(uint64_t x) { /* ac::log2_ceil<X>::val */
  int lf = 63;
  while (lf > 0 && !((x >> lf) & 1)) { lf--; }
  return (x == (1ull << lf)) ? lf : lf + 1;
}

/* find_inter_type_cic_dec (ac_cic_dec_full.h:116-137) / _intr (ac_cic_intr_full.h:107-127).
 * power<> is an int enum there; products >= 2^31 do not compile in the reference. */
int32_t orc_cic_int_type(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, orc_fmt_t *it) {
  if (R < 1 || M < 1 || N < 1) { return -1; }
  uint64_t pr = 1, pm = 1;
  int er = interp ? N - 1 : N;
  for (int i = 0; i < er; i++) { pr *= (uint64_t)R; if (pr >= (1ull << 31)) { return -1; } }
  for (int i = 0; i < N; i++) { pm *= (uint64_t)M; if (pm >= (1ull << 31)) { return -1; } }
  uint64_t p = pr * pm;
  if (p >= (1ull << 31)) { return -1; }
  int outF = in->W - in->I;
  int outW = log2_ceil_u64(p) + in->W + (in->S ? 0 : 1);
  it->W = outW; it->I = outW - outF; it->S = 1; it->Q = ORC_TRN; it->O = ORC_WRAP;
  return 0;
}

orc_cic_t *orc_cic_new(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, const orc_fmt_t *out) {
  orc_fmt_t it;
  if (orc_cic_int_type(interp, R, M, N, in, &it) || it.W > 64 || out->W < 1 || out->W > 64) { return NULL; }
  orc_cic_t *c = (orc_cic_t *)calloc(1, sizeof *c);
  c->interp = interp; c->R = R; c->M = M; c->N = N;
  c->in = *in; c->out = *out; c->it = it; c->fi = in->W - in->I;
  /* ac_cic_full_core_intg(bool value) with value = true (ac_cic_dec_full.h:154, ac_cic_full_core.h:94-102) */
  c->valid = 1; c->rate_cnt = 0; c->dvalid = 1; c->cnt = 0; c->rate_cnt1 = (uint8_t)(R - 1);
  c->intg_reg = (int64_t *)calloc((size_t)N, sizeof(int64_t));
  c->comb_dly_ln = (int64_t *)calloc((size_t)N * (size_t)M, sizeof(int64_t)); /* :180-190 */
  c->inf_cap = 1024; c->inf = (int64_t *)malloc(sizeof(int64_t) * (size_t)c->inf_cap);
  return c;
}
void orc_cic_free(orc_cic_t *c) {
  if (!c) { return; }
  free(c->intg_reg); free(c->comb_dly_ln); free(c->inf); free(c);
}

static void inf_write(orc_cic_t *c, int64_t v) {
  if (c->inf_head + c->inf_size == c->inf_cap) {
    if (c->inf_head > 0) {
      memmove(c->inf, c->inf + c->inf_head, sizeof(int64_t) * (size_t)c->inf_size);
      c->inf_head = 0;
    } else {
      c->inf_cap *= 2;
      c->inf = (int64_t *)realloc(c->inf, sizeof(int64_t) * (size_t)c->inf_cap);
    }
  }
  c->inf[c->inf_head + c->inf_size++] = v;
}
static int64_t inf_read(orc_cic_t *c) { c->inf_size--; return c->inf[c->inf_head++]; }

/* INT_TYPE a + b: exact sum then INT_TYPE (AC_TRN, AC_WRAP) */
static int64_t it_add(const orc_cic_t *c, int64_t a, int64_t b) { return requant((i128)a + (i128)b, c->fi, &c->it); }
static int64_t it_sub(const orc_cic_t *c, int64_t a, int64_t b) { return requant((i128)a - (i128)b, c->fi, &c->it); }

/* intStage -- ac_cic_full_core.h:80-87 (pipelined: stage i adds the OLD value of stage i-1) */
static int64_t int_stage(orc_cic_t *c, int64_t data_in) {
  for (int i = c->N - 1; i > 0; i--) { c->intg_reg[i] = it_add(c, c->intg_reg[i], c->intg_reg[i - 1]); }
  c->intg_reg[0] = it_add(c, data_in, c->intg_reg[0]);
  return c->intg_reg[c->N - 1];
}

/* diffStage -- ac_cic_full_core.h:246-255.  The delay-line loop runs ASCENDING
 * (`comb_dly_ln[k][i] = comb_dly_ln[k][i-1]` for i = 1..M-1), so for M >= 3 the
 * old element 0 is smeared over the whole line: the effective differential
 * delay is min(M, 2).  Restated literally. */
static int64_t diff_stage(orc_cic_t *c, int64_t data_in, int k) {
  int64_t *d = c->comb_dly_ln + (size_t)k * (size_t)c->M;
  int64_t out = it_sub(c, data_in, d[c->M - 1]);
  for (int i = 0; i < c->M; i++) {
    if (i != 0) { d[i] = d[i - 1]; }
  }
  d[0] = data_in;
  return out;
}

/* comb -- ac_cic_full_core.h:228-241 */
static int64_t comb(orc_cic_t *c, int64_t data_in) {
  int64_t v = data_in;
  for (int i = 0; i < c->N; i++) { v = diff_stage(c, v, i); }
  return v;
}

static int64_t cic_run_dec(orc_cic_t *c, const int64_t *x, int64_t n_in, int64_t *y, int64_t cap) {
  /* decIntg -- ac_cic_dec_full.h:187-200 ; decIntgCore -- ac_cic_full_core.h:110-135 */
  for (int64_t t = 0; t < n_in; t++) {
    int64_t data_in_t = requant((i128)x[t], c->fi, &c->it); /* (OUT_TYPE) data_in, :114 */
    c->valid = (c->rate_cnt == 0);                         /* :116-120 */
    int64_t data_out_t = int_stage(c, data_in_t);
    c->dvalid = c->valid;
    c->rate_cnt++;                                          /* :130-133 */
    if (c->rate_cnt > (unsigned)(c->R - 1)) { c->rate_cnt = 0; }
    if (c->dvalid) { inf_write(c, data_out_t); }
  }
  /* decDiff -- ac_cic_dec_full.h:209-222 */
  int64_t n_out = 0;
  while (c->inf_size > 0) {
    int64_t d = inf_read(c);
    int64_t o = comb(c, d);                                  /* decDiffCore :198-203 */
    if (n_out >= cap) { return -1; }
    y[n_out++] = requant((i128)o, c->fi, &c->out);           /* data_out_final = data_out_t, :219 */
  }
  return n_out;
}

static int64_t cic_run_intr(orc_cic_t *c, const int64_t *x, int64_t n_in, int64_t *y, int64_t cap) {
  /* intrDiff -- ac_cic_intr_full.h:173-185 ; intrDiffCore -- ac_cic_full_core.h:211-216 */
  for (int64_t t = 0; t < n_in; t++) {
    int64_t data_in_t = requant((i128)x[t], c->fi, &c->it);
    inf_write(c, comb(c, data_in_t));
  }
  /* intrIntg -- ac_cic_intr_full.h:195-215 */
  int64_t n_out = 0;
  int64_t data_in_t = 0; /* local, re-initialised on every run() call (:196) */
  while (c->inf_size > 0) {
    if (c->dvalid) { data_in_t = inf_read(c); }
    /* intrIntgCore -- ac_cic_full_core.h:143-160 */
    int64_t stage_in;
    if (c->rate_cnt1 == (unsigned)(c->R - 1)) {
      stage_in = data_in_t; c->rate_cnt1 = 0; c->dvalid = 0;
    } else if (c->rate_cnt1 == (unsigned)(c->R - 2)) {
      stage_in = 0; c->rate_cnt1++; c->dvalid = 1;
    } else {
      stage_in = 0; c->rate_cnt1++; c->dvalid = 0;
    }
    int64_t data_out_t = int_stage(c, stage_in);
    int64_t fin = requant((i128)data_out_t, c->fi, &c->out);
    if (c->cnt < c->N - 1) {
      c->cnt++;
    } else {
      if (n_out >= cap) { return -1; }
      y[n_out++] = fin;
    }
  }
  return n_out;
}

int64_t orc_cic_run(orc_cic_t *c, const int64_t *x, int64_t n_in, int64_t *y, int64_t cap) {
  return c->interp ? cic_run_intr(c, x, n_in, y, cap) : cic_run_dec(c, x, n_in, y, cap);
}

/* ------------------------------------------------------------------ */
/* Polyphase decimator -- reference include/ac_dsp/ac_poly_dec.h        */
/* ------------------------------------------------------------------ */

struct orc_polydec {
  int32_t ntaps, df;
  orc_fmt_t in, cf, acc, out;
  int fi, fc, fa;
  int64_t *taps; /* IN_TYPE taps[NTAPS * DF]  (ac_poly_dec.h:132) */
};

orc_polydec_t *orc_polydec_new(int32_t ntaps, int32_t df, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                               const orc_fmt_t *out) {
  if (ntaps < 1 || df < 1) { return NULL; }
  orc_polydec_t *f = (orc_polydec_t *)calloc(1, sizeof *f);
  f->ntaps = ntaps; f->df = df;
  f->in = *in; f->cf = *coeff; f->acc = *acc; f->out = *out;
  f->fi = in->W - in->I; f->fc = coeff->W - coeff->I; f->fa = acc->W - acc->I;
  f->taps = (int64_t *)calloc((size_t)ntaps * (size_t)df, sizeof(int64_t)); /* init_array<AC_VAL_0>, :88 */
  return f;
}
void orc_polydec_free(orc_polydec_t *f) {
  if (!f) { return; }
  free(f->taps); free(f);
}

int64_t orc_polydec_run(orc_polydec_t *f, const int64_t *c, const int64_t *x, int64_t n_in, int64_t *y) {
  const int NT = f->ntaps, DF = f->df;
  int64_t n_out = 0, pos = 0;
  while (n_in - pos >= DF) { /* while (data_in.available(DF)) -- :109 */
    int64_t acc = 0;         /* ACC_TYPE acc, reset after every output (:127) */
    for (int df = DF - 1; df >= 0; df--) { /* :112 */
      for (int i = NT * DF - 1; i >= 0; i--) { f->taps[i] = (i == 0) ? x[pos] : f->taps[i - 1]; } /* SHIFT :114-116 */
      pos++;
      int64_t acc1 = 0;      /* acc1[df] is zero on entry: cleared after each use (:123) */
      for (int tp = 0; tp < NT; tp++) { /* MAC :118-121 */
        int fs;
        i128 s = add_aligned((i128)acc1, f->fa, (i128)f->taps[tp * DF] * c[tp + NT * df], f->fi + f->fc, &fs);
        acc1 = requant(s, fs, &f->acc);
      }
      acc = requant((i128)acc + (i128)acc1, f->fa, &f->acc); /* acc = acc + acc1[df], :122 */
    }
    y[n_out++] = requant((i128)acc, f->fa, &f->out); /* OUT_TYPE acc_t = acc, :125 */
  }
  return n_out;
}

/* ------------------------------------------------------------------ */
/* Synthetic stimulus (counter hash; the GPU generator is bit-identical) */
/* ------------------------------------------------------------------ */

/* ---- ac_poly_intr (second half of SURVEY 8 row f2): reference include/ac_dsp/ac_poly_intr.h ----
 * One core call = one input sample (run() reads one flag per call and, when it is false, calls the core once,
 * :289-320).  ftype: 0 FOLD_EVEN (:126-177), 1 FOLD_ODD (:179-236), 2 FOLD_ANTI (:238-258) -- the enum of
 * ac_poly_intr.h:71, NOT the FIR FTYPE.  State: taps[], the two accumulator banks acc_a / acc_b, flip, init (:107-123). */
struct orc_polyintr {
  int32_t ntaps, coeffsz, ifac, ftype;
  orc_fmt_t in, cf, acc, out;
  int fi, fc, fa;
  int64_t *taps, *acc_a, *acc_b;
  int flip, init;
};

orc_polyintr_t *orc_polyintr_new(int32_t ntaps, int32_t coeffsz, int32_t ifac, int32_t ftype, const orc_fmt_t *in,
                                 const orc_fmt_t *coeff, const orc_fmt_t *acc, const orc_fmt_t *out) {
  if (ntaps < 1 || coeffsz < 1 || ifac < 1 || ifac > 255 || ftype < 0 || ftype > 2) { return NULL; }
  orc_polyintr_t *f = (orc_polyintr_t *)calloc(1, sizeof *f);
  f->ntaps = ntaps; f->coeffsz = coeffsz; f->ifac = ifac; f->ftype = ftype;
  f->in = *in; f->cf = *coeff; f->acc = *acc; f->out = *out;
  f->fi = in->W - in->I; f->fc = coeff->W - coeff->I; f->fa = acc->W - acc->I;
  f->taps = (int64_t *)calloc((size_t)ntaps, sizeof(int64_t));  /* init_array<AC_VAL_0>, :117-119 */
  f->acc_a = (int64_t *)calloc((size_t)ifac, sizeof(int64_t));
  f->acc_b = (int64_t *)calloc((size_t)ifac, sizeof(int64_t));
  f->flip = 0; f->init = 0;                                      /* :120-121 */
  return f;
}
void orc_polyintr_free(orc_polyintr_t *f) {
  if (!f) { return; }
  free(f->taps); free(f->acc_a); free(f->acc_b); free(f);
}

/* Returns the number of outputs written to y (0 or IF), or -1 where the reference would index outside coeffs[] / the
 * accumulator banks. */
int64_t orc_polyintr_step(orc_polyintr_t *f, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr, int64_t x, int64_t *y) {
  const int N = f->ntaps, IF = f->ifac;
  int64_t n_out = 0;
  for (int i = N - 1; i >= 1; i--) { f->taps[i] = f->taps[i - 1]; } /* SHIFT_REG */
  for (int j = 0; j < IF; j++) {                                      /* INTR_F */
    if (j == 0) {
      f->taps[0] = x;                                                 /* data_in.read() */
      f->flip = !f->flip;
    }
    int64_t acc = 0; /* acc = 0.0 */
    if (f->ftype == 2) { /* :246-255: plain MAC, i = N-1 .. 0, output written at once */
      for (int i = N - 1; i >= 0; i--) {
        const int ci = i + N * j;
        if (ci >= f->coeffsz) { return -1; }
        int fs;
        i128 s = add_aligned((i128)acc, f->fa, (i128)f->taps[i] * coeffs[ci], f->fi + f->fc, &fs);
        acc = requant(s, fs, &f->acc);
      }
      y[n_out++] = requant((i128)acc, f->fa, &f->out);
      continue;
    }
    if (f->ftype == 0) { /* MAC_E :141-151: i = N/2-1 .. 0 */
      for (int i = (N / 2) - 1; i >= 0; i--) {
        /* IN_TYPE tp = sign[j] ? taps[N-1-i] : -taps[N-1-i]  (the negation is assigned to an IN_TYPE) */
        int64_t tp = sign[j] ? f->taps[N - 1 - i] : requant(-(i128)f->taps[N - 1 - i], f->fi, &f->in);
        int64_t fold = requant((i128)f->taps[i] + (i128)tp, f->fi, &f->acc); /* ACC_TYPE fold */
        const int ci = i + j * N / 2;                                         /* C precedence: (j * NTAPS) / 2 */
        if (ci >= f->coeffsz) { return -1; }
        int fs;
        i128 s = add_aligned((i128)acc, f->fa, (i128)coeffs[ci] * (i128)fold, f->fc + f->fa, &fs);
        acc = requant(s, fs, &f->acc);
      }
    } else { /* MAC_O :194-209: i = 0 .. (N-1)/2, centre tap passes through */
      for (int i = 0; i < ((N - 1) / 2) + 1; i++) {
        int64_t fold;
        if (i == (N - 1) / 2) {
          fold = requant((i128)f->taps[i], f->fi, &f->acc);
        } else {
          int64_t tp = sign[j] ? f->taps[N - 1 - i] : requant(-(i128)f->taps[N - 1 - i], f->fi, &f->in);
          fold = requant((i128)f->taps[i] + (i128)tp, f->fi, &f->acc);
        }
        const int ci = i + (N / 2 + 1) * j;
        if (ci >= f->coeffsz) { return -1; }
        int fs;
        i128 s = add_aligned((i128)acc, f->fa, (i128)coeffs[ci] * (i128)fold, f->fc + f->fa, &fs);
        acc = requant(s, fs, &f->acc);
      }
    }
    if (corr[j] >= IF) { return -1; }
    int64_t t1, t2; /* :153-161 / :211-221: this sample's sums go to one bank, the outputs come from the other */
    if (f->flip) {
      f->acc_b[j] = acc; t1 = f->acc_a[j]; t2 = f->acc_a[corr[j]];
    } else {
      f->acc_a[j] = acc; t1 = f->acc_b[j]; t2 = f->acc_b[corr[j]];
    }
    if (f->init) {
      if (j != corr[j]) { /* symmetric-pair technique :163-171 */
        int64_t tn = sign[j] ? requant(-(i128)t2, f->fa, &f->acc) : t2; /* ACC_TYPE tn */
        i128 sum = (i128)t1 + (i128)tn;                                    /* exact sum type */
        i128 half = sum >> 1;                                              /* ac_fixed >> 1: same type, LSB dropped */
        y[n_out++] = requant(half, f->fa, &f->out);
      } else {
        y[n_out++] = requant((i128)t1, f->fa, &f->out);
      }
    }
  }
  f->init = 1;
  return n_out;
}

int64_t orc_polyintr_run(orc_polyintr_t *f, const int64_t *coeffs, const uint8_t *sign, const uint8_t *corr, const int64_t *x,
                         int64_t n_in, int64_t *y) {
  int64_t k = 0;
  for (int64_t t = 0; t < n_in; t++) {
    int64_t r = orc_polyintr_step(f, coeffs, sign, corr, x[t], y + k);
    if (r < 0) { return -1; }
    k += r;
  }
  return k;
}

/* ---- ac_intg_dump: reference include/ac_dsp/ac_intg_dump.h:77-151 ----
 * State: ACC_TYPE temp[CHN], zero-initialised (:84-90).  One block of run() (:133-147): read n_sample, then for
 * j = 1..NS and every channel i read one sample, temp[i] = ACC(temp[i] + sample) (:97), and when j == n_sample write
 * OUT(temp[i]) and clear temp[i] (:98-102); the block ends after the channel loop in which that happened (:144) -- or
 * after NS rounds without any output when n_sample is 0 or > NS (the sums then carry into the next block).
 * x is the interleaved stream (round-major, channel-minor); returns the number of outputs, *used = samples consumed. */
int64_t orc_intg_dump_run(int64_t *temp, int32_t ns, int32_t chn, const orc_fmt_t *in, const orc_fmt_t *acc, const orc_fmt_t *out,
                          const int64_t *n_sample, int64_t n_blocks, const int64_t *x, int64_t *y, int64_t *used) {
  const int fi = in->W - in->I, fa = acc->W - acc->I;
  int64_t k = 0, pos = 0;
  for (int64_t b = 0; b < n_blocks; b++) {
    const int64_t n_sample_t = n_sample[b];
    int flag = 0;
    for (int j = 1; j <= ns; j++) {
      for (int i = 0; i < chn; i++) {
        const int64_t data_in_t = x[pos++];
        int fs;
        i128 s = add_aligned((i128)temp[i], fa, (i128)data_in_t, fi, &fs);
        temp[i] = requant(s, fs, acc);
        if (j == n_sample_t) {
          y[k++] = requant((i128)temp[i], fa, out);
          temp[i] = 0;
          flag = 1;
        }
      }
      if (flag) { break; }
    }
  }
  if (used) { *used = pos; }
  return k;
}

/* ------------------------------------------------------------------ */
/* moving average (row f4 of SURVEY 8: reference include/ac_dsp/ac_mv_avg.h:93-196)                       */
/* ------------------------------------------------------------------ */
/* ac_window_1d_flag<IN_TYPE, TAPS, WIN_TYPE> as ac_mv_avg drives it (write / valid / operator[], :111-123), kept as the
 * literal shift register plus the sol / eol flags travelling with the samples.  The class belongs to ac_types, which is not
 * in the image: the boundary behaviour is restated from its documented modes -- PARITY UNPINNED (acdsp_oracle.h). */
typedef struct {
  int taps, mode;
  int64_t data[ORC_MVAVG_MAX_TAPS];
  uint8_t sol[ORC_MVAVG_MAX_TAPS], eol[ORC_MVAVG_MAX_TAPS], live[ORC_MVAVG_MAX_TAPS];
} mv_window_t;

static void win_write(mv_window_t *w, int64_t v, int sol, int eol) {
  for (int i = 0; i + 1 < w->taps; i++) { w->data[i] = w->data[i + 1]; w->sol[i] = w->sol[i + 1]; w->eol[i] = w->eol[i + 1]; w->live[i] = w->live[i + 1]; }
  const int last = w->taps - 1;
  w->data[last] = v; w->sol[last] = (uint8_t)sol; w->eol[last] = (uint8_t)eol; w->live[last] = 1;
}
/* slot of the current line's first sample (or -1: it has left the window) and of its last sample (or -1: not seen yet),
 * looking outwards from the centre slot c: the line of the centre sample starts at the nearest sol at or left of c */
static void win_line(const mv_window_t *w, int *first, int *last, int *centre_ok) {
  const int c = w->taps / 2;
  *first = -1; *last = -1; *centre_ok = w->live[c];
  for (int i = c; i >= 0; i--) {
    if (!w->live[i]) { break; }                                            /* never written: the line starts to the right of it */
    if (i < c && w->eol[i]) { *first = i + 1; *centre_ok = 0; break; }   /* the centre is filler written after a closed line */
    if (w->sol[i]) { *first = i; break; }
  }
  for (int i = c; i < w->taps; i++) {
    if (w->eol[i]) { *last = i; break; }
    if (i > c && w->sol[i]) { *last = i - 1; break; }
  }
}
static int win_valid(const mv_window_t *w) {
  int first, last, ok;
  win_line(w, &first, &last, &ok);
  if (!ok) { return 0; }
  if (w->mode == ORC_WIN_PLAIN) {   /* every slot belongs to the centre's line */
    const int lo = first < 0 ? 0 : first, hi = last < 0 ? w->taps - 1 : last;
    if (!w->live[0]) { return 0; }
    return lo == 0 && hi == w->taps - 1;
  }
  return 1;
}
static int64_t win_at(const mv_window_t *w, int j) {
  const int c = w->taps / 2;
  int first, last, ok;
  win_line(w, &first, &last, &ok);
  int i = c + j;
  if (w->mode != ORC_WIN_PLAIN) {
    /* first / last = -1 means the boundary is outside the window: nothing to fold on that side */
    for (int guard = 0; guard < 4 * w->taps; guard++) {
      if (first >= 0 && i < first) { i = (w->mode == ORC_WIN_CLIP) ? first : 2 * first - i; continue; }
      if (last >= 0 && i > last) { i = (w->mode == ORC_WIN_CLIP) ? last : 2 * last - i; continue; }
      break;
    }
    if (first >= 0 && last >= 0 && first == last) { i = first; }
  }
  if (i < 0) { i = 0; }
  if (i >= w->taps) { i = w->taps - 1; }
  return w->data[i];
}

/* One run() call of one ac_mv_avg object (ac_mv_avg.h:146-190): n_frames frames of n_sample inputs each (the n_sample
 * word read at :155 applies to every frame of the call); a fresh core object per call (:154), so nothing carries over.
 * Per valid window position (:113-121):  acc = 0; for j = -TAPS/2 .. TAPS/2: acc = ACC(acc + ACC(w[j]) * c[j + TAPS/2]);
 * out = OUT(acc).  Returns the outputs written, -1 on parameters the reference cannot run (even TAPS read c[TAPS]). */
int64_t orc_mv_avg_run(int32_t taps, int32_t win_mode, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                       const orc_fmt_t *out, const int64_t *c, const int64_t *x, int64_t n_sample, int64_t n_frames, int64_t *y) {
  if (taps < 1 || taps > ORC_MVAVG_MAX_TAPS || !(taps & 1) || n_sample < 1) { return -1; }
  const int fi = in->W - in->I, fc = coeff->W - coeff->I, fa = acc->W - acc->I;
  const int h = taps / 2;
  mv_window_t w;
  memset(&w, 0, sizeof w);
  w.taps = taps; w.mode = win_mode;
  int64_t k = 0, pos = 0;
  for (int64_t f = 0; f < n_frames; f++) {
    const int64_t sample = (win_mode == ORC_WIN_PLAIN) ? n_sample : n_sample + h;   /* :158-162 */
    int64_t data_in_t = 0;
    for (int64_t cnt = 0; cnt < sample; cnt++) {
      if (cnt < n_sample) { data_in_t = x[pos++]; }                                  /* :174-176 */
      win_write(&w, data_in_t, cnt == 0, cnt == n_sample - 1);                       /* :177-180, :112 */
      if (win_valid(&w)) {
        int64_t a = 0;
        for (int j = -h; j <= h; j++) {
          const int64_t xq = requant((i128)win_at(&w, j), fi, acc);                  /* (ACC_TYPE) w[j] */
          int f_sum;
          const i128 sum = add_aligned((i128)a, fa, (i128)xq * (i128)c[j + h], fa + fc, &f_sum);
          a = requant(sum, f_sum, acc);
        }
        y[k++] = requant((i128)a, fa, out);
      }
    }
  }
  return k;
}

uint64_t orc_splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int64_t orc_stimulus(uint64_t seed, uint64_t ch, uint64_t t, int32_t bits) {
  uint64_t z = orc_splitmix64(seed, (ch << 32) | (t & 0xffffffffull));
  return ((int64_t)z) >> (64 - bits);
}
