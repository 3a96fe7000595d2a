/* acdsp_oracle_wide.cpp -- the FIR / CIC restatement of acdsp_oracle.c for types WIDER than 64 bits.
 * TEST INFRASTRUCTURE ONLY (see acdsp_oracle.h): nothing under ac_dsp_amd/ or include/ may call, link or import this.
 *
 * The reference's templates take ac_fixed of any width -- ac_cic_dec_full derives INT_TYPE = W + N log2(R M) bits
 * (include/ac_dsp/ac_cic_dec_full.h:116-137; ac_cic_intr_full.h:107-127) and the FIR cores accumulate into whatever ACC_TYPE the
 * user names (ac_fir_const_coeffs.h:190-296) -- so the same per-sample loops are restated here on 128-bit raw words (IN / COEFF
 * <= 64 bits; ACC / OUT / INT_TYPE <= 128 bits) with 256-bit exact intermediates.  The loops are written a second time on
 * purpose (C++ with an own 256-bit integer, independent of acdsp_oracle.c and of the product's wide_int.hpp): on formats of
 * <= 64 bits tests/test_wide_cpu.py requires this file and acdsp_oracle.c to agree word for word, and a pure-Python big-integer
 * model checks both on small cases.
 *
 * PINNING: parity of the > 64-bit widths rests on (a) that agreement at <= 64 bits, where acdsp_oracle.c is pinned by the
 * reference's vectors as its header says, (b) the Python model, (c) the golden vectors the reference's own headers produce
 * over this repo's ac_types subset at 72 / 96 bits (tests/golden/ref_hdr/wide.json).  No reference artefact exists for these
 * widths beyond that (the reference ships no test wider than <64,32>).
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "acdsp_oracle.h"

namespace {

typedef __int128 i128;
typedef unsigned __int128 u128;

/* 256-bit two's complement: value = hi * 2^128 + lo */
struct W256 {
  u128 lo;
  i128 hi;
  W256() : lo(0), hi(0) {}
  W256(i128 v) : lo((u128)v), hi(v < 0 ? (i128)-1 : (i128)0) {}
  bool neg() const { return hi < 0; }
};

W256 operator+(const W256 &a, const W256 &b) {
  W256 r;
  r.lo = a.lo + b.lo;
  r.hi = (i128)((u128)a.hi + (u128)b.hi + (r.lo < a.lo ? 1 : 0));
  return r;
}
W256 operator-(const W256 &a) {
  W256 r;
  r.lo = ~a.lo + 1;
  r.hi = (i128)(~(u128)a.hi + (r.lo == 0 ? 1 : 0));
  return r;
}
W256 operator-(const W256 &a, const W256 &b) { return a + (-b); }
bool operator<(const W256 &a, const W256 &b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
bool operator>(const W256 &a, const W256 &b) { return b < a; }

/* |a| * |b| on 64-bit limbs, low 256 bits, then the sign */
W256 operator*(const W256 &a, const W256 &b) {
  const bool sa = a.neg(), sb = b.neg();
  const W256 ua = sa ? -a : a, ub = sb ? -b : b;
  const uint64_t x[4] = {(uint64_t)ua.lo, (uint64_t)(ua.lo >> 64), (uint64_t)(u128)ua.hi, (uint64_t)((u128)ua.hi >> 64)};
  const uint64_t y[4] = {(uint64_t)ub.lo, (uint64_t)(ub.lo >> 64), (uint64_t)(u128)ub.hi, (uint64_t)((u128)ub.hi >> 64)};
  uint64_t z[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 carry = 0;
    for (int j = 0; i + j < 4; j++) {
      const u128 t = (u128)x[i] * y[j] + z[i + j] + carry;
      z[i + j] = (uint64_t)t;
      carry = t >> 64;
    }
  }
  W256 r;
  r.lo = ((u128)z[1] << 64) | z[0];
  r.hi = (i128)(((u128)z[3] << 64) | z[2]);
  return sa != sb ? -r : r;
}

W256 shl(const W256 &a, int s) {
  W256 r;
  if (s <= 0) { return s == 0 ? a : r; }
  if (s >= 256) { return r; }
  if (s >= 128) { r.hi = (i128)(a.lo << (s - 128)); r.lo = 0; return r; }
  r.lo = a.lo << s;
  r.hi = (i128)(((u128)a.hi << s) | (a.lo >> (128 - s)));
  return r;
}
/* floor(a / 2^s) */
W256 sar(const W256 &a, int s) {
  W256 r;
  if (s <= 0) { return a; }
  if (s >= 256) { r.hi = a.neg() ? -1 : 0; r.lo = a.neg() ? ~(u128)0 : 0; return r; }
  if (s >= 128) { r.lo = (u128)(a.hi >> (s - 128)); r.hi = a.neg() ? -1 : 0; return r; }
  r.lo = (a.lo >> s) | ((u128)a.hi << (128 - s));
  r.hi = a.hi >> s;
  return r;
}
int bit_of(const W256 &a, int k) {
  if (k < 0) { return 0; }
  if (k >= 256) { return a.neg() ? 1 : 0; }
  return k < 128 ? (int)((a.lo >> k) & 1) : (int)(((u128)a.hi >> (k - 128)) & 1);
}
/* OR of bits [0, k) */
int any_below(const W256 &a, int k) {
  if (k <= 0) { return 0; }
  if (k >= 256) { return (a.lo != 0 || a.hi != 0) ? 1 : 0; }
  if (k <= 128) { return (k == 128 ? a.lo : (a.lo & (((u128)1 << k) - 1))) != 0 ? 1 : 0; }
  return (a.lo != 0 || ((u128)a.hi & (((u128)1 << (k - 128)) - 1)) != 0) ? 1 : 0;
}

/* ---- ac_fixed conversion rules (AC Datatypes semantics, restated as in acdsp_oracle.c) ---- */
i128 fmt_min(const orc_fmt_t *f) { return f->S ? (i128)(~(u128)0 << (f->W - 1)) : (i128)0; }   /* (not -(1 << (W-1)): W = 128 negates INT128_MIN -- found by the UBSan build) */
i128 fmt_max(const orc_fmt_t *f) { return f->S ? (i128)(((u128)1 << (f->W - 1)) - 1) : (i128)(((u128)1 << f->W) - 1); }

i128 wrap_w(const W256 &q, int W, int S) {
  if (W >= 128) { return (i128)q.lo; }
  const u128 m = ((u128)1 << W) - 1;
  u128 u = q.lo & m;
  if (S && ((u >> (W - 1)) & 1)) { u |= ~m; }
  return (i128)u;
}

W256 quantize(const W256 &x, int f_src, int f_dst, int Q) {
  const int sh = f_src - f_dst;
  if (sh <= 0) { return shl(x, -sh); }
  const int neg = x.neg() ? 1 : 0;
  const int qb = bit_of(x, sh - 1);        /* most significant dropped bit */
  const int r = any_below(x, sh - 1);      /* OR of the remaining dropped bits */
  W256 q = sar(x, sh);
  const int lsb = bit_of(q, 0);
  int inc = 0;
  switch (Q) {
    case ORC_TRN: inc = 0; break;
    case ORC_RND: inc = qb; break;
    case ORC_TRN_ZERO: inc = neg && (qb || r); break;
    case ORC_RND_ZERO: inc = qb && (r || neg); break;
    case ORC_RND_INF: inc = qb && (r || !neg); break;
    case ORC_RND_MIN_INF: inc = qb && r; break;
    case ORC_RND_CONV: inc = qb && (r || lsb); break;
    case ORC_RND_CONV_ODD: inc = qb && (r || !lsb); break;
    default: break;
  }
  return inc ? q + W256((i128)1) : q;
}

i128 overflow(const W256 &q, const orc_fmt_t *f) {
  const i128 lo = fmt_min(f), hi = fmt_max(f);
  const bool under = q < W256(lo), over = q > W256(hi);
  switch (f->O) {
    case ORC_WRAP: return wrap_w(q, f->W, f->S);
    case ORC_SAT: return under ? lo : (over ? hi : (i128)q.lo);
    case ORC_SAT_ZERO: return (under || over) ? (i128)0 : (i128)q.lo;
    case ORC_SAT_SYM:
      if (f->S) {
        if (under || over) { return q.neg() ? lo + 1 : hi; }
        return ((i128)q.lo == lo && f->W > 1) ? lo + 1 : (i128)q.lo;
      }
      return under ? lo : (over ? hi : (i128)q.lo);
    default: return (i128)q.lo;
  }
}

i128 requant(const W256 &x, int f_src, const orc_fmt_t *dst) { return overflow(quantize(x, f_src, dst->W - dst->I, dst->Q), dst); }

/* exact a*2^-fa + b*2^-fb at max(fa, fb) fractional bits */
W256 add_aligned(const W256 &a, int fa, const W256 &b, int fb, int *f_out) {
  const int f = fa > fb ? fa : fb;
  *f_out = f;
  return shl(a, f - fa) + shl(b, f - fb);
}

bool fmt_ok(const orc_fmt_t *f, int max_w) { return f->W >= 1 && f->W <= max_w && (f->S || (f->W != 64 && f->W != 128)); }

/* ---- FIR cores: ac_fir_const_coeffs.h:153-296 (load_ / prog_ twins: ac_fir_load_coeffs.h:145-278, ac_fir_prog_coeffs.h:110-247) ---- */
struct WFir {
  int n, ftype;
  orc_fmt_t in, cf, acc, out;
  int fi, fc, fa;
  std::vector<int64_t> reg;      /* IN_TYPE reg[N_TAPS]        (:124) */
  std::vector<i128> reg_trans;   /* ACC_TYPE reg_trans[N_TAPS] (:125) */
  int wptr;                      /* (:126) */
};

void fir_shift_reg(WFir *f, int64_t din) { /* :153-159 */
  for (int i = f->n - 1; i >= 0; i--) { f->reg[(size_t)i] = (i == 0) ? din : f->reg[(size_t)i - 1]; }
}

/* `acc += a * b`: exact product, exact sum with acc, then ACC_TYPE quantise / overflow */
i128 mac(const WFir *f, i128 acc, const W256 &prod, int f_prod) {
  int fs;
  const W256 s = add_aligned(W256(acc), f->fa, prod, f_prod, &fs);
  return requant(s, fs, &f->acc);
}

int fir_step(WFir *f, const int64_t *c, int64_t x, i128 *y) {
  const int N = f->n;
  i128 acc = 0;
  switch (f->ftype) {
    case ORC_SHIFT_REG: /* :190-199 */
      fir_shift_reg(f, x);
      for (int i = N - 1; i >= 0; i--) { acc = mac(f, acc, W256((i128)f->reg[(size_t)i]) * W256((i128)c[i]), f->fi + f->fc); }
      break;
    case ORC_ROTATE_SHIFT: { /* :205-220 */
      int64_t temp_rotate;
      for (int i = N; i >= 0; i--) {
        if (i == N) {
          temp_rotate = x;
        } else {
          temp_rotate = f->reg[(size_t)N - 1];
          acc = mac(f, acc, W256((i128)f->reg[(size_t)N - 1]) * W256((i128)c[i]), f->fi + f->fc);
        }
        fir_shift_reg(f, temp_rotate);
      }
      break;
    }
    case ORC_C_BUFF: /* :226-237 with :165-184 */
      for (int i = 0; i <= N - 1; i++) {
        if (i == 0) {
          f->reg[(size_t)f->wptr] = x;
          if (f->wptr == N - 1) { f->wptr = 0; } else { f->wptr++; }
        }
        int rptr = f->wptr - 1 - i;
        if (rptr < 0) { rptr += N; }
        acc = mac(f, acc, W256((i128)f->reg[(size_t)rptr]) * W256((i128)c[i]), f->fi + f->fc);
      }
      break;
    case ORC_FOLD_EVEN: /* :244-253: exact pre-add */
      fir_shift_reg(f, x);
      for (int i = (N / 2) - 1; i >= 0; i--) {
        const W256 pre = W256((i128)f->reg[(size_t)i]) + W256((i128)f->reg[(size_t)(N - 1 - i)]);
        acc = mac(f, acc, W256((i128)c[i]) * pre, f->fc + f->fi);
      }
      break;
    case ORC_FOLD_ODD: /* :260-275: `fold` is an ACC_TYPE variable */
      fir_shift_reg(f, x);
      for (int i = 0; i < ((N - 1) / 2) + 1; i++) {
        i128 fold;
        if (i == (N - 1) / 2) {
          fold = requant(W256((i128)f->reg[(size_t)i]), f->fi, &f->acc);
        } else {
          fold = requant(W256((i128)f->reg[(size_t)i]) + W256((i128)f->reg[(size_t)((N - 1) - i)]), f->fi, &f->acc);
        }
        acc = mac(f, acc, W256((i128)c[i]) * W256(fold), f->fc + f->fa);
      }
      break;
    case ORC_TRANSPOSED: /* :281-296 */
      for (int i = N - 1; i >= 0; i--) {
        const i128 temp = (i == 0) ? (i128)0 : f->reg_trans[(size_t)i - 1];
        int fs;
        const W256 s = add_aligned(W256((i128)x) * W256((i128)c[(N - 1) - i]), f->fi + f->fc, W256(temp), f->fa, &fs);
        f->reg_trans[(size_t)i] = requant(s, fs, &f->acc);
      }
      acc = f->reg_trans[(size_t)N - 1];
      break;
    default: /* FOLD_*_ANTI: run() has no branch (:330-352) */
      return -1;
  }
  *y = requant(W256(acc), f->fa, &f->out); /* data_out = acc */
  return 0;
}

/* ---- CIC: ac_cic_full_core.h:80-160,198-255; ac_cic_dec_full.h:187-222; ac_cic_intr_full.h:173-215 ---- */
struct WCic {
  int interp, R, M, N;
  orc_fmt_t in, out, it;
  int fi;
  int valid, dvalid;
  uint8_t rate_cnt1, rate_cnt, cnt; /* ac_int<8,false> (ac_cic_full_core.h:72-74) */
  std::vector<i128> intg_reg;       /* [N] */
  std::vector<i128> comb_dly_ln;    /* [N][M] (:219) */
  std::vector<i128> inf;            /* ac_channel<INT_TYPE> inf (ac_cic_dec_full.h:177) */
  size_t inf_head;
};

i128 it_add(const WCic *c, i128 a, i128 b) { return requant(W256(a) + W256(b), c->fi, &c->it); }
i128 it_sub(const WCic *c, i128 a, i128 b) { return requant(W256(a) - W256(b), c->fi, &c->it); }

i128 int_stage(WCic *c, i128 data_in) { /* :80-87, pipelined */
  for (int i = c->N - 1; i > 0; i--) { c->intg_reg[(size_t)i] = it_add(c, c->intg_reg[(size_t)i], c->intg_reg[(size_t)i - 1]); }
  c->intg_reg[0] = it_add(c, data_in, c->intg_reg[0]);
  return c->intg_reg[(size_t)c->N - 1];
}
i128 diff_stage(WCic *c, i128 data_in, int k) { /* :246-255, ascending delay-line loop restated literally */
  i128 *d = c->comb_dly_ln.data() + (size_t)k * (size_t)c->M;
  const i128 out = it_sub(c, data_in, d[c->M - 1]);
  for (int i = 0; i < c->M; i++) {
    if (i != 0) { d[i] = d[i - 1]; }
  }
  d[0] = data_in;
  return out;
}
i128 comb(WCic *c, i128 data_in) { /* :228-241 */
  i128 v = data_in;
  for (int i = 0; i < c->N; i++) { v = diff_stage(c, v, i); }
  return v;
}

int64_t cic_run_dec(WCic *c, const int64_t *x, int64_t n_in, i128 *y, int64_t cap) {
  for (int64_t t = 0; t < n_in; t++) { /* decIntg :187-200; decIntgCore ac_cic_full_core.h:110-135 */
    const i128 data_in_t = requant(W256((i128)x[t]), c->fi, &c->it);
    c->valid = (c->rate_cnt == 0);
    const i128 data_out_t = int_stage(c, data_in_t);
    c->dvalid = c->valid;
    c->rate_cnt++;
    if (c->rate_cnt > (unsigned)(c->R - 1)) { c->rate_cnt = 0; }
    if (c->dvalid) { c->inf.push_back(data_out_t); }
  }
  int64_t n_out = 0;
  while (c->inf_head < c->inf.size()) { /* decDiff :209-222 */
    const i128 o = comb(c, c->inf[c->inf_head++]);
    if (n_out >= cap) { return -1; }
    y[n_out++] = requant(W256(o), c->fi, &c->out);
  }
  c->inf.clear(); c->inf_head = 0;
  return n_out;
}

int64_t cic_run_intr(WCic *c, const int64_t *x, int64_t n_in, i128 *y, int64_t cap) {
  for (int64_t t = 0; t < n_in; t++) { /* intrDiff ac_cic_intr_full.h:173-185 */
    c->inf.push_back(comb(c, requant(W256((i128)x[t]), c->fi, &c->it)));
  }
  int64_t n_out = 0;
  i128 data_in_t = 0; /* local, re-initialised on every run() call (:196) */
  while (c->inf_head < c->inf.size()) { /* intrIntg :195-215 */
    if (c->dvalid) { data_in_t = c->inf[c->inf_head++]; }
    i128 stage_in; /* intrIntgCore ac_cic_full_core.h:143-160 */
    if (c->rate_cnt1 == (unsigned)(c->R - 1)) {
      stage_in = data_in_t; c->rate_cnt1 = 0; c->dvalid = 0;
    } else if (c->rate_cnt1 == (unsigned)(c->R - 2)) {
      stage_in = 0; c->rate_cnt1++; c->dvalid = 1;
    } else {
      stage_in = 0; c->rate_cnt1++; c->dvalid = 0;
    }
    const i128 fin = requant(W256(int_stage(c, stage_in)), c->fi, &c->out);
    if (c->cnt < c->N - 1) {
      c->cnt++;
    } else {
      if (n_out >= cap) { return -1; }
      y[n_out++] = fin;
    }
  }
  c->inf.clear(); c->inf_head = 0;
  return n_out;
}

void put(orcw_word_t *w, i128 v) { w->lo = (uint64_t)(u128)v; w->hi = (int64_t)(v >> 64); }

}  // namespace

extern "C" {

orcw_fir_t *orcw_fir_new(int32_t n_taps, int32_t ftype, const orc_fmt_t *in, const orc_fmt_t *coeff, const orc_fmt_t *acc,
                         const orc_fmt_t *out) {
  if (n_taps < 1 || !fmt_ok(in, 64) || !fmt_ok(coeff, 64) || !fmt_ok(acc, 128) || !fmt_ok(out, 128)) { return NULL; }
  WFir *f = new WFir();
  f->n = n_taps; f->ftype = ftype;
  f->in = *in; f->cf = *coeff; f->acc = *acc; f->out = *out;
  f->fi = in->W - in->I; f->fc = coeff->W - coeff->I; f->fa = acc->W - acc->I;
  f->reg.assign((size_t)n_taps, 0);
  f->reg_trans.assign((size_t)n_taps, (i128)0);
  f->wptr = 0;
  return (orcw_fir_t *)f;
}
void orcw_fir_free(orcw_fir_t *f) { delete (WFir *)f; }

int32_t orcw_fir_run(orcw_fir_t *fp, const int64_t *coeffs, const int64_t *x, int64_t n, orcw_word_t *y) {
  WFir *f = (WFir *)fp;
  for (int64_t t = 0; t < n; t++) {
    i128 v;
    if (fir_step(f, coeffs, x[t], &v)) { return -1; }
    put(&y[t], v);
  }
  return 0;
}

orcw_cic_t *orcw_cic_new(int32_t interp, int32_t R, int32_t M, int32_t N, const orc_fmt_t *in, const orc_fmt_t *out) {
  orc_fmt_t it;
  if (orc_cic_int_type(interp, R, M, N, in, &it) || it.W > 128 || !fmt_ok(in, 64) || !fmt_ok(out, 128)) { return NULL; }
  WCic *c = new WCic();
  c->interp = interp; c->R = R; c->M = M; c->N = N;
  c->in = *in; c->out = *out; c->it = it; c->fi = in->W - in->I;
  c->valid = 1; c->rate_cnt = 0; c->dvalid = 1; c->cnt = 0; c->rate_cnt1 = (uint8_t)(R - 1); /* ac_cic_full_core.h:94-102 */
  c->intg_reg.assign((size_t)N, (i128)0);
  c->comb_dly_ln.assign((size_t)N * (size_t)M, (i128)0);
  c->inf_head = 0;
  return (orcw_cic_t *)c;
}
void orcw_cic_free(orcw_cic_t *c) { delete (WCic *)c; }

int64_t orcw_cic_run(orcw_cic_t *cp, const int64_t *x, int64_t n_in, orcw_word_t *y, int64_t cap) {
  WCic *c = (WCic *)cp;
  std::vector<i128> tmp((size_t)(cap > 0 ? cap : 1));
  const int64_t k = c->interp ? cic_run_intr(c, x, n_in, tmp.data(), cap) : cic_run_dec(c, x, n_in, tmp.data(), cap);
  for (int64_t i = 0; i < k; i++) { put(&y[i], tmp[(size_t)i]); }
  return k;
}

/* exact x * 2^-f_src, x a 256-bit little-endian two's-complement integer -> raw word of *dst */
int32_t orcw_requant(const uint64_t x[4], int32_t f_src, const orc_fmt_t *dst, orcw_word_t *out) {
  if (!fmt_ok(dst, 128)) { return -1; }
  W256 v;
  v.lo = ((u128)x[1] << 64) | x[0];
  v.hi = (i128)(((u128)x[3] << 64) | x[2]);
  put(out, requant(v, f_src, dst));
  return 0;
}

}  // extern "C"
