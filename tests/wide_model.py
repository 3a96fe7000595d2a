"""Pure-Python big-integer model of the ac_fixed conversion rules and of the reference's FIR / CIC loops, for SMALL cases.

Third, independent statement of what oracle/acdsp_oracle.c, oracle/acdsp_oracle_wide.cpp and the kernels implement: Python ints
have no width, so every exact product / aligned sum of the reference (`acc += reg[i] * coeffs[i]`, reference
include/ac_dsp/ac_fir_const_coeffs.h:196; `intg_reg[i] += intg_reg[i-1]`, ac_cic_full_core.h:80-87) is formed literally and then
converted with the AC Datatypes rule "quantise with the destination's Q mode, then apply its O mode".  Test infrastructure."""
from fractions import Fraction  # noqa: F401  (kept for interactive checks)

Q = {"TRN": 0, "RND": 1, "TRN_ZERO": 2, "RND_ZERO": 3, "RND_INF": 4, "RND_MIN_INF": 5, "RND_CONV": 6, "RND_CONV_ODD": 7}
O = {"WRAP": 0, "SAT": 1, "SAT_ZERO": 2, "SAT_SYM": 3}


class F:
    def __init__(self, W, I, S=True, Q="TRN", O="WRAP"):
        self.W, self.I, self.S, self.Q, self.O = W, I, bool(S), Q, O
        self.F = W - I
        self.lo = -(1 << (W - 1)) if S else 0
        self.hi = (1 << (W - 1)) - 1 if S else (1 << W) - 1


def requant(x, f_src, d):
    """exact x * 2^-f_src -> raw word of d"""
    sh = f_src - d.F
    if sh <= 0:
        q = x << (-sh)
    else:
        q = x >> sh                                    # floor
        rem = x - (q << sh)
        half = 1 << (sh - 1)
        qb, r, neg, lsb = rem >= half, (rem & (half - 1)) != 0, x < 0, q & 1
        inc = {"TRN": False, "RND": qb, "TRN_ZERO": neg and (qb or r), "RND_ZERO": qb and (r or neg), "RND_INF": qb and (r or not neg),
               "RND_MIN_INF": qb and r, "RND_CONV": qb and (r or lsb), "RND_CONV_ODD": qb and (r or not lsb)}[d.Q]
        q += 1 if inc else 0
    under, over = q < d.lo, q > d.hi
    if d.O == "WRAP":
        u = q & ((1 << d.W) - 1)
        return u - (1 << d.W) if d.S and (u >> (d.W - 1)) & 1 else u
    if d.O == "SAT":
        return d.lo if under else (d.hi if over else q)
    if d.O == "SAT_ZERO":
        return 0 if (under or over) else q
    if d.S:                                            # SAT_SYM
        if under or over:
            return d.lo + 1 if q < 0 else d.hi
        return d.lo + 1 if (q == d.lo and d.W > 1) else q
    return d.lo if under else (d.hi if over else q)


def mac(acc, prod, f_prod, A):
    f = max(f_prod, A.F)
    return requant((acc << (f - A.F)) + (prod << (f - f_prod)), f, A)


def fir(ftype, c, x, fin, fc, fa, fo):
    """One channel; the loops of ac_fir_const_coeffs.h:190-296 in the reference's order."""
    N = len(c)
    reg, rt, wptr, y = [0] * N, [0] * N, 0, []
    fp = fin.F + fc.F
    for s in x:
        acc = 0
        if ftype in ("SHIFT_REG", "ROTATE_SHIFT", "FOLD_EVEN", "FOLD_ODD"):
            reg = [s] + reg[:-1]
        if ftype in ("SHIFT_REG", "ROTATE_SHIFT"):
            for i in range(N - 1, -1, -1):
                acc = mac(acc, reg[i] * c[i], fp, fa)
        elif ftype == "C_BUFF":
            reg[wptr] = s
            wptr = 0 if wptr == N - 1 else wptr + 1
            for i in range(N):
                acc = mac(acc, reg[(wptr - 1 - i) % N] * c[i], fp, fa)
        elif ftype == "FOLD_EVEN":
            for i in range(N // 2 - 1, -1, -1):
                acc = mac(acc, c[i] * (reg[i] + reg[N - 1 - i]), fp, fa)
        elif ftype == "FOLD_ODD":
            mid = (N - 1) // 2
            for i in range(mid + 1):
                fold = requant(reg[i] if i == mid else reg[i] + reg[N - 1 - i], fin.F, fa)
                acc = mac(acc, c[i] * fold, fc.F + fa.F, fa)
        elif ftype == "TRANSPOSED":
            for i in range(N - 1, -1, -1):
                rt[i] = mac(rt[i - 1] if i else 0, s * c[N - 1 - i], fp, fa)
            acc = rt[N - 1]
        y.append(requant(acc, fa.F, fo))
    return y


def cic_int_type(interp, R, M, N, fin):
    p = R ** (N - 1 if interp else N) * M ** N
    w = (p - 1).bit_length() + fin.W + (0 if fin.S else 1)
    return F(w, w - fin.F, True, "TRN", "WRAP")


def cic(interp, R, M, N, x, fin, fo, state=None):
    """One run() call of ac_cic_dec_full / ac_cic_intr_full on one channel (state dict carried between calls)."""
    it = cic_int_type(interp, R, M, N, fin)
    st = state if state is not None else {"ig": [0] * N, "dl": [[0] * M for _ in range(N)], "rc": 0, "rc1": R - 1, "dvalid": True, "cnt": 0}

    def stage(v):
        ig = st["ig"]
        for i in range(N - 1, 0, -1):
            ig[i] = requant(ig[i] + ig[i - 1], fin.F, it)
        ig[0] = requant(v + ig[0], fin.F, it)
        return ig[N - 1]

    def comb(v):
        for k in range(N):
            d = st["dl"][k]
            o = requant(v - d[M - 1], fin.F, it)
            for i in range(1, M):                      # ascending, as the reference writes it (ac_cic_full_core.h:249-254)
                d[i] = d[i - 1]
            d[0] = v
            v = o
        return v

    y = []
    if not interp:
        inf = []
        for s in x:
            valid = st["rc"] == 0
            o = stage(requant(s, fin.F, it))
            st["rc"] = 0 if st["rc"] + 1 > R - 1 else st["rc"] + 1
            if valid:
                inf.append(o)
        for d in inf:
            y.append(requant(comb(d), fin.F, fo))
    else:
        inf = [comb(requant(s, fin.F, it)) for s in x]
        data = 0
        while inf:
            if st["dvalid"]:
                data = inf.pop(0)
            if st["rc1"] == R - 1:
                sin, st["rc1"], st["dvalid"] = data, 0, False
            elif st["rc1"] == R - 2:
                sin, st["rc1"], st["dvalid"] = 0, st["rc1"] + 1, True
            else:
                sin, st["rc1"], st["dvalid"] = 0, st["rc1"] + 1, False
            fin_v = requant(stage(sin), fin.F, fo)
            if st["cnt"] < N - 1:
                st["cnt"] += 1
            else:
                y.append(fin_v)
    return y, st
