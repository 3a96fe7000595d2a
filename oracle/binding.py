"""ctypes binding of oracle/libacdsp_oracle.so (see acdsp_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libacdsp_oracle.so")
_SAN = os.environ.get("ACDSP_ORACLE_LIB")   # the sanitizer build (oracle/Makefile: sanitize), tests/test_sanitizers.py

Q_MODES = {"TRN": 0, "RND": 1, "TRN_ZERO": 2, "RND_ZERO": 3, "RND_INF": 4, "RND_MIN_INF": 5, "RND_CONV": 6,
           "RND_CONV_ODD": 7}
O_MODES = {"WRAP": 0, "SAT": 1, "SAT_ZERO": 2, "SAT_SYM": 3}
FTYPES = {"SHIFT_REG": 0, "ROTATE_SHIFT": 1, "C_BUFF": 2, "FOLD_EVEN": 3, "FOLD_ODD": 4, "TRANSPOSED": 5,
          "FOLD_EVEN_ANTI": 6, "FOLD_ODD_ANTI": 7}


class Fmt(C.Structure):
    """ac_fixed<W,I,S,Q,O> descriptor (same field order as orc_fmt_t / acdsp_fmt_t)."""
    _fields_ = [("W", C.c_int32), ("I", C.c_int32), ("S", C.c_int32), ("Q", C.c_int32), ("O", C.c_int32)]

    def __init__(self, W, I, S=True, Q="TRN", O="WRAP"):
        super().__init__(W, I, int(bool(S)), Q_MODES[Q] if isinstance(Q, str) else Q,
                         O_MODES[O] if isinstance(O, str) else O)

    @property
    def F(self):
        return self.W - self.I

    def __repr__(self):
        return "Fmt(%d,%d,%d,%d,%d)" % (self.W, self.I, self.S, self.Q, self.O)


def _build():
    srcs = [os.path.join(_HERE, f) for f in ("acdsp_oracle.c", "acdsp_oracle_wide.cpp", "acdsp_oracle.h")]
    if (not os.path.exists(_SO)) or any(os.path.exists(f) and os.path.getmtime(_SO) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_build()
lib = C.CDLL(_SAN if _SAN else _SO)
_i64p = C.POINTER(C.c_int64)
lib.orc_requant.restype = C.c_int64
lib.orc_requant.argtypes = [C.c_int64, C.c_int32, C.POINTER(Fmt)]
lib.orc_requant128.restype = C.c_int64
lib.orc_requant128.argtypes = [C.c_int64, C.c_uint64, C.c_int32, C.POINTER(Fmt)]
lib.orc_from_double.restype = C.c_int64
lib.orc_from_double.argtypes = [C.c_double, C.POINTER(Fmt)]
lib.orc_fir_new.restype = C.c_void_p
lib.orc_fir_new.argtypes = [C.c_int32, C.c_int32] + [C.POINTER(Fmt)] * 4
lib.orc_fir_free.argtypes = [C.c_void_p]
lib.orc_fir_reset.argtypes = [C.c_void_p]
lib.orc_fir_run.restype = C.c_int32
lib.orc_fir_run.argtypes = [C.c_void_p, _i64p, _i64p, C.c_int64, _i64p]
lib.orc_fir_reg_share_run.restype = C.c_int32
lib.orc_fir_reg_share_run.argtypes = [C.c_void_p, _i64p, C.c_int32, C.c_int32, C.c_int32, _i64p, C.c_int64, _i64p]
lib.orc_fir_reg_share_delay_line.restype = C.c_int64
lib.orc_fir_reg_share_delay_line.argtypes = [C.c_void_p]
lib.orc_cic_new.restype = C.c_void_p
lib.orc_cic_new.argtypes = [C.c_int32] * 4 + [C.POINTER(Fmt)] * 2
lib.orc_cic_free.argtypes = [C.c_void_p]
lib.orc_cic_int_type.restype = C.c_int32
lib.orc_cic_int_type.argtypes = [C.c_int32] * 4 + [C.POINTER(Fmt)] * 2
lib.orc_cic_run.restype = C.c_int64
lib.orc_cic_run.argtypes = [C.c_void_p, _i64p, C.c_int64, _i64p, C.c_int64]
lib.orc_polydec_new.restype = C.c_void_p
lib.orc_polydec_new.argtypes = [C.c_int32, C.c_int32] + [C.POINTER(Fmt)] * 4
lib.orc_polydec_free.argtypes = [C.c_void_p]
lib.orc_polydec_run.restype = C.c_int64
lib.orc_polydec_run.argtypes = [C.c_void_p, _i64p, _i64p, C.c_int64, _i64p]
_u8p = C.POINTER(C.c_uint8)
lib.orc_polyintr_new.restype = C.c_void_p
lib.orc_polyintr_new.argtypes = [C.c_int32] * 4 + [C.POINTER(Fmt)] * 4
lib.orc_polyintr_free.argtypes = [C.c_void_p]
lib.orc_polyintr_run.restype = C.c_int64
lib.orc_polyintr_run.argtypes = [C.c_void_p, _i64p, _u8p, _u8p, _i64p, C.c_int64, _i64p]
lib.orc_intg_dump_run.restype = C.c_int64
lib.orc_intg_dump_run.argtypes = [_i64p, C.c_int32, C.c_int32] + [C.POINTER(Fmt)] * 3 + [_i64p, C.c_int64, _i64p, _i64p, _i64p]
lib.orc_mv_avg_run.restype = C.c_int64
lib.orc_mv_avg_run.argtypes = [C.c_int32, C.c_int32] + [C.POINTER(Fmt)] * 4 + [_i64p, _i64p, C.c_int64, C.c_int64, _i64p]
lib.orc_stimulus.restype = C.c_int64
lib.orc_stimulus.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32]
lib.orc_splitmix64.restype = C.c_uint64
lib.orc_splitmix64.argtypes = [C.c_uint64, C.c_uint64]


class WWord(C.Structure):
    """orcw_word_t: a 128-bit raw word, low quadword first."""
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_int64)]


lib.orcw_fir_new.restype = C.c_void_p
lib.orcw_fir_new.argtypes = [C.c_int32, C.c_int32] + [C.POINTER(Fmt)] * 4
lib.orcw_fir_free.argtypes = [C.c_void_p]
lib.orcw_fir_run.restype = C.c_int32
lib.orcw_fir_run.argtypes = [C.c_void_p, _i64p, _i64p, C.c_int64, C.POINTER(WWord)]
lib.orcw_cic_new.restype = C.c_void_p
lib.orcw_cic_new.argtypes = [C.c_int32] * 4 + [C.POINTER(Fmt)] * 2
lib.orcw_cic_free.argtypes = [C.c_void_p]
lib.orcw_cic_run.restype = C.c_int64
lib.orcw_cic_run.argtypes = [C.c_void_p, _i64p, C.c_int64, C.POINTER(WWord), C.c_int64]
lib.orcw_requant.restype = C.c_int32
lib.orcw_requant.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.POINTER(Fmt), C.POINTER(WWord)]


def _p(a):
    return a.ctypes.data_as(_i64p)


def _words_to_int(buf, k):
    """first k orcw_word_t of a ctypes array -> object array of Python ints"""
    a = np.frombuffer(buf, dtype=np.dtype([("lo", np.uint64), ("hi", np.int64)]), count=k)
    return a["hi"].astype(object) * (1 << 64) + a["lo"].astype(object)


def requant_wide(x, f_src, fmt):
    """Exact python-int x * 2^-f_src (|x| < 2^255) -> raw word of fmt (W <= 128), through the wide oracle."""
    x = int(x) & ((1 << 256) - 1)
    limbs = (C.c_uint64 * 4)(*[(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)])
    w = WWord()
    if lib.orcw_requant(limbs, f_src, C.byref(fmt), C.byref(w)):
        raise ValueError("oracle: unsupported format")
    return w.hi * (1 << 64) + w.lo


class OracleFirW:
    """OracleFir for ACC_TYPE / OUT_TYPE of up to 128 bits (acdsp_oracle_wide.cpp); run() returns an object array of Python ints."""

    def __init__(self, n_taps, ftype, fin, fcoeff, facc, fout, n_ch=1):
        self.n_taps, self.n_ch = n_taps, n_ch
        ft = FTYPES[ftype] if isinstance(ftype, str) else ftype
        self._h = [lib.orcw_fir_new(n_taps, ft, C.byref(fin), C.byref(fcoeff), C.byref(facc), C.byref(fout)) for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported wide FIR configuration")

    def run(self, coeffs, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        n = x.shape[1]
        out = np.empty((self.n_ch, n), dtype=object)
        buf = (WWord * max(n, 1))()
        for ch in range(self.n_ch):
            c = np.ascontiguousarray(coeffs[ch] if coeffs.ndim == 2 else coeffs)
            if lib.orcw_fir_run(self._h[ch], _p(c), _p(x[ch]), n, buf):
                raise ValueError("oracle: ftype not handled by the reference run()")
            out[ch] = _words_to_int(buf, n)
        return out

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orcw_fir_free(h)


class OracleCicW:
    """OracleCic for INT_TYPE / OUT_TYPE of up to 128 bits; run() returns an object array of Python ints."""

    def __init__(self, interp, R, M, N, fin, fout, n_ch=1):
        self.interp, self.R, self.n_ch = int(interp), R, n_ch
        self._h = [lib.orcw_cic_new(int(interp), R, M, N, C.byref(fin), C.byref(fout)) for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported wide CIC configuration")

    def run(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        n = x.shape[1]
        cap = (n + 2) * (self.R if self.interp else 1) + 8
        buf = (WWord * cap)()
        outs = []
        for ch in range(self.n_ch):
            k = lib.orcw_cic_run(self._h[ch], _p(x[ch]), n, buf, cap)
            assert k >= 0
            outs.append(_words_to_int(buf, k))
        return np.stack(outs) if outs else np.empty((0, 0), dtype=object)

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orcw_cic_free(h)


def requant(x, f_src, fmt):
    """Exact python-int x * 2^-f_src -> raw word of fmt."""
    x = int(x)
    lo = x & ((1 << 64) - 1)
    hi = (x >> 64)
    if hi >= (1 << 63):
        hi -= (1 << 64)
    return lib.orc_requant128(hi, lo, f_src, C.byref(fmt))


def from_double(d, fmt):
    return lib.orc_from_double(float(d), C.byref(fmt))


def stimulus(seed, n_ch, n, bits, ch0=0, t0=0):
    """[n_ch][n] int64 array of the shared counter-hash stimulus (vectorised numpy restatement)."""
    ch = (np.arange(ch0, ch0 + n_ch, dtype=np.uint64)[:, None] << np.uint64(32))
    t = (np.arange(t0, t0 + n, dtype=np.uint64)[None, :] & np.uint64(0xffffffff))
    idx = ch | t
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.view(np.int64) >> np.int64(64 - bits)


def cic_int_type(interp, R, M, N, fin):
    it = Fmt(1, 1)
    if lib.orc_cic_int_type(int(interp), R, M, N, C.byref(fin), C.byref(it)):
        raise ValueError("CIC intermediate type not representable (reference would not compile)")
    return it


class OracleFir:
    """One reference-style FIR object per channel; run() takes/returns [n_ch][n] int64 raw words."""

    def __init__(self, n_taps, ftype, fin, fcoeff, facc, fout, n_ch=1, reg_share=None):
        """reg_share = (MEM_WORD_WIDTH, BLK_SZ, BLK_OFFSET): follow ac_fir_reg_share::run instead of the
        const/load/prog cores (run() then takes the coeffs[N_TAPS] array as that class receives it)."""
        self.n_taps, self.n_ch, self.reg_share = n_taps, n_ch, reg_share
        ft = FTYPES[ftype] if isinstance(ftype, str) else ftype
        self._h = [lib.orc_fir_new(n_taps, ft, C.byref(fin), C.byref(fcoeff), C.byref(facc), C.byref(fout))
                   for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported FIR configuration")

    def run(self, coeffs, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        y = np.empty_like(x)
        for ch in range(self.n_ch):
            c = coeffs[ch] if coeffs.ndim == 2 else coeffs
            c = np.ascontiguousarray(c)
            if self.reg_share:
                mww, bs, bo = self.reg_share
                rc = lib.orc_fir_reg_share_run(self._h[ch], _p(c), mww, bs, bo, _p(x[ch]), x.shape[1], _p(y[ch]))
            else:
                rc = lib.orc_fir_run(self._h[ch], _p(c), _p(x[ch]), x.shape[1], _p(y[ch]))
            if rc:
                raise ValueError("oracle: ftype / block parameters not handled by the reference run()")
        return y

    def delay_line(self):
        """ac_fir_reg_share::ac_firProgCoeffs_delay_line(): reg[N_TAPS-1] as OUT_TYPE, per channel"""
        return np.array([lib.orc_fir_reg_share_delay_line(h) for h in self._h], dtype=np.int64)

    def reset(self):
        for h in self._h:
            lib.orc_fir_reset(h)

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orc_fir_free(h)


class OracleCic:
    def __init__(self, interp, R, M, N, fin, fout, n_ch=1):
        self.interp, self.R, self.M, self.N, self.n_ch = int(interp), R, M, N, n_ch
        self._h = [lib.orc_cic_new(int(interp), R, M, N, C.byref(fin), C.byref(fout)) for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported CIC configuration")

    def run(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        n = x.shape[1]
        cap = (n + 2) * (self.R if self.interp else 1) + 8
        outs = []
        for ch in range(self.n_ch):
            y = np.empty(cap, dtype=np.int64)
            k = lib.orc_cic_run(self._h[ch], _p(x[ch]), n, _p(y), cap)
            assert k >= 0
            outs.append(y[:k].copy())
        return np.stack(outs)

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orc_cic_free(h)


class OraclePolyDec:
    """n_ch reference-style ac_poly_dec objects; run() consumes floor(n/DF)*DF samples of [n_ch][n]."""

    def __init__(self, ntaps, df, fin, fcoeff, facc, fout, n_ch=1):
        self.ntaps, self.df, self.n_ch = ntaps, df, n_ch
        self._h = [lib.orc_polydec_new(ntaps, df, C.byref(fin), C.byref(fcoeff), C.byref(facc), C.byref(fout)) for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported poly_dec configuration")

    def run(self, coeffs, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        assert coeffs.shape == (self.ntaps * self.df,)
        outs = []
        for ch in range(self.n_ch):
            y = np.empty(x.shape[1] // self.df + 1, dtype=np.int64)
            k = lib.orc_polydec_run(self._h[ch], _p(coeffs), _p(x[ch]), x.shape[1], _p(y))
            outs.append(y[:k].copy())
        return np.stack(outs)

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orc_polydec_free(h)


POLY_FTYPES = {"FOLD_EVEN": 0, "FOLD_ODD": 1, "FOLD_ANTI": 2}


class OraclePolyIntr:
    """One ac_poly_intr object per channel (reference ac_poly_intr.h:104-320); run() -> [n_ch][outputs] int64 raw words."""

    def __init__(self, n_taps, coeff_sz, ifac, ftype, fin, fcoeff, facc, fout, n_ch=1):
        self.ifac, self.n_ch = ifac, n_ch
        ft = POLY_FTYPES[ftype] if isinstance(ftype, str) else ftype
        self._h = [lib.orc_polyintr_new(n_taps, coeff_sz, ifac, ft, C.byref(fin), C.byref(fcoeff), C.byref(facc), C.byref(fout))
                   for _ in range(n_ch)]
        if any(h is None for h in self._h):
            raise ValueError("oracle: unsupported poly_intr configuration")

    def run(self, coeffs, sign, corr, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        c = np.ascontiguousarray(coeffs, dtype=np.int64)
        sg = np.ascontiguousarray(sign, dtype=np.uint8)
        cr = np.ascontiguousarray(corr, dtype=np.uint8)
        outs = []
        for ch in range(self.n_ch):
            y = np.empty(x.shape[1] * self.ifac + 1, dtype=np.int64)
            k = lib.orc_polyintr_run(self._h[ch], _p(c), sg.ctypes.data_as(_u8p), cr.ctypes.data_as(_u8p), _p(x[ch]), x.shape[1], _p(y))
            if k < 0:
                raise ValueError("oracle: the reference would index outside coeffs[] / the accumulator banks")
            outs.append(y[:k].copy())
        return np.stack(outs)

    def __del__(self):
        for h in getattr(self, "_h", []):
            if h:
                lib.orc_polyintr_free(h)


class OracleIntgDump:
    """ac_intg_dump objects (reference ac_intg_dump.h:77-151), one per row; state temp[CHN] carries across run() calls."""

    def __init__(self, ns, chn, fin, facc, fout, n_obj=1):
        self.ns, self.chn, self.fin, self.facc, self.fout, self.n_obj = ns, chn, fin, facc, fout, n_obj
        self.temp = np.zeros((n_obj, chn), dtype=np.int64)

    def run(self, x, n_sample):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        ns = np.ascontiguousarray(n_sample, dtype=np.int64)
        outs = []
        for o in range(self.n_obj):
            y = np.empty(len(ns) * self.chn + 1, dtype=np.int64)
            used = np.zeros(1, dtype=np.int64)
            t = np.ascontiguousarray(self.temp[o])
            k = lib.orc_intg_dump_run(_p(t), self.ns, self.chn, C.byref(self.fin), C.byref(self.facc), C.byref(self.fout), _p(ns), len(ns),
                                      _p(x[o]), _p(y), _p(used))
            assert used[0] <= x.shape[1], "stream shorter than the blocks need"
            self.temp[o] = t
            outs.append(y[:k].copy())
        return np.stack(outs)


WIN_MODES = {"WIN": 0, "MIRROR": 1, "CLIP": 2}


class OracleMvAvg:
    """ac_mv_avg objects (reference ac_mv_avg.h:93-196), one per row; run() = one run() call of n_frames frames of
    n_sample inputs.  Nothing carries across calls (the reference builds a fresh core object per call, :154)."""

    def __init__(self, taps, win_mode, fin, fcoeff, facc, fout, n_obj=1):
        self.taps, self.n_obj = taps, n_obj
        self.mode = WIN_MODES[win_mode] if isinstance(win_mode, str) else win_mode
        self.fmts = (fin, fcoeff, facc, fout)

    def out_per_frame(self, n_sample):
        return max(0, n_sample - self.taps + 1) if self.mode == 0 else n_sample

    def run(self, coeffs, x, n_sample):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.int64)
        c = np.ascontiguousarray(coeffs, dtype=np.int64)
        assert c.shape == (self.taps,) and x.shape[1] % n_sample == 0
        n_frames = x.shape[1] // n_sample
        outs = []
        for o in range(self.n_obj):
            y = np.empty(n_frames * n_sample + 1, dtype=np.int64)
            k = lib.orc_mv_avg_run(self.taps, self.mode, *[C.byref(f) for f in self.fmts], _p(c), _p(x[o]), n_sample, n_frames, _p(y))
            if k < 0:
                raise ValueError("oracle: parameters the reference cannot run (even TAPS / empty frame)")
            outs.append(y[:k].copy())
        return np.stack(outs)
