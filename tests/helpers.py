"""Shared test helpers: golden-vector loading, stimulus, oracle/engine pairing."""
import math
import os
from fractions import Fraction

import numpy as np

from oracle import Fmt as OFmt, from_double

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TXT = os.path.join(GOLD, "ref_txt")


def read_fracs(name):
    with open(os.path.join(TXT, name)) as f:
        return [Fraction(t) for t in f.read().replace(",", " ").split()]


def to_raw(vals, F):
    out = []
    for v in vals:
        r = v * (1 << F)
        assert r.denominator == 1, "golden value %s is not a multiple of 2^-%d" % (v, F)
        out.append(int(r))
    return np.array(out, dtype=np.int64)


def two_tone(fmt, n=1024):
    """Stimulus of the reference FIR testbenches (tests/rtest_ac_fir_const_coeffs.cpp:126-151)."""
    mx = ((1 << (fmt.W - 1)) - 1) / 2.0 ** (fmt.W - fmt.I)
    v = [math.sin(2 * math.pi * 25 * i / 500) + math.sin(2 * math.pi * 150 * i / 500) for i in range(n)]
    am = max(abs(t) for t in v)
    return np.array([from_double((t / am) * mx, fmt) for t in v], dtype=np.int64)


def sqnr_db(y_raw, F, ref):
    yd = y_raw.astype(np.float64) / 2.0 ** F
    r = np.array([float(t) for t in ref], dtype=np.float64)
    return 10 * math.log10((r ** 2).sum() / ((yd - r) ** 2).sum())


def ofmt(f):
    """engine Fmt -> oracle Fmt (same fields, separate ctypes class)."""
    return OFmt(f.W, f.I, f.S, f.Q, f.O)


def windowed_sinc(n_taps, cutoff, fmt, gain=1.0):
    """Symmetric low-pass (Hamming-windowed sinc) quantised to fmt; returns raw int64 coefficients."""
    m = (n_taps - 1) / 2.0
    k = np.arange(n_taps) - m
    h = np.sinc(2 * cutoff * k) * 2 * cutoff * (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n_taps) / max(n_taps - 1, 1)))
    h = h / h.sum() * gain
    raw = np.round(h * 2.0 ** (fmt.W - fmt.I)).astype(np.int64)
    raw = (raw + raw[::-1]) // 2  # exact symmetry so that the folded architectures are valid
    lo, hi = -(1 << (fmt.W - 1)), (1 << (fmt.W - 1)) - 1
    return np.clip(raw, lo, hi)
