"""CPU oracle for the ac_dsp FIR/CIC hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (ac_dsp_amd/, include/) never does.
"""
from .binding import (Fmt, OracleFir, OracleCic, OraclePolyDec, OraclePolyIntr, OracleIntgDump, OracleMvAvg, WIN_MODES, requant, from_double, stimulus, cic_int_type,  # noqa: F401
                      Q_MODES, O_MODES, FTYPES, lib, OracleFirW, OracleCicW, requant_wide)
