"""ac_dsp_amd -- MI355X-native fixed-point streaming-filter engine (ac_dsp FIR/CIC hot path).

The product is the C-ABI library ``ac_dsp_amd/lib/libacdsp.so`` (HIP kernels for gfx950, declared in
``include/acdsp.h``) and the drop-in C++ class templates in ``include/ac_dsp/``.  This Python package
is plumbing for tests and ``bench.py``: ctypes bindings that hand torch device pointers and streams
to the same C ABI.  Importing it fails loudly when the HIP library has not been built.
"""
from ._lib import lib, AcdspError, Fmt, FirDesc, CicDesc, PolyDecDesc, PolyIntrDesc, IntgDumpDesc, MvAvgDesc, StreamHdr, Q_MODES, O_MODES, FTYPES, KINDS, PATHS, elem_bytes  # noqa: F401
from .engine import Fir, Cic, PolyDec, PolyIntr, IntgDump, MvAvg, Ddc, save_stream, load_stream, fill_stimulus, device_count, torch_dtype_for, is_wide, wide_to_int, diag_copy_ms, diag_mix_ms, empty_paired, shop_output, diag_fir_envelope_ms, diag_fir_envelope_copygeom_ms, diag_shader_clock_mhz, node_shard, NodeFir, NodeCic, NodeDdc, NodePolyDec, NodePolyIntr, NodeIntgDump, NodeMvAvg  # noqa: F401
