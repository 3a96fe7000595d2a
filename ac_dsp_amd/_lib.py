"""ctypes binding of libacdsp.so (include/acdsp.h).  No fallbacks: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACDSP_LIB") or os.path.join(_HERE, "lib", "libacdsp.so")  # ACDSP_LIB: A/B testing of builds

Q_MODES = {"TRN": 0, "RND": 1, "TRN_ZERO": 2, "RND_ZERO": 3, "RND_INF": 4, "RND_MIN_INF": 5, "RND_CONV": 6,
           "RND_CONV_ODD": 7}
O_MODES = {"WRAP": 0, "SAT": 1, "SAT_ZERO": 2, "SAT_SYM": 3}
FTYPES = {"SHIFT_REG": 0, "ROTATE_SHIFT": 1, "C_BUFF": 2, "FOLD_EVEN": 3, "FOLD_ODD": 4, "TRANSPOSED": 5,
          "FOLD_EVEN_ANTI": 6, "FOLD_ODD_ANTI": 7}
KINDS = {"const": 0, "load": 1, "prog": 2, "reg_share": 3}
PATHS = {0: "generic", 1: "lossless64", 2: "mfma_i8", 3: "mfma_gen", 4: "wide", 5: "mfma_lossy"}   # (6 = ACDSP_PATH_CIC_2STAGE: Cic.path only)
KCLASSES = {**PATHS, 6: "lossy16", 7: "satacc16"}
FLAG_FORCE_GENERIC = 1


class AcdspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("acdsp error %d: %s" % (code, msg))
        self.code = code


class Fmt(C.Structure):
    """ac_fixed<W,I,S,Q,O> descriptor == acdsp_fmt_t."""
    _fields_ = [("W", C.c_int32), ("I", C.c_int32), ("S", C.c_int32), ("Q", C.c_int32), ("O", C.c_int32)]

    def __init__(self, W=1, I=1, S=True, Q="TRN", O="WRAP"):
        super().__init__(W, I, int(bool(S)), Q_MODES[Q] if isinstance(Q, str) else Q,
                         O_MODES[O] if isinstance(O, str) else O)

    @property
    def F(self):
        return self.W - self.I

    def __repr__(self):
        return "ac_fixed<%d,%d,%s,%d,%d>" % (self.W, self.I, bool(self.S), self.Q, self.O)


class FirDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ftype", C.c_int32), ("n_taps", C.c_int32), ("n_channels", C.c_int32),
                ("coeffs_per_channel", C.c_int32), ("fin", Fmt), ("fcoeff", Fmt), ("facc", Fmt), ("fout", Fmt),
                ("device", C.c_int32), ("flags", C.c_int32)]


class PolyDecDesc(C.Structure):
    _fields_ = [("n_taps", C.c_int32), ("df", C.c_int32), ("n_channels", C.c_int32), ("fin", Fmt), ("fcoeff", Fmt),
                ("facc", Fmt), ("fout", Fmt), ("device", C.c_int32), ("flags", C.c_int32)]


class PolyIntrDesc(C.Structure):
    _fields_ = [("n_taps", C.c_int32), ("coeff_sz", C.c_int32), ("ifac", C.c_int32), ("ftype", C.c_int32), ("n_channels", C.c_int32),
                ("fin", Fmt), ("fcoeff", Fmt), ("facc", Fmt), ("fout", Fmt), ("device", C.c_int32), ("flags", C.c_int32)]


class IntgDumpDesc(C.Structure):
    _fields_ = [("ns", C.c_int32), ("chn", C.c_int32), ("n_objects", C.c_int32), ("fin", Fmt), ("facc", Fmt), ("fout", Fmt),
                ("device", C.c_int32), ("flags", C.c_int32)]


class StreamHdr(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_uint32), ("elem_bytes", C.c_uint32), ("fmt", Fmt), ("reserved", C.c_uint32),
                ("n_channels", C.c_uint64), ("n_samples", C.c_uint64), ("stride", C.c_uint64)]


class MvAvgDesc(C.Structure):
    _fields_ = [("max_sample", C.c_int32), ("taps", C.c_int32), ("win_mode", C.c_int32), ("n_objects", C.c_int32),
                ("fin", Fmt), ("fcoeff", Fmt), ("facc", Fmt), ("fout", Fmt), ("device", C.c_int32), ("flags", C.c_int32)]


class CicDesc(C.Structure):
    _fields_ = [("interp", C.c_int32), ("R", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("n_channels", C.c_int32), ("fin", Fmt), ("fout", Fmt), ("device", C.c_int32), ("flags", C.c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "ac_dsp_amd: %s is missing -- build it with `make` (or __graft_entry__.build()); there is no CPU "
        "fallback for the HIP engine" % LIB_PATH)



def _torch_runtime_first():
    """This package hands torch tensors to the library, and torch carries its own copy of the HIP runtime: in one process torch's copy has to open
    the device BEFORE the library's does ("No HIP GPUs are available" from torch otherwise, seen when a process created an engine handle before its
    first CUDA tensor).  Importing the package therefore initialises torch's runtime when a GPU is there; a C / C++ caller of libacdsp.so has no
    torch and nothing to order.  ACDSP_NO_TORCH_INIT=1 skips it."""
    if os.environ.get("ACDSP_NO_TORCH_INIT"):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.empty(1, device="cuda")
    except Exception:   # noqa: BLE001 -- no torch / no GPU: the library reports its own errors at the first call
        pass


_torch_runtime_first()
lib = C.CDLL(LIB_PATH)
_vp, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32

# every symbol include/acdsp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "acdsp_abi_version": (_i32, []),
    "acdsp_last_error": (C.c_char_p, []),
    "acdsp_device_count": (_i32, []),
    "acdsp_elem_bytes": (_i32, [_i32]),
    "acdsp_dev_alloc": (_i32, [_i32, C.c_uint64, C.POINTER(_vp)]),
    "acdsp_dev_free": (_i32, [_i32, _vp]),
    "acdsp_copy_h2d": (_i32, [_i32, _vp, _vp, C.c_uint64]),
    "acdsp_copy_d2h": (_i32, [_i32, _vp, _vp, C.c_uint64]),
    "acdsp_sync": (_i32, [_i32, _vp]),
    "acdsp_fill_stimulus": (_i32, [_i32, _vp, _i32, _i64, _i64, _i64, C.c_uint64, _i32, C.c_uint64, C.c_uint64, _vp]),
    "acdsp_node_shard": (_i32, [_i64, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "acdsp_node_n_shards": (_i32, [_vp]),
    "acdsp_node_shard_info": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp)]),
    "acdsp_node_last_ms": (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "acdsp_node_destroy": (_i32, [_vp]),
    "acdsp_node_fir_create": (_i32, [C.POINTER(FirDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_fir_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_node_fir_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, C.POINTER(_vp), _i64]),
    "acdsp_node_fir_run_host": (_i32, [_vp, _vp, _i64, _vp]),
    "acdsp_node_cic_create": (_i32, [C.POINTER(CicDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_cic_out_count": (_i64, [_vp, _i64]),
    "acdsp_node_cic_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, C.POINTER(_vp), _i64, C.POINTER(_i64)]),
    "acdsp_node_ddc_create": (_i32, [C.POINTER(CicDesc), C.POINTER(FirDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_ddc_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_node_ddc_out_count": (_i64, [_vp, _i64]),
    "acdsp_node_ddc_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, C.POINTER(_vp), _i64, C.POINTER(_i64)]),
    "acdsp_node_polydec_create": (_i32, [C.POINTER(PolyDecDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_polydec_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_node_polydec_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, C.POINTER(_vp), _i64]),
    "acdsp_node_polyintr_create": (_i32, [C.POINTER(PolyIntrDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_polyintr_set_ctrl": (_i32, [_vp, C.POINTER(_i64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "acdsp_node_polyintr_out_count": (_i64, [_vp, _i64]),
    "acdsp_node_polyintr_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, C.POINTER(_vp), _i64, C.POINTER(_i64)]),
    "acdsp_node_intgdump_create": (_i32, [C.POINTER(IntgDumpDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_intgdump_run": (_i32, [_vp, C.POINTER(_vp), _i64, C.POINTER(_i64), _i64, C.POINTER(_vp), _i64, C.POINTER(_i64)]),
    "acdsp_node_mvavg_create": (_i32, [C.POINTER(MvAvgDesc), _i32, C.POINTER(_i32), C.POINTER(_vp)]),
    "acdsp_node_mvavg_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_node_mvavg_run": (_i32, [_vp, C.POINTER(_vp), _i64, _i64, _i64, C.POINTER(_vp), _i64, C.POINTER(_i64)]),
    "acdsp_diag_mix_ms": (_i32, [_i32, _vp, C.c_uint64, _vp, C.c_uint64, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    "acdsp_dev_alloc_shop": (_i32, [_i32, C.c_uint64, _i32, _vp, _vp, _i32, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "acdsp_dev_alloc_paired": (_i32, [_i32, C.c_uint64, _vp, C.c_uint64, _i32, _i32, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "acdsp_diag_copy_ms": (_i32, [_i32, _vp, _vp, C.c_uint64, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    "acdsp_diag_shader_clock_mhz": (_i32, [_i32, _vp, C.POINTER(C.c_float)]),
    "acdsp_diag_fir_envelope_ms": (_i32, [_i32, C.POINTER(_i64), _i32, _i32, _i32, _vp, _vp, C.c_uint64, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    "acdsp_diag_fir_envelope_copygeom_ms": (_i32, [_i32, C.POINTER(_i64), _i32, _i32, _i32, _vp, _vp, C.c_uint64, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    "acdsp_fir_create": (_i32, [C.POINTER(FirDesc), C.POINTER(_vp)]),
    "acdsp_fir_destroy": (_i32, [_vp]),
    "acdsp_fir_clone": (_i32, [_vp, C.POINTER(_vp)]),
    "acdsp_fir_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_fir_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "acdsp_fir_run_host": (_i32, [_vp, _vp, _i64, _vp]),
    "acdsp_fir_reset": (_i32, [_vp]),
    "acdsp_fir_path": (_i32, [_vp]),
    "acdsp_fir_kernel_class": (_i32, [_vp]),
    "acdsp_fir_mfma_epilogue": (_i32, [_vp]),
    "acdsp_fir_last_kernel_ms": (_i32, [_vp, C.POINTER(C.c_float)]),
    "acdsp_fir_kernel_stats": (_i32, [_vp, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "acdsp_fir_mfma_issued": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "acdsp_cic_create": (_i32, [C.POINTER(CicDesc), C.POINTER(_vp)]),
    "acdsp_cic_destroy": (_i32, [_vp]),
    "acdsp_cic_clone": (_i32, [_vp, C.POINTER(_vp)]),
    "acdsp_cic_int_type": (_i32, [C.POINTER(CicDesc), C.POINTER(Fmt)]),
    "acdsp_cic_out_count": (_i64, [_vp, _i64]),
    "acdsp_cic_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(_i64), _vp]),
    "acdsp_cic_run_host": (_i32, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "acdsp_cic_reset": (_i32, [_vp]),
    "acdsp_cic_path": (_i32, [_vp]),
    "acdsp_cic_last_kernel_ms": (_i32, [_vp, C.POINTER(C.c_float)]),
    "acdsp_cic_kernel_stats": (_i32, [_vp, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "acdsp_polydec_create": (_i32, [C.POINTER(PolyDecDesc), C.POINTER(_vp)]),
    "acdsp_polydec_destroy": (_i32, [_vp]),
    "acdsp_polydec_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_polydec_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "acdsp_polydec_run_host": (_i32, [_vp, _vp, _i64, _vp]),
    "acdsp_polydec_reset": (_i32, [_vp]),
    "acdsp_polydec_path": (_i32, [_vp]),
    "acdsp_polyintr_create": (_i32, [C.POINTER(PolyIntrDesc), C.POINTER(_vp)]),
    "acdsp_polyintr_destroy": (_i32, [_vp]),
    "acdsp_polyintr_set_ctrl": (_i32, [_vp, C.POINTER(_i64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    "acdsp_polyintr_out_count": (_i64, [_vp, _i64]),
    "acdsp_polyintr_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(_i64), _vp]),
    "acdsp_polyintr_run_host": (_i32, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "acdsp_polyintr_reset": (_i32, [_vp]),
    "acdsp_polyintr_path": (_i32, [_vp]),
    "acdsp_intgdump_create": (_i32, [C.POINTER(IntgDumpDesc), C.POINTER(_vp)]),
    "acdsp_intgdump_destroy": (_i32, [_vp]),
    "acdsp_intgdump_counts": (_i32, [_vp, C.POINTER(_i64), _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "acdsp_intgdump_run": (_i32, [_vp, _vp, _i64, C.POINTER(_i64), _i64, _vp, _i64, C.POINTER(_i64), _vp]),
    "acdsp_intgdump_run_host": (_i32, [_vp, _vp, C.POINTER(_i64), _i64, _vp, _i64, C.POINTER(_i64)]),
    "acdsp_intgdump_reset": (_i32, [_vp]),
    "acdsp_intgdump_path": (_i32, [_vp]),
    "acdsp_mvavg_create": (_i32, [C.POINTER(MvAvgDesc), C.POINTER(_vp)]),
    "acdsp_mvavg_destroy": (_i32, [_vp]),
    "acdsp_mvavg_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_mvavg_out_per_frame": (_i64, [_vp, _i64]),
    "acdsp_mvavg_path": (_i32, [_vp]),
    "acdsp_mvavg_run": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, C.POINTER(_i64), _vp]),
    "acdsp_mvavg_run_host": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(_i64)]),
    "acdsp_fir_state_size": (_i64, [_vp]),
    "acdsp_fir_state_get": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_fir_state_set": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_cic_state_size": (_i64, [_vp]),
    "acdsp_cic_state_get": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_cic_state_set": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_ddc_state_size": (_i64, [_vp]),
    "acdsp_ddc_state_get": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_ddc_state_set": (_i32, [_vp, _vp, C.c_uint64]),
    "acdsp_stream_write": (_i32, [C.c_char_p, C.POINTER(StreamHdr), _vp]),
    "acdsp_stream_read_header": (_i32, [C.c_char_p, C.POINTER(StreamHdr)]),
    "acdsp_stream_read": (_i32, [C.c_char_p, _vp, C.c_uint64]),
    "acdsp_ddc_create": (_i32, [C.POINTER(CicDesc), C.POINTER(FirDesc), C.POINTER(_vp)]),
    "acdsp_ddc_destroy": (_i32, [_vp]),
    "acdsp_ddc_set_coeffs": (_i32, [_vp, C.POINTER(_i64)]),
    "acdsp_ddc_out_count": (_i64, [_vp, _i64]),
    "acdsp_ddc_run": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(_i64), _vp]),
    "acdsp_ddc_reset": (_i32, [_vp]),
    "acdsp_ddc_path": (_i32, [_vp]),
    "acdsp_ddc_kernel_stats": (_i32, [_vp, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}
for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(lib, _name)  # AttributeError here == the library does not export what the header declares
    _f.restype = _res
    _f.argtypes = _args


def check(rc):
    if rc != 0:
        raise AcdspError(rc, lib.acdsp_last_error().decode())


def elem_bytes(W):
    return lib.acdsp_elem_bytes(W)
