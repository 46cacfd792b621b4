"""ctypes binding of libsmx.so (include/smx.h).  There is no Python/CPU implementation of the hot
path: if the HIP library is missing or cannot be loaded, importing the compute entry points fails
loudly with instructions to build it."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMX_LIB") or os.path.join(_HERE, "libsmx.so")   # SMX_LIB: A/B builds (tools/tune.sh)

ABI_VERSION = 4
OK, ERR_ARG, ERR_UNSUPPORTED, ERR_DEVICE, ERR_OVERFLOW = 0, -1, -2, -3, -4
TRIM = {"none": 0, "tails": 1, "barcodes": 2, "primers": 3}
DEREP = {"none": 0, "best": 1}
R_FILTERED, R_FULL, R_PARTIAL_FWD, R_PARTIAL_REV, R_MULTIPLE, R_UNKNOWN, R_DEREP_FULL = range(7)
OPF_REVERSE, OPF_TRIM_EMPTY, OPF_NO_SPECIMEN = 1, 2, 4
CNT_TOTAL, CNT_MATCHED, CNT_FILTERED, CNT_OPS_FULL, CNT_OPS_PARTIAL, CNT_OPS_UNKNOWN, CNT_MULTI_OP_READS, \
    CNT_OVERFLOW, CNT_SPECIMEN0 = range(9)

# numpy views of the ABI records (must match include/smx.h byte for byte)
OP_DTYPE = np.dtype([("sample", "<i4"), ("trim_start", "<i4"), ("trim_end", "<i4"), ("pool", "<i2"),
                     ("p1", "<i2"), ("p2", "<i2"), ("barcode", "<i2"), ("dist", "i1", (4,)), ("rtype", "u1"),
                     ("flags", "u1"), ("n_ops", "<u2"), ("read", "<u4")])
HIT_DTYPE = np.dtype([("first_start", "<i4"), ("first_end", "<i4"), ("tail_end", "<i4"), ("pdist", "<i2"),
                      ("nloc", "<i2"), ("bbest", "<i2"), ("ntied", "<i2"), ("first_tied", "<i2"), ("flags", "<i2")])
assert OP_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 24


class PanelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_primers", C.c_uint32), ("n_barcodes", C.c_uint32),
                ("n_specimens", C.c_uint32), ("n_pools", C.c_uint32), ("n_pairs", C.c_uint32),
                ("primer_rc", C.c_char_p), ("primer_rc_off", C.c_void_p), ("primer_dir", C.c_void_p),
                ("primer_k", C.c_void_p), ("primer_file_index", C.c_void_p), ("primer_bc_off", C.c_void_p),
                ("primer_bc", C.c_void_p), ("barcode_rc", C.c_char_p), ("barcode_rc_off", C.c_void_p),
                ("pair_fwd", C.c_void_p), ("pair_rev", C.c_void_p), ("pair_pool", C.c_void_p),
                ("spec_b1", C.c_void_p), ("spec_b2", C.c_void_p), ("spec_p1mask", C.c_void_p),
                ("spec_p2mask", C.c_void_p), ("spec_pool", C.c_void_p),
                ("k_index", C.c_int32), ("search_len", C.c_int32), ("barcode_len_max", C.c_int32),
                ("prefilter_min_len", C.c_int32), ("preorient", C.c_int32), ("trim", C.c_int32),
                ("dereplicate", C.c_int32), ("min_length", C.c_int32), ("max_length", C.c_int32),
                ("want_starts", C.c_int32)]


# every symbol include/smx.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("smx_abi_version", C.c_int, []),
    ("smx_last_error", C.c_char_p, []),
    ("smx_device_init", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("smx_panel_create", C.c_int, [C.POINTER(PanelDesc), C.POINTER(_P)]),
    ("smx_panel_destroy", None, [_P]),
    ("smx_counts_len", C.c_size_t, [_P]),
    ("smx_window_stride", C.c_size_t, [_P]),
    ("smx_hits_per_read", C.c_size_t, [_P]),
    ("smx_bdist_per_read", C.c_size_t, [_P]),
    ("smx_pack_windows", C.c_int, [_P, _P, C.c_uint32, C.c_int32, _P, _P]),
    ("smx_batch_run_device", C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, _P, C.c_uint32, _P, _P, _P, _P]),
    ("smx_batch_run", C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, C.c_uint32, C.POINTER(C.c_uint32), _P, _P, _P]),
    ("smx_debug_kernel_times", C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    ("smx_panel_set_streams", C.c_int, [_P, C.c_int]),
    ("smx_lane_create", C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    ("smx_lane_destroy", None, [_P]),
    ("smx_lane_windows", _P, [_P]),
    ("smx_lane_lens", _P, [_P]),
    ("smx_lane_submit", C.c_int, [_P, C.c_uint32]),
    ("smx_lane_submit_packed", C.c_int, [_P, C.c_uint32]),
    ("smx_packed_stride", C.c_size_t, [_P]),
    ("smx_pack_windows4", C.c_int, [_P, _P, C.c_uint32, C.c_int32, _P, _P, C.POINTER(C.c_uint32)]),
    ("smx_pack_windows4_batch", C.c_int, [_P, C.c_int32, _P, _P, C.POINTER(C.c_uint32)]),
    ("smx_min_pairwise_distance", C.c_int, [C.c_char_p, _P, C.c_uint32, C.POINTER(C.c_int32)]),
    ("smx_unpack_windows_device", C.c_int, [_P, _P, _P, C.c_uint32, _P]),
    ("smx_lane_wait", C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_uint32), _P]),
    ("smx_align", C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    ("smx_align_batch", C.c_int, [_P, _P, C.c_uint32, _P, _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, C.c_uint32]),
    ("smx_comm_unique_id", C.c_int, [_P]),
    ("smx_comm_init", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("smx_counts_allreduce", C.c_int, [_P, C.c_size_t, _P, _P]),
    ("smx_comm_destroy", None, [_P]),
    # host streaming helpers
    ("smx_reader_open", C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int)]),
    ("smx_reader_open_range", C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_int)]),
    ("smx_reader_close", None, [_P]),
    ("smx_batch_new", _P, []),
    ("smx_batch_free", None, [_P]),
    ("smx_reader_next", C.c_int, [_P, C.c_uint32, C.c_uint64, _P, C.POINTER(C.c_uint32)]),
    ("smx_batch_size", C.c_uint32, [_P]),
    ("smx_batch_record", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                   C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_uint32)]),
    ("smx_pack_windows_batch", C.c_int, [_P, C.c_int32, _P, _P]),
    ("smx_writer_open", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, _P, C.POINTER(_P)]),
    ("smx_writer_write", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32]),
    ("smx_writer_close", C.c_int, [_P]),
]


class Names(C.Structure):
    _fields_ = [("specimens", C.c_char_p), ("specimen_off", C.c_void_p), ("n_specimens", C.c_uint32),
                ("pools", C.c_char_p), ("pool_off", C.c_void_p), ("n_pools", C.c_uint32),
                ("primers", C.c_char_p), ("primer_off", C.c_void_p), ("n_primers", C.c_uint32),
                ("barcodes", C.c_char_p), ("barcode_off", C.c_void_p), ("n_barcodes", C.c_uint32)]


class SmxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libsmx error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load libsmx.so (built in-tree by `make -C specimux_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension was not built. specimux_amd has no CPU fallback; "
                "run `make -C specimux_amd/csrc` (needs hipcc, targets gfx950).")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 / libhsa-runtime64 / librccl (same
        # sonames as /opt/rocm's).  If torch is imported AFTER libsmx.so pulled in /opt/rocm's copies, two HSA
        # runtimes end up in the process and the second one finds no GPU.  Importing torch first makes the loader
        # resolve libsmx.so's NEEDED entries to the copies torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)   # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        if lib.smx_abi_version() != ABI_VERSION:
            raise ImportError(f"libsmx ABI {lib.smx_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise SmxError(rc, load().smx_last_error().decode("utf-8", "replace"))


def ptr(a):
    """numpy array -> void* (array must stay alive for the call)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
