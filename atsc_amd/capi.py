"""ctypes binding of include/atsc_hip.h (libatsc_hip.so).

This module is plumbing: it loads the in-tree shared library and declares the C ABI.  It fails
loudly when the library is missing -- there is no Python or CPU fallback for the compressor.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANT = os.environ.get("ATSC_LIB_VARIANT", "")  # dev builds of atsc_amd.build (instrumented kernels), never the default
LIB_PATH = os.path.join(_HERE, "libatsc_hip%s.so" % ("_" + _VARIANT if _VARIANT else ""))

NOOP, FFT, IDW, CONSTANT, POLYNOMIAL, AUTO, RLE = 0, 1, 2, 3, 4, 5, 6
COMPRESSOR_NAMES = {0: "noop", 1: "fft", 2: "idw", 3: "constant", 4: "polynomial", 5: "auto", 6: "rle"}

OK = 0
E_INVALID, E_NOMEM, E_UNSUPPORTED, E_NO_DEVICE, E_HIP, E_CAPACITY, E_FORMAT, E_VERSION, E_IO = (
    -1, -2, -3, -4, -5, -6, -7, -8, -9)


class AtscError(RuntimeError):
    def __init__(self, rc, msg=""):
        self.rc = rc
        super().__init__("atsc rc=%d (%s) %s" % (rc, lib().atsc_strerror(rc).decode(), msg))


class FrameDiag(C.Structure):
    _fields_ = [
        ("fft_size", C.c_uint32), ("poly_size", C.c_uint32), ("rle_size", C.c_uint32),
        ("fft_trips", C.c_uint16), ("fft_k", C.c_uint16),
        ("poly_trips", C.c_uint16), ("poly_step", C.c_uint16),
        ("poly_points", C.c_uint32),
        ("fft_err", C.c_double), ("poly_err", C.c_double),
    ]


_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); must cover every function declared in include/atsc_hip.h
SIGNATURES = {
    "atsc_strerror": (C.c_char_p, [C.c_int]),
    "atsc_version": (C.c_char_p, []),
    "atsc_ctx_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "atsc_ctx_destroy": (None, [_vp]),
    "atsc_ctx_last_error": (C.c_char_p, [_vp]),
    "atsc_ctx_trim": (C.c_int, [_vp]),
    "atsc_release_caches": (None, []),
    "atsc_host_register": (C.c_int, [_vp, C.c_uint64]),
    "atsc_host_unregister": (C.c_int, [_vp]),
    "atsc_plan_create": (C.c_int, [_vp, _u64p, C.c_uint64, C.POINTER(_vp)]),
    "atsc_plan_destroy": (None, [_vp]),
    "atsc_plan_n_frames": (C.c_uint64, [_vp]),
    "atsc_plan_n_samples": (C.c_uint64, [_vp]),
    "atsc_plan_body_bound": (C.c_uint64, [_vp]),
    "atsc_payload_bound_bytes": (C.c_uint64, [C.c_uint64]),
    "atsc_compress_plan_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int,
                                         _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "atsc_compress_plan_dev_pipelined": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int,
                                                   _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "atsc_plan_join": (C.c_int, [_vp, _vp, _vp]),
    "atsc_plan_input_release": (C.c_int, [_vp, _vp, _vp]),
    "atsc_ctx_set_chains": (C.c_int, [_vp, C.c_int]),
    "atsc_ctx_set_adaptive_order": (C.c_int, [_vp, C.c_int]),
    "atsc_ctx_enable_diag": (C.c_int, [_vp, C.c_int]),
    "atsc_ctx_last_diag": (C.c_int, [_vp, C.POINTER(FrameDiag), C.c_uint64]),
    "atsc_ctx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "atsc_ctx_profile_read": (C.c_int, [_vp, _f64p, _u64p]),
    "atsc_compress_frames": (C.c_int, [_vp, _f64p, _u64p, C.c_uint64, C.c_int, C.c_int, C.c_float,
                                       C.c_int, _u8p, C.c_uint64, _u64p, _u64p, _u8p, _f64p]),
    "atsc_shard_range": (None, [C.c_uint64, C.c_uint32, C.c_uint32, _u64p, _u64p]),
    "atsc_shard_range_weighted": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _u64p]),
    "atsc_compress_frames_sharded": (C.c_int, [C.POINTER(_vp), C.c_uint32, _f64p, _u64p, C.c_uint64, C.c_int, C.c_int,
                                               C.c_float, C.c_int, _u8p, C.c_uint64, _u64p, _u64p, _u8p, _f64p]),
    "atsc_dplan_create": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_int, C.POINTER(_vp)]),
    "atsc_dplan_destroy": (None, [_vp]),
    "atsc_dplan_n_frames": (C.c_uint64, [_vp]),
    "atsc_dplan_n_samples": (C.c_uint64, [_vp]),
    "atsc_decompress_plan_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "atsc_decompress_frames": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_int, _f64p, C.c_uint64, _u64p]),
    "atsc_decompress_frames_alloc": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_int, C.POINTER(_f64p), _u64p]),
    "atsc_stream_new": (C.c_int, [_vp, C.POINTER(_vp)]),
    "atsc_stream_from_bytes": (C.c_int, [_vp, _u8p, C.c_uint64, C.POINTER(_vp)]),
    "atsc_stream_free": (None, [_vp]),
    "atsc_stream_compress_chunk": (C.c_int, [_vp, _f64p, C.c_uint64]),
    "atsc_stream_compress_chunk_with": (C.c_int, [_vp, _f64p, C.c_uint64, C.c_int]),
    "atsc_stream_compress_chunk_bounded_with": (C.c_int, [_vp, _f64p, C.c_uint64, C.c_int, C.c_float, C.c_int]),
    "atsc_stream_frame_count": (C.c_uint64, [_vp]),
    "atsc_stream_to_bytes": (C.c_int, [_vp, C.POINTER(_u8p), _u64p]),
    "atsc_stream_decompress": (C.c_int, [_vp, C.POINTER(_f64p), _u64p]),
    "atsc_free": (None, [_vp]),
    "atsc_compress_data": (C.c_int, [_vp, _f64p, C.c_uint64, C.c_int, C.c_uint8, C.c_int, C.POINTER(_u8p), _u64p]),
    "atsc_decompress_data": (C.c_int, [_vp, _u8p, C.c_uint64, C.POINTER(_f64p), _u64p]),
    "atsc_wbro_from_bytes": (C.c_int, [_u8p, C.c_uint64, C.POINTER(_f64p), _u64p]),
    "atsc_wbro_to_bytes": (C.c_int, [_f64p, C.c_uint64, C.POINTER(_u8p), _u64p]),
    "atsc_wbro_read": (C.c_int, [C.c_char_p, C.POINTER(_f64p), _u64p]),
    "atsc_wbro_write": (C.c_int, [C.c_char_p, _f64p, C.c_uint64]),
    "atsc_bro_read_file": (C.c_int, [C.c_char_p, C.POINTER(_u8p), _u64p]),
    "atsc_csv_read": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(_f64p), _u64p]),
    "atsc_chunk_sizes": (C.c_uint64, [C.c_uint64, _u64p, C.c_uint64]),
    "atsc_clean_data": (C.c_uint64, [_f64p, C.c_uint64, _f64p]),
    "atsc_next_size": (C.c_uint64, [C.c_uint64]),
    "atsc_bro_prefix": (C.c_uint64, [C.c_uint64, _u8p]),
    "atsc_bro_open": (C.c_int, [_u8p, C.c_uint64, _u64p, _u64p]),
    "atsc_bro_scan": (C.c_int, [_u8p, C.c_uint64, _u64p, _u64p]),
    # csv-compressor front end (atsc_vsri.cpp)
    "atsc_vsri_new": (_vp, []),
    "atsc_vsri_free": (None, [_vp]),
    "atsc_vsri_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "atsc_vsri_flush_to": (C.c_int, [_vp, C.c_char_p]),
    "atsc_vsri_update_for_point": (C.c_int, [_vp, C.c_int32]),
    "atsc_vsri_min": (C.c_int32, [_vp]),
    "atsc_vsri_max": (C.c_int32, [_vp]),
    "atsc_vsri_segment_count": (C.c_uint64, [_vp]),
    "atsc_vsri_segment": (C.c_int, [_vp, C.c_uint64, _i32p]),
    "atsc_vsri_get_sample_count": (C.c_int32, [_vp]),
    "atsc_vsri_get_sample": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_get_next_sample": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_get_previous_sample": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_get_this_or_next": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_get_this_or_previous": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_get_time": (C.c_int, [_vp, C.c_int32, _i32p]),
    "atsc_vsri_is_empty": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "atsc_vsri_get_all_timestamps": (C.c_int, [_vp, C.POINTER(_i32p), _u64p]),
    "atsc_day_elapsed_seconds": (C.c_int, [C.c_int64, _i32p]),
    "atsc_samples_csv_read": (C.c_int, [C.c_char_p, C.POINTER(_i64p), C.POINTER(_f64p), _u64p]),
    "atsc_samples_csv_write": (C.c_int, [C.c_char_p, _i64p, _f64p, C.c_uint64]),
    "atsc_metric_index_samples": (C.c_int, [_vp, _i64p, C.c_uint64, _u64p]),
    "atsc_metric_sample_times": (C.c_int, [_vp, C.c_uint64, _i64p]),
}

_lib = None


def lib():
    """Loads libatsc_hip.so (built by __graft_entry__.build / atsc_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libatsc_hip.so is missing (%s): build it with `python -m atsc_amd.build`; "
                "the compressor has no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != OK:
        msg = ""
        if ctx:
            msg = lib().atsc_ctx_last_error(ctx).decode()
        raise AtscError(rc, msg)
