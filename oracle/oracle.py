"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (atsc_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libatsc_oracle.so")

NOOP, FFT, IDW, CONSTANT, POLYNOMIAL, AUTO, RLE = 0, 1, 2, 3, 4, 5, 6
BD_F64, BD_I32, BD_I16, BD_U8 = 0, 1, 2, 3


class _Buf(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class Stats(C.Structure):
    _fields_ = [
        ("max", C.c_double),
        ("max_loc", C.c_uint64),
        ("min", C.c_double),
        ("min_loc", C.c_uint64),
        ("mean", C.c_double),
        ("bitdepth", C.c_int32),
        ("fractional", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "atsc_oracle.c")
    if (
        force
        or not os.path.exists(_LIB)
        or os.path.getmtime(_LIB) < os.path.getmtime(src)
        or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "atsc_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_next_size.restype = C.c_size_t
        L.orc_next_size.argtypes = [C.c_size_t]
        L.orc_is_decomposable.argtypes = [C.c_size_t]
        L.orc_prev_power_of_two.restype = C.c_size_t
        L.orc_prev_power_of_two.argtypes = [C.c_size_t]
        L.orc_round_f64.restype = C.c_double
        L.orc_round_f64.argtypes = [C.c_double, C.c_uint32]
        L.orc_round_and_limit_f64.restype = C.c_double
        L.orc_round_and_limit_f64.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint32]
        L.orc_error_mape.restype = C.c_double
        L.orc_error_mape.argtypes = [dp, dp, C.c_size_t]
        L.orc_stats_new.argtypes = [dp, C.c_size_t, C.POINTER(Stats)]
        L.orc_chunk_sizes.restype = C.c_size_t
        L.orc_chunk_sizes.argtypes = [C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t]
        L.orc_clean_data.restype = C.c_size_t
        L.orc_clean_data.argtypes = [dp, C.c_size_t, dp]
        L.orc_gibbs_sizing.restype = C.c_size_t
        L.orc_gibbs_sizing.argtypes = [dp, C.c_size_t, dp]
        bp = C.POINTER(_Buf)
        for name in ("orc_noop", "orc_constant", "orc_rle", "orc_fft"):
            getattr(L, name).argtypes = [dp, C.c_size_t, bp]
        L.orc_fft_set.argtypes = [dp, C.c_size_t, C.c_size_t, bp]
        L.orc_polynomial.argtypes = [dp, C.c_size_t, C.c_int, bp]
        L.orc_fft_allowed_error.argtypes = [dp, C.c_size_t, C.c_double, bp, dp, C.POINTER(C.c_int)]
        L.orc_polynomial_allowed_error.argtypes = [
            dp, C.c_size_t, C.c_double, C.c_int, bp, dp, C.POINTER(C.c_int)]
        L.orc_compress.argtypes = [C.c_int, dp, C.c_size_t, C.c_int, C.c_double, bp, dp]
        L.orc_decompress.argtypes = [
            C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_size_t, C.POINTER(dp),
            C.POINTER(C.c_size_t)]
        L.orc_compress_best.argtypes = [
            dp, C.c_size_t, C.c_float, C.c_int, bp, C.POINTER(C.c_int), dp]
        L.orc_stream_compress.argtypes = [
            dp, C.POINTER(C.c_uint64), C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int, bp,
            C.POINTER(C.c_uint8), dp]
        L.orc_compress_data.argtypes = [dp, C.c_size_t, C.c_int, C.c_uint8, C.c_int, bp]
        L.orc_decompress_data.argtypes = [
            C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(dp), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def _arr(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _take(buf):
    out = bytes(C.string_at(buf.ptr, buf.len)) if buf.len else b""
    lib().orc_free(buf.ptr)
    return out


def _bytes_ptr(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def next_size(n):
    return lib().orc_next_size(n)


def prev_power_of_two(n):
    return lib().orc_prev_power_of_two(n)


def is_decomposable(n):
    return bool(lib().orc_is_decomposable(n))


def round_f64(x, d):
    return lib().orc_round_f64(x, d)


def round_and_limit_f64(x, mn, mx, d):
    return lib().orc_round_and_limit_f64(x, mn, mx, d)


def error_mape(orig, gen):
    a, pa = _arr(orig)
    b, pb = _arr(gen)
    assert len(a) == len(b)
    return lib().orc_error_mape(pa, pb, len(a))


def stats(x):
    a, pa = _arr(x)
    s = Stats()
    lib().orc_stats_new(pa, len(a), C.byref(s))
    return s


def chunk_sizes(n):
    cnt = lib().orc_chunk_sizes(n, None, 0)
    out = (C.c_size_t * max(cnt, 1))()
    lib().orc_chunk_sizes(n, out, cnt)
    return [int(out[i]) for i in range(cnt)]


def clean_data(x):
    a, pa = _arr(x)
    out = np.empty(max(len(a), 1), dtype=np.float64)
    k = lib().orc_clean_data(pa, len(a), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:k].copy()


def gibbs_sizing(x):
    a, pa = _arr(x)
    out = np.empty(next_size(len(a)), dtype=np.float64)
    k = lib().orc_gibbs_sizing(pa, len(a), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:k]


def noop(x):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_noop(pa, len(a), C.byref(b))
    return _take(b)


def constant(x):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_constant(pa, len(a), C.byref(b))
    return _take(b)


def rle(x):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_rle(pa, len(a), C.byref(b))
    return _take(b)


def fft(x):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_fft(pa, len(a), C.byref(b))
    return _take(b)


def fft_set(x, freqs):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_fft_set(pa, len(a), freqs, C.byref(b))
    return _take(b)


def polynomial(x, idw=False):
    a, pa = _arr(x)
    b = _Buf()
    lib().orc_polynomial(pa, len(a), int(idw), C.byref(b))
    return _take(b)


def fft_allowed_error(x, max_err):
    a, pa = _arr(x)
    b = _Buf()
    e = C.c_double()
    it = C.c_int()
    lib().orc_fft_allowed_error(pa, len(a), max_err, C.byref(b), C.byref(e), C.byref(it))
    return _take(b), e.value, it.value


def polynomial_allowed_error(x, max_err, idw=False):
    a, pa = _arr(x)
    b = _Buf()
    e = C.c_double()
    it = C.c_int()
    lib().orc_polynomial_allowed_error(
        pa, len(a), max_err, int(idw), C.byref(b), C.byref(e), C.byref(it))
    return _take(b), e.value, it.value


def compress(compressor, x, bounded=False, max_err=0.0):
    a, pa = _arr(x)
    b = _Buf()
    e = C.c_double()
    rc = lib().orc_compress(compressor, pa, len(a), int(bounded), max_err, C.byref(b), C.byref(e))
    if rc:
        raise RuntimeError("orc_compress rc=%d" % rc)
    return _take(b), e.value


def decompress(compressor, data, samples):
    arr, p = _bytes_ptr(data)
    out = C.POINTER(C.c_double)()
    n = C.c_size_t()
    rc = lib().orc_decompress(compressor, p, len(data), samples, C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError("orc_decompress rc=%d" % rc)
    res = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
    lib().orc_free(out)
    return res


def compress_best(x, max_error, level=0):
    """max_error is the f32 the reference passes (e as f32 / 100.0)."""
    a, pa = _arr(x)
    b = _Buf()
    ch = C.c_int()
    e = C.c_double()
    rc = lib().orc_compress_best(
        pa, len(a), np.float32(max_error), level, C.byref(b), C.byref(ch), C.byref(e))
    if rc:
        raise RuntimeError("orc_compress_best rc=%d" % rc)
    return _take(b), ch.value, e.value


def stream_compress(x, chunk_off, compressor, bounded, max_error=0.0, level=0):
    a, pa = _arr(x)
    off = np.ascontiguousarray(np.asarray(chunk_off, dtype=np.uint64))
    nch = len(off) - 1
    b = _Buf()
    chosen = np.zeros(max(nch, 1), dtype=np.uint8)
    errs = np.zeros(max(nch, 1), dtype=np.float64)
    rc = lib().orc_stream_compress(
        pa, off.ctypes.data_as(C.POINTER(C.c_uint64)), nch, compressor, int(bounded),
        np.float32(max_error), level, C.byref(b),
        chosen.ctypes.data_as(C.POINTER(C.c_uint8)), errs.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError("orc_stream_compress rc=%d" % rc)
    return _take(b), chosen[:nch], errs[:nch]


def heap_order(norms, k):
    """Pop order (bin positions) of the reference's BinaryHeap over bins with these f32 norms."""
    a = np.ascontiguousarray(np.asarray(norms, dtype=np.float32))
    out = np.zeros(max(k, 1), dtype=np.uint32)
    L = lib()
    L.orc_heap_order.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32)]
    L.orc_heap_order.restype = C.c_int
    rc = L.orc_heap_order(a.ctypes.data_as(C.POINTER(C.c_float)), len(a), k, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc:
        raise RuntimeError("orc_heap_order rc=%d" % rc)
    return out[:k]


def compress_data(x, compressor, cli_error=3, level=0):
    a, pa = _arr(x)
    b = _Buf()
    rc = lib().orc_compress_data(pa, len(a), compressor, cli_error, level, C.byref(b))
    if rc:
        raise RuntimeError("orc_compress_data rc=%d" % rc)
    return _take(b)


def decompress_data(bro):
    arr, p = _bytes_ptr(bro)
    out = C.POINTER(C.c_double)()
    n = C.c_size_t()
    rc = lib().orc_decompress_data(p, len(bro), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError("orc_decompress_data rc=%d" % rc)
    res = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
    lib().orc_free(out)
    return res
