"""Python face of the host-side mirror in libatsc_hip.so (atsc_stream.cpp): same names and argument
meaning as the reference's CompressedStream (atsc/src/data.rs) and the helpers of atsc/src/main.rs,
wavbrro/ and atsc/src/csv.rs.  Pure ctypes plumbing."""
import ctypes as C

import numpy as np

from . import capi


def _f64(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _take_bytes(p, n):
    out = bytes(C.string_at(p, n.value)) if n.value else b""
    capi.lib().atsc_free(p)
    return out


def _take_f64(p, n):
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[: n.value].copy()
    capi.lib().atsc_free(p)
    return out


class CompressedStream:
    """atsc/src/data.rs:29-110"""

    def __init__(self, ctx, _handle=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if _handle is None:
            capi.check(capi.lib().atsc_stream_new(ctx._h, C.byref(self._h)), ctx._h)
        else:
            self._h = _handle

    def __del__(self):
        try:
            if self._h:
                capi.lib().atsc_stream_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def compress_chunk(self, chunk):
        a, p = _f64(chunk)
        capi.check(capi.lib().atsc_stream_compress_chunk(self._h, p, len(a)), self.ctx._h)

    def compress_chunk_with(self, chunk, compressor):
        a, p = _f64(chunk)
        capi.check(capi.lib().atsc_stream_compress_chunk_with(self._h, p, len(a), int(compressor)), self.ctx._h)

    def compress_chunk_bounded_with(self, chunk, compressor, max_error, compression_speed=0):
        a, p = _f64(chunk)
        capi.check(capi.lib().atsc_stream_compress_chunk_bounded_with(
            self._h, p, len(a), int(compressor), C.c_float(np.float32(max_error)), int(compression_speed)),
            self.ctx._h)

    @property
    def frame_count(self):
        return int(capi.lib().atsc_stream_frame_count(self._h))

    def to_bytes(self):
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        capi.check(capi.lib().atsc_stream_to_bytes(self._h, C.byref(p), C.byref(n)), self.ctx._h)
        return _take_bytes(p, n)

    @classmethod
    def from_bytes(cls, ctx, data):
        b = np.frombuffer(bytes(data), dtype=np.uint8)
        h = C.c_void_p()
        capi.check(capi.lib().atsc_stream_from_bytes(
            ctx._h, b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), C.byref(h)), ctx._h)
        return cls(ctx, h)

    def decompress(self):
        p = C.POINTER(C.c_double)()
        n = C.c_uint64()
        capi.check(capi.lib().atsc_stream_decompress(self._h, C.byref(p), C.byref(n)), self.ctx._h)
        return _take_f64(p, n)


def compress_data(ctx, vec, compressor=capi.AUTO, error=3, sample_level=0):
    """atsc/src/main.rs:130-165"""
    a, pa = _f64(vec)
    p = C.POINTER(C.c_uint8)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_compress_data(ctx._h, pa, len(a), int(compressor), int(error), int(sample_level),
                                             C.byref(p), C.byref(n)), ctx._h)
    return _take_bytes(p, n)


def decompress_data(ctx, bro):
    """atsc/src/main.rs:168-172"""
    b = np.frombuffer(bytes(bro), dtype=np.uint8)
    p = C.POINTER(C.c_double)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_decompress_data(ctx._h, b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b),
                                               C.byref(p), C.byref(n)), ctx._h)
    return _take_f64(p, n)


def wbro_from_bytes(data):
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    p = C.POINTER(C.c_double)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_wbro_from_bytes(b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), C.byref(p), C.byref(n)))
    return _take_f64(p, n)


def wbro_to_bytes(samples):
    a, pa = _f64(samples)
    p = C.POINTER(C.c_uint8)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_wbro_to_bytes(pa, len(a), C.byref(p), C.byref(n)))
    return _take_bytes(p, n)


def wbro_read(path):
    p = C.POINTER(C.c_double)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_wbro_read(str(path).encode(), C.byref(p), C.byref(n)))
    return _take_f64(p, n)


def wbro_write(path, samples):
    a, pa = _f64(samples)
    capi.check(capi.lib().atsc_wbro_write(str(path).encode(), pa, len(a)))


def bro_read_file(path):
    """-> bytes, or None when the file does not start with "BRRO" (bro_reader.rs:31-38)"""
    p = C.POINTER(C.c_uint8)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_bro_read_file(str(path).encode(), C.byref(p), C.byref(n)))
    if not p:
        return None
    return _take_bytes(p, n)


def csv_read(path, header=True, time_field="time", value_field="value"):
    p = C.POINTER(C.c_double)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_csv_read(str(path).encode(), int(bool(header)), time_field.encode(),
                                        value_field.encode(), C.byref(p), C.byref(n)))
    return _take_f64(p, n)
