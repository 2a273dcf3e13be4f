"""Python face of the csv-compressor front end in libatsc_hip.so (atsc_vsri.cpp): the VSRI timestamp
index (vsri/src/lib.rs), the `timestamp,value` sample files (csv-compressor/src/csv.rs) and Metric
(csv-compressor/src/metric.rs).  Same names and argument meaning as the reference; where the
reference returns Option, None comes back; where it panics, AtscError is raised.  ctypes plumbing."""
import ctypes as C

import numpy as np

from . import capi


def day_elapsed_seconds(timestamp_sec):
    """vsri/src/lib.rs:49-57"""
    out = C.c_int32()
    capi.check(capi.lib().atsc_day_elapsed_seconds(int(timestamp_sec), C.byref(out)))
    return out.value


class Vsri:
    """vsri/src/lib.rs:100-486"""

    def __init__(self, _handle=None):
        self._h = C.c_void_p(_handle if _handle is not None else capi.lib().atsc_vsri_new())
        if not self._h:
            raise MemoryError("atsc_vsri_new")

    def __del__(self):
        try:
            if self._h:
                capi.lib().atsc_vsri_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @classmethod
    def load(cls, filename):
        h = C.c_void_p()
        capi.check(capi.lib().atsc_vsri_load(str(filename).encode(), C.byref(h)))
        return cls(h.value)

    def flush_to(self, path):
        capi.check(capi.lib().atsc_vsri_flush_to(self._h, str(path).encode()))

    def update_for_point(self, y):
        capi.check(capi.lib().atsc_vsri_update_for_point(self._h, int(y)))

    def min(self):
        return int(capi.lib().atsc_vsri_min(self._h))

    def max(self):
        return int(capi.lib().atsc_vsri_max(self._h))

    def segments(self):
        out = []
        seg = (C.c_int32 * 4)()
        for i in range(int(capi.lib().atsc_vsri_segment_count(self._h))):
            capi.check(capi.lib().atsc_vsri_segment(self._h, i, seg))
            out.append([int(v) for v in seg])
        return out

    def get_sample_count(self):
        return int(capi.lib().atsc_vsri_get_sample_count(self._h))

    def _opt(self, fn, arg):
        out = C.c_int32()
        r = fn(self._h, int(arg), C.byref(out))
        if r < 0:
            raise capi.AtscError(r)
        return out.value if r == 1 else None

    def get_sample(self, y):
        return self._opt(capi.lib().atsc_vsri_get_sample, y)

    def get_next_sample(self, y):
        return self._opt(capi.lib().atsc_vsri_get_next_sample, y)

    def get_previous_sample(self, y):
        return self._opt(capi.lib().atsc_vsri_get_previous_sample, y)

    def get_this_or_next(self, y):
        return self._opt(capi.lib().atsc_vsri_get_this_or_next, y)

    def get_this_or_previous(self, y):
        return self._opt(capi.lib().atsc_vsri_get_this_or_previous, y)

    def get_time(self, x):
        return self._opt(capi.lib().atsc_vsri_get_time, x)

    def is_empty(self, time_segment):
        r = capi.lib().atsc_vsri_is_empty(self._h, int(time_segment[0]), int(time_segment[1]))
        if r < 0:
            raise capi.AtscError(r)
        return bool(r)

    def get_all_timestamps(self):
        p = C.POINTER(C.c_int32)()
        n = C.c_uint64()
        capi.check(capi.lib().atsc_vsri_get_all_timestamps(self._h, C.byref(p), C.byref(n)))
        out = [int(p[i]) for i in range(n.value)]
        capi.lib().atsc_free(p)
        return out


def read_samples_from_csv_file(path):
    """csv-compressor/src/csv.rs:41-45 -> (timestamps int64[n], values float64[n])"""
    pt = C.POINTER(C.c_int64)()
    pv = C.POINTER(C.c_double)()
    n = C.c_uint64()
    capi.check(capi.lib().atsc_samples_csv_read(str(path).encode(), C.byref(pt), C.byref(pv), C.byref(n)))
    ts = np.array([pt[i] for i in range(n.value)], dtype=np.int64)
    vals = np.array([pv[i] for i in range(n.value)], dtype=np.float64)
    capi.lib().atsc_free(pt)
    capi.lib().atsc_free(pv)
    return ts, vals


def write_samples_to_csv_file(path, timestamps, values):
    """csv-compressor/src/csv.rs:48-56"""
    ts = np.ascontiguousarray(np.asarray(timestamps, dtype=np.int64))
    vals = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
    assert len(ts) == len(vals)
    capi.check(capi.lib().atsc_samples_csv_write(
        str(path).encode(), ts.ctypes.data_as(C.POINTER(C.c_int64)),
        vals.ctypes.data_as(C.POINTER(C.c_double)), len(ts)))


class Metric:
    """csv-compressor/src/metric.rs:24-98: the samples' values (the WavBrro side) plus their index."""

    def __init__(self, values=None, vsri=None):
        self.values = np.asarray(values if values is not None else [], dtype=np.float64)
        self.vsri = vsri if vsri is not None else Vsri()

    @classmethod
    def from_samples(cls, timestamps_ms, values):
        m = cls()
        m.append_samples(timestamps_ms, values)
        return m

    def append_samples(self, timestamps_ms, values):
        ts = np.ascontiguousarray(np.asarray(timestamps_ms, dtype=np.int64))
        vals = np.asarray(values, dtype=np.float64)
        assert len(ts) == len(vals)
        bad = C.c_uint64()
        rc = capi.lib().atsc_metric_index_samples(self.vsri._h, ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  len(ts), C.byref(bad))
        if rc:
            # the reference has already pushed the samples before the offending one
            self.values = np.concatenate([self.values, vals[: bad.value]])
            raise capi.AtscError(rc, "updating for point failed, sample %d" % bad.value)
        self.values = np.concatenate([self.values, vals])

    def get_samples(self):
        n = len(self.values)
        out = np.zeros(n, dtype=np.int64)
        capi.check(capi.lib().atsc_metric_sample_times(self.vsri._h, n, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out, self.values.copy()
