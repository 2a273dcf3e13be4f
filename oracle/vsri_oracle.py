"""TEST INFRASTRUCTURE -- CPU restatement of the csv-compressor front end, pure Python.

Follows vsri/src/lib.rs (the VSRI index), csv-compressor/src/csv.rs (sample files) and
csv-compressor/src/metric.rs, statement by statement, with Rust's release-mode i32 arithmetic
(wrapping add/sub/mul, truncating division, panic on /0).  Only tests/ may import it; the product
(atsc_amd/, libatsc_hip.so) never does.

Pinned by: the index example in vsri/README.md and vsri/src/lib.rs:37-41 (tests/golden/kat.py
VSRI_README_*), and the three csv.rs tests (csv-compressor/src/csv.rs:66-119).  The reference
holds no other vector for this front end (vsri/src/lib.rs has no tests): beyond those the parity of
this restatement is unpinned, and DESIGN.md says so.
"""
import math


class Panic(Exception):
    """Where the reference would panic (unwrap on None / Err, integer division by zero)."""


class UpdateIndexForPointError(Exception):
    """vsri/src/lib.rs:494-497"""


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def _div(a, b):
    if b == 0:
        raise Panic("attempt to divide by zero")
    if a == -(1 << 31) and b == -1:
        raise Panic("attempt to divide with overflow")
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def day_elapsed_seconds(timestamp_sec):
    """lib.rs:49-57; chrono 0.4.38 DateTime<Utc> spans -262143-01-01 ..= +262142-12-31"""
    if not (-8334601228800 <= timestamp_sec <= 8210266876799):
        raise Panic("DateTime::from_timestamp -> None")
    return timestamp_sec % 86400  # hour*3600 + minute*60 + second of the (proleptic) day


class Vsri:
    def __init__(self):  # lib.rs:110-119
        self.min_ts = 0
        self.max_ts = 0
        self.vsri_segments = []

    # --- lib.rs:276-298
    def min(self):
        return self.min_ts

    def max(self):
        return self.max_ts

    @staticmethod
    def calculate_b(segment):
        return _i32(segment[2] - _i32(segment[0] * segment[1]))

    def current_segment(self):
        return list(self.vsri_segments[-1]) if self.vsri_segments else [0, 0, 0, 0]

    # --- lib.rs:236-273
    def update_for_point(self, y):
        if y < self.max_ts:
            raise UpdateIndexForPointError()
        self.max_ts = y
        segment_count = len(self.vsri_segments)
        if segment_count == 0:
            self.min_ts = y
            self.vsri_segments.append(self.create_fake_segment(y))
            return
        if self.is_fake_segment():
            self.vsri_segments[segment_count - 1] = self.generate_segment(y)
        else:
            if self.fits_segment(y):
                self.vsri_segments[segment_count - 1][3] = _i32(self.vsri_segments[segment_count - 1][3] + 1)
                return
            self.vsri_segments.append(self.create_fake_segment(y))

    def generate_segment(self, y):  # lib.rs:362-375
        last = self.current_segment()
        if last[0] != 0:
            return last
        return [_i32(y - last[2]), last[1], last[2], 2]

    def create_fake_segment(self, y):  # lib.rs:380-386
        seg = self.current_segment()
        return [0, _i32(seg[1] + seg[3]), y, 1]

    def is_fake_segment(self):  # lib.rs:389-392
        return self.current_segment()[0] == 0

    def fits_segment(self, y):  # lib.rs:394-413
        last = self.current_segment()
        b = self.calculate_b(last)
        x_value = _div(_i32(y - b), last[0])
        return x_value == _i32(last[3] + last[1])

    # --- look-ups
    def get_sample(self, y):  # lib.rs:301-317
        for s in self.vsri_segments:
            end = _i32(s[2] + _i32(s[0] * _i32(s[3] - 1)))
            if s[2] <= y <= end:
                return _div(_i32(y - self.calculate_b(s)), s[0])
        return None

    def get_next_sample(self, y):  # lib.rs:154-169
        if y < self.min():
            return 0
        if y >= self.max():
            return None
        for s in reversed(self.vsri_segments):
            if y <= s[2]:
                return s[1]
        return None

    def get_previous_sample(self, y):  # lib.rs:175-193
        if y < self.min():
            return None
        if y >= self.max():
            return self.get_sample_count()
        for s in self.vsri_segments:
            if y < s[2]:
                return _i32(s[1] - 1)
        return None

    def get_this_or_next(self, y):  # lib.rs:137-141
        r = self.get_sample(y)
        return r if r is not None else self.get_next_sample(y)

    def get_this_or_previous(self, y):  # lib.rs:144-148
        r = self.get_sample(y)
        p = self.get_previous_sample(y)
        return r if r is not None else p

    def get_sample_count(self):  # lib.rs:355-358
        last = self.current_segment()
        return _i32(last[3] + last[1])

    def get_time(self, x):  # lib.rs:320-341
        if x == 0:
            return self.min()
        if x > self.get_sample_count():
            return None
        if x == self.get_sample_count():
            return self.max()
        for s in self.vsri_segments:
            if s[1] <= x < _i32(s[1] + s[3]):
                return _i32(s[2] + _i32(s[0] * x))
        return None

    def get_all_timestamps(self):  # lib.rs:344-353
        out = []
        for s in self.vsri_segments:
            out.extend(_i32(_i32(f * s[0]) + s[2]) for f in range(max(s[3], 0)))
        return out

    def is_empty(self, time_segment):  # lib.rs:198-232
        t0, t1 = time_segment
        if len(self.vsri_segments) == 1:
            if (self.min() <= t0 <= self.max()) or (self.min() <= t1 <= self.max()):
                return False
            if t0 < self.min() and t1 > self.max():
                return False
        else:
            previous_seg_end = 0
            for count, s in enumerate(self.vsri_segments):
                y0 = s[2]
                end = _i32(y0 + _i32(s[0] * _i32(s[3] - 1)))
                if count >= 1 and (t0 > previous_seg_end and t1 < y0):
                    return True
                if (y0 <= t0 < end) or (y0 <= t1 < end):
                    return False
                if t0 < y0 and t1 > end:
                    return False
                previous_seg_end = end
        return True

    # --- lib.rs:424-486
    def to_text(self):
        lines = ["%d" % self.min_ts, "%d" % self.max_ts]
        lines += ["%d,%d,%d,%d" % tuple(s) for s in self.vsri_segments]
        return "".join(l + "\n" for l in lines)

    @classmethod
    def from_text(cls, text):
        v = cls()
        lines = text.split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        for i, line in enumerate(l[:-1] if l.endswith("\r") else l for l in lines):
            try:
                if i == 0:
                    v.min_ts = _parse_i32(line.strip())
                elif i == 1:
                    v.max_ts = _parse_i32(line.strip())
                else:
                    vals = [_parse_i32(f.strip()) for f in line.split(",")]
                    if len(vals) != 4:
                        raise Panic("try_into [i32; 4]")
                    v.vsri_segments.append(vals)
            except ValueError:
                raise Panic("parse::<i32>")
        return v


def _parse_i32(s):
    body = s[1:] if s[:1] in "+-" else s
    if not body or not body.isascii() or not body.isdigit():
        raise ValueError(s)
    v = int(s)
    if not (-(1 << 31) <= v < (1 << 31)):
        raise ValueError(s)
    return v


# ---- csv-compressor/src/csv.rs ---------------------------------------------------------------
def format_f64(v):
    """ryu 1.0.18 Buffer::format, the csv crate's f64 serialiser."""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    if v == 0.0:
        return "-0.0" if math.copysign(1.0, v) < 0 else "0.0"
    sign = "-" if v < 0 else ""
    mant, exp = ("%r" % abs(v)), 0
    if "e" in mant:
        mant, e = mant.split("e")
        exp = int(e)
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    digits = (ip + fp).lstrip("0")
    point = len(ip) + exp - (len(ip + fp) - len((ip + fp).lstrip("0")))  # digits before the point
    digits = digits.rstrip("0") or "0"
    # value = 0.digits * 10^point ; kk = point
    length, kk = len(digits), point
    k = kk - length
    if 0 <= k and kk <= 16:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 16:
        return sign + digits[:kk] + "." + digits[kk:]
    if -5 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if length == 1:
        return sign + digits + "e%d" % (kk - 1)
    return sign + digits[0] + "." + digits[1:] + "e%d" % (kk - 1)


def samples_to_csv_text(timestamps, values):
    """csv.rs:48-56: the header goes out with the first record"""
    if len(timestamps) == 0:
        return ""
    return "timestamp,value\n" + "".join("%d,%s\n" % (int(t), format_f64(float(v))) for t, v in zip(timestamps, values))


# ---- csv-compressor/src/metric.rs -------------------------------------------------------------
def metric_from_samples(timestamps_ms):
    """metric.rs:53-74, index side"""
    v = Vsri()
    for t in timestamps_ms:
        t = int(t)
        sec = int(abs(t) // 1000) * (1 if t >= 0 else -1)  # i64 `/`: toward zero
        v.update_for_point(day_elapsed_seconds(sec))
    return v


def metric_sample_times(vsri, n):
    """metric.rs:83-97"""
    out = []
    for i in range(n):
        t = vsri.get_time(i)
        if t is None:
            raise Panic("get_time(%d) is None" % i)
        out.append(t)
    return out
