"""CPU tests of the csv-compressor front end: the VSRI index, the sample CSV
files and Metric in libatsc_hip.so (atsc_vsri.cpp, host code -- no GPU needed) against the Python
restatement in oracle/vsri_oracle.py, both pinned by the reference's own vectors (vsri/README.md
index example, csv-compressor/src/csv.rs:66-119)."""
import datetime
import os
import random
import struct

import numpy as np
import pytest

from oracle import vsri_oracle as VO
from tests.golden import kat as K


@pytest.fixture(scope="module")
def A():
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    return atsc_amd


def _both(A, points):
    """Feeds the same points to the library and to the oracle; returns (lib index, oracle index)."""
    v, o = A.Vsri(), VO.Vsri()
    for y in points:
        e_lib = e_orc = False
        try:
            v.update_for_point(y)
        except A.AtscError:
            e_lib = True
        try:
            o.update_for_point(y)
        except (VO.UpdateIndexForPointError, VO.Panic):
            e_orc = True
        assert e_lib == e_orc, y
    return v, o


def _opt(fn, *a):
    """(value | None | 'panic') for either implementation"""
    try:
        return fn(*a)
    except Exception as e:  # AtscError / Panic
        assert type(e).__name__ in ("AtscError", "Panic"), e
        return "panic"


def _same_index(v, o):
    assert v.min() == o.min() and v.max() == o.max()
    assert v.segments() == [list(s) for s in o.vsri_segments]
    assert v.get_sample_count() == o.get_sample_count()


# ---- the reference's vectors pin the oracle and the library ------------------------------------
def test_readme_index_example_oracle():
    o = VO.Vsri()
    for y in K.VSRI_README_POINTS:
        o.update_for_point(y)
    assert o.to_text() == K.VSRI_README_TEXT
    back = VO.Vsri.from_text(K.VSRI_README_TEXT)
    assert back.vsri_segments == o.vsri_segments and (back.min(), back.max()) == (55745, 59435)
    assert back.get_all_timestamps() == K.VSRI_README_POINTS


def test_readme_index_example_library(A, tmp_path):
    v = A.Vsri()
    for y in K.VSRI_README_POINTS:
        v.update_for_point(y)
    p = tmp_path / "m.vsri"
    v.flush_to(p)
    assert p.read_text() == K.VSRI_README_TEXT
    w = A.Vsri.load(p)
    assert w.segments() == [[15, 0, 55745, 166], [15, 166, 58505, 63]]
    assert w.get_all_timestamps() == K.VSRI_README_POINTS
    assert w.get_sample(55745) == 0 and w.get_sample(55760) == 1 and w.get_sample(58505) == 166
    assert w.get_sample(58300) is None           # in the gap
    assert w.get_this_or_next(58300) == 166 and w.get_this_or_previous(58300) == 165
    assert w.is_empty([58230, 58400]) and not w.is_empty([58000, 58600])


def test_samples_csv_kat(A, tmp_path):
    ts, vals = K.SAMPLES_CSV
    assert VO.samples_to_csv_text(ts, vals) == K.SAMPLES_CSV_TEXT
    p = tmp_path / "samples.csv"
    A.write_samples_to_csv_file(p, ts, vals)
    assert p.read_text() == K.SAMPLES_CSV_TEXT
    rt, rv = A.read_samples_from_csv_file(p)
    assert list(rt) == ts and list(rv) == vals


# ---- library against the restatement ------------------------------------------------------------
def _sequence(rng, n):
    """Monotone timestamps with rate changes, gaps and repeats; now and then a step back (rejected)."""
    y, step, out = rng.randrange(0, 4000), rng.choice([1, 5, 15, 60]), []
    for _ in range(n):
        out.append(y)
        r = rng.random()
        if r < 0.04:
            step = rng.choice([1, 2, 5, 15, 30, 60])
        if r > 0.97:
            y += rng.randrange(2, 50) * step          # gap
        elif r > 0.95:
            pass                                      # the same second again (lib.rs:239 TODO #11)
        elif r > 0.94:
            out.append(y - rng.randrange(1, 10))      # a point in the past
            y += step
        else:
            y += step
    return out


@pytest.mark.parametrize("seed", range(12))
def test_index_fuzz_against_oracle(A, seed, tmp_path):
    rng = random.Random(seed)
    pts = _sequence(rng, rng.choice([1, 2, 3, 10, 100, 600]))
    v, o = _both(A, pts)
    _same_index(v, o)
    lo, hi = min(pts) - 20, max(pts) + 20
    for y in [rng.randrange(lo, hi + 1) for _ in range(300)] + pts[:50]:
        for name in ("get_sample", "get_next_sample", "get_previous_sample", "get_this_or_next",
                     "get_this_or_previous"):
            assert _opt(getattr(v, name), y) == _opt(getattr(o, name), y), (name, y)
    for x in list(range(-2, 12)) + [rng.randrange(0, o.get_sample_count() + 5) for _ in range(100)]:
        assert _opt(v.get_time, x) == _opt(o.get_time, x), x
    for _ in range(200):
        a = rng.randrange(lo, hi)
        seg = [a, a + rng.randrange(0, 200)]
        assert v.is_empty(seg) == o.is_empty(seg), seg
    assert v.get_all_timestamps() == o.get_all_timestamps()
    p = tmp_path / "i.vsri"
    v.flush_to(p)
    assert p.read_text() == o.to_text()
    _same_index(A.Vsri.load(p), VO.Vsri.from_text(p.read_text()))


def test_index_edge_cases(A):
    v, o = _both(A, [])
    _same_index(v, o)
    assert v.get_time(0) == o.get_time(0) == 0 and v.is_empty([0, 5]) and o.is_empty([0, 5])
    # one point: a "fake" segment with m = 0 -> the reference divides by zero on a direct hit
    v, o = _both(A, [100])
    assert _opt(v.get_sample, 100) == _opt(o.get_sample, 100) == "panic"
    assert v.get_sample(99) is None and v.get_sample(101) is None
    # same second twice keeps m = 0 and the count at 2 (generate_segment, lib.rs:362-375)
    v, o = _both(A, [100, 100, 100, 130, 160])
    _same_index(v, o)
    # wrapping i32 arithmetic near the top of the range
    v, o = _both(A, [2147483000, 2147483600, 2147483647])
    _same_index(v, o)
    for y in (2147483000, 2147483600, 2147483647, -5):
        assert _opt(v.get_sample, y) == _opt(o.get_sample, y)
    # a point in the past
    v = A.Vsri()
    v.update_for_point(50)
    with pytest.raises(A.AtscError):
        v.update_for_point(49)
    with pytest.raises(A.AtscError):
        A.Vsri().update_for_point(-1)  # max_ts starts at 0


@pytest.mark.parametrize("text", ["", "5\n", "5\n9\n", "x\n9\n", "5\n9\n1,2,3\n", "5\n9\n1,2,3,4,5\n",
                                  "5\n9\n1,2,,4\n", "5\n9\n\n1,2,3,4\n", " 5 \n+9\n 1, 2 ,3,4 \r\n",
                                  "5\n9\n1,2,3,4", "99999999999\n1\n"])
def test_index_load_text_forms(A, tmp_path, text):
    p = tmp_path / "t.vsri"
    p.write_bytes(text.encode())
    try:
        o = VO.Vsri.from_text(text)
    except VO.Panic:
        o = None
    if o is None:
        with pytest.raises(A.AtscError):
            A.Vsri.load(p)
    else:
        _same_index(A.Vsri.load(p), o)
    with pytest.raises(A.AtscError):
        A.Vsri.load(tmp_path / "missing.vsri")


def test_day_elapsed_seconds(A):
    rng = random.Random(3)
    for ts in [0, 1, 86399, 86400, 1700000000, -1, -86400, -86401, 8210266876799, -8334601228800] + \
              [rng.randrange(-10**10, 10**10) for _ in range(200)]:
        assert A.day_elapsed_seconds(ts) == VO.day_elapsed_seconds(ts)
        if 0 <= ts < 250000000000:
            d = datetime.datetime.fromtimestamp(ts, datetime.timezone.utc)
            assert A.day_elapsed_seconds(ts) == d.hour * 3600 + d.minute * 60 + d.second
    for ts in (8210266876800, -8334601228801):
        with pytest.raises(A.AtscError):
            A.day_elapsed_seconds(ts)
        with pytest.raises(VO.Panic):
            VO.day_elapsed_seconds(ts)


def test_float_formatting_matches_ryu_layout(A, tmp_path):
    cases = [(1.0, "1.0"), (100000.0, "100000.0"), (1e15, "1000000000000000.0"), (1e16, "1e16"),
             (1.5e16, "1.5e16"), (9007199254740992.0, "9007199254740992.0"), (12.34, "12.34"),
             (0.001234, "0.001234"), (0.00001, "0.00001"), (0.000001, "1e-6"), (1.5e-7, "1.5e-7"),
             (-0.0, "-0.0"), (0.0, "0.0"), (-2.5, "-2.5"), (1.2345678901234568e17, "1.2345678901234568e17"),
             (5e-324, "5e-324"), (1.7976931348623157e308, "1.7976931348623157e308"),
             (float("inf"), "inf"), (float("-inf"), "-inf"), (float("nan"), "NaN")]
    rng = random.Random(11)
    vals = [c[0] for c in cases]
    vals += [struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0] for _ in range(3000)]
    vals += [rng.uniform(-1e6, 1e6) for _ in range(1000)] + [round(rng.uniform(0, 100), 2) for _ in range(1000)]
    for v, want in cases:
        assert VO.format_f64(v) == want
    p = tmp_path / "f.csv"
    A.write_samples_to_csv_file(p, list(range(len(vals))), vals)
    lines = p.read_text().split("\n")
    assert lines[0] == "timestamp,value" and lines[-1] == ""
    for i, v in enumerate(vals):
        t, s = lines[1 + i].split(",")
        assert int(t) == i and s == VO.format_f64(v), (v, s)
    # and the reader gives every finite value back bit for bit
    rt, rv = A.read_samples_from_csv_file(p)
    assert np.array_equal(rt, np.arange(len(vals)))
    assert np.array_equal(np.array(vals).view(np.uint64)[np.isfinite(vals)], rv.view(np.uint64)[np.isfinite(vals)])
    assert np.array_equal(np.isnan(vals), np.isnan(rv))
    A.write_samples_to_csv_file(p, [], [])
    assert p.read_text() == ""  # no record, no header (csv.rs:48-56)


def test_samples_csv_reader_forms(A, tmp_path):
    p = tmp_path / "r.csv"

    def rd(text):
        p.write_bytes(text.encode())
        t, v = A.read_samples_from_csv_file(p)
        return list(t), list(v)

    assert rd("timestamp,value\n") == ([], [])
    assert rd("") == ([], [])
    assert rd("value,timestamp\n1.5,7\n") == ([7], [1.5])                    # by header name
    assert rd("host,timestamp,value\nweb,1,2\nweb,2,3\n") == ([1, 2], [2.0, 3.0])   # extra column
    assert rd('timestamp,value\r\n"10","1e3"\r\n\r\n-4,+.5\r\n') == ([10, -4], [1000.0, 0.5])
    assert rd("timestamp,value\n9223372036854775807,inf") == ([9223372036854775807], [float("inf")])
    for bad in ("timestamp,value\n1.0,2\n", "timestamp,value\n1\n", "timestamp,value\n1, 2\n",
                "time,value\n1,2\n", "timestamp,value\n9223372036854775808,1\n", "timestamp,value\n1,abc\n"):
        p.write_bytes(bad.encode())
        with pytest.raises(A.AtscError):
            A.read_samples_from_csv_file(p)
    with pytest.raises(A.AtscError):
        A.read_samples_from_csv_file(tmp_path / "missing.csv")


def test_metric_index_and_times(A):
    rng = random.Random(5)
    day = 1700006400  # a midnight UTC
    ms = []
    t = day * 1000 + 3000
    for i in range(500):
        ms.append(t + rng.randrange(0, 1000))       # sub-second jitter disappears in the /1000
        t += 15000 if i != 250 else 15000 * 40       # one gap
    vals = np.arange(len(ms), dtype=np.float64)
    m = A.Metric.from_samples(ms, vals)
    o = VO.metric_from_samples(ms)
    _same_index(m.vsri, o)
    lt, lv = m.get_samples()
    assert list(lt) == VO.metric_sample_times(o, len(ms)) and np.array_equal(lv, vals)
    # a sample of the next day falls back to a small second-of-day: Error::UpdateForPointError
    with pytest.raises(A.AtscError):
        m.append_samples([(day + 86400 + 5) * 1000], [1.0])
    with pytest.raises(VO.UpdateIndexForPointError):
        VO.metric_from_samples(ms + [(day + 86400 + 5) * 1000])
