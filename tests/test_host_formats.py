"""CPU-only tests of the host-side format code in libatsc_hip.so (atsc_stream.cpp): WBRO (rkyv)
reader/writer against the reference's fixtures and known-answer bytes, the CSV reader, BRO sniffing."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import kat as K


@pytest.fixture(scope="module")
def A():
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    return atsc_amd


@pytest.mark.parametrize("name,count", [("go_gc_heap_goal_bytes", 2953), ("memory_used", 2301), ("uptime", 2301)])
def test_wbro_fixtures_roundtrip_bytes(A, golden_dir, name, count):
    path = os.path.join(golden_dir, "wbros", name + ".wbro")
    raw = open(path, "rb").read()
    d = A.wbro_from_bytes(raw)
    assert len(d) == count
    ref = H.read_wbro(path)  # independent python parser of the rkyv layout
    assert np.array_equal(d, ref, equal_nan=True)
    assert np.array_equal(A.wbro_read(path), ref, equal_nan=True)
    # the writer reproduces the reference's file byte for byte (wavbrro.rs:114-132)
    assert A.wbro_to_bytes(d) == raw


def test_wbro_one_sample_kat(A):
    # wavbrro.rs:223-233
    assert A.wbro_to_bytes([1.0]) == b"WBRO0000WBRO" + bytes(K.WBRO_ONE_SAMPLE)
    assert list(A.wbro_from_bytes(b"WBRO0000WBRO" + bytes(K.WBRO_ONE_SAMPLE))) == [1.0]


def test_wbro_write_read(A, tmp_path):
    # wavbrro.rs:236-271
    p = tmp_path / "test.wbro"
    A.wbro_write(p, [1.0, 2.0, 3.0])
    assert list(A.wbro_read(p)) == [1.0, 2.0, 3.0]
    big = np.arange(5000, dtype=np.float64) * 0.5
    A.wbro_write(p, big)
    assert np.array_equal(A.wbro_read(p), big)
    assert np.array_equal(H.read_wbro(str(p)), big)


def test_wbro_rejects_other_files(A, tmp_path):
    with pytest.raises(A.AtscError) as ei:
        A.wbro_from_bytes(b"RIFF" + bytes(64))
    assert ei.value.rc == A.capi.E_FORMAT  # wavbrro Error::FormatError
    with pytest.raises(A.AtscError) as ei:
        A.wbro_read(tmp_path / "missing.wbro")
    assert ei.value.rc == A.capi.E_IO


def test_csv_fixtures(A, golden_dir):
    for name in ("cpu_utilization.csv", "iowait.csv"):
        p = os.path.join(golden_dir, "csv", name)
        got = A.csv_read(p, header=True, time_field="time", value_field="value")
        assert np.array_equal(got, H.read_csv_values(p))
    p = os.path.join(golden_dir, "csv", "cpu_utilization_no_headers_only_values.csv")
    got = A.csv_read(p, header=False)
    assert np.array_equal(got, H.read_csv_values(p, header=False))
    assert len(got) == 2854


def test_csv_semantics(A, tmp_path):
    # atsc/src/csv.rs:112-251
    p = tmp_path / "a.csv"
    p.write_text("time,value\n1,1.5\n2,-2e3\n\n3,inf\n4,NaN\n5,.5\n6,+7.\n")
    v = A.csv_read(p)
    assert v[0] == 1.5 and v[1] == -2000.0 and np.isinf(v[2]) and np.isnan(v[3]) and v[4] == 0.5 and v[5] == 7.0
    p.write_text("ts,val\n1,2\n")
    assert list(A.csv_read(p, time_field="ts", value_field="val")) == [2.0]
    for fields in (("time", "val"), ("ts", "value")):  # field not found
        with pytest.raises(A.AtscError):
            A.csv_read(p, time_field=fields[0], value_field=fields[1])
    p.write_text("time,value\n1, 2\n")  # Rust's f64 parser rejects surrounding whitespace
    with pytest.raises(A.AtscError):
        A.csv_read(p)
    p.write_text("time,value\n1,0x10\n")
    with pytest.raises(A.AtscError):
        A.csv_read(p)
    p.write_text("3.5\n4.5\n")
    assert list(A.csv_read(p, header=False)) == [3.5, 4.5]


def test_bro_read_file(A, tmp_path, golden_dir):
    # utils/readers/bro_reader.rs:31-46
    p = tmp_path / "x.bro"
    p.write_bytes(bytes(K.STREAM_CONSTANT_1024))
    assert A.bro_read_file(p) == bytes(K.STREAM_CONSTANT_1024)
    assert A.bro_read_file(os.path.join(golden_dir, "wbros", "uptime.wbro")) is None  # skipped silently
    p.write_bytes(b"BRRO")
    with pytest.raises(A.AtscError):
        A.bro_read_file(p)  # shorter than the 12-byte probe
