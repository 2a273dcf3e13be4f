"""Pins the CPU oracle against every known-answer test the reference holds for the
hot path.  CPU only."""
import os
import struct

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import kat as K


def B(lst):
    return bytes(lst)


# ---- helpers ---------------------------------------------------------------

def test_next_size(oracle):
    for n, want in K.NEXT_SIZE:
        assert oracle.next_size(n) == want
    assert oracle.is_decomposable(2048) and oracle.is_decomposable(512)


def test_chunk_sizes(oracle):
    for n, want in K.CHUNK_SIZES:
        assert oracle.chunk_sizes(n) == want
    # optimizer/mod.rs:142-148: 2049 x 12.23 -> 2 chunks; :167-173: 132671 -> 4 chunks
    assert len(oracle.chunk_sizes(2049)) == 2
    assert len(oracle.chunk_sizes(132671)) == 4


def test_round_and_limit(oracle):
    for (x, mn, mx, d), want in K.ROUND_LIMIT:
        assert oracle.round_and_limit_f64(x, mn, mx, d) == want


def test_mape(oracle):
    a, b, want = K.MAPE
    assert oracle.error_mape(a, a) == 0.0
    assert oracle.error_mape(a, b) == want
    assert oracle.error_mape([1.0], [1.1]) < 0.101


def test_stats(oracle):
    for data, bd, mn, mx, mean, mnl, mxl, frac in K.STATS:
        s = oracle.stats(data)
        assert (s.bitdepth, s.min, s.max, s.mean, s.min_loc, s.max_loc, bool(s.fractional)) == (
            bd, mn, mx, mean, mnl, mxl, frac)


def test_stats_bitdepth_edges(oracle):
    # optimizer/utils.rs:91-113
    assert oracle.stats([0.0, 255.0]).bitdepth == oracle.BD_U8
    assert oracle.stats([0.0, 256.0]).bitdepth == oracle.BD_I16
    assert oracle.stats([-1.0, 5.0]).bitdepth == oracle.BD_I16
    assert oracle.stats([-32769.0, 5.0]).bitdepth == oracle.BD_I32
    assert oracle.stats([0.0, 40000.0]).bitdepth == oracle.BD_I32
    assert oracle.stats([0.0, 3e9]).bitdepth == oracle.BD_F64
    assert oracle.stats([0.5, 3.0]).bitdepth == oracle.BD_F64


def test_gibbs_sizing(oracle):
    # fft.rs:592-600
    v = np.full(2048, 2.0)
    v[0] = 1.0
    v[2047] = 3.0
    g = oracle.gibbs_sizing(v)
    assert len(g) == 2187 and g[2] == 1.0 and g[2185] == 3.0


# ---- byte-exact codec KATs -------------------------------------------------

def test_constant_bytes(oracle):
    for data, want in (K.CONSTANT_U8, K.CONSTANT_F64):
        assert oracle.constant(data) == B(want)
    out = oracle.decompress(oracle.CONSTANT, B(K.CONSTANT_U8[1]), 5)
    assert list(out) == K.CONSTANT_U8[0]


def test_noop_bytes(oracle):
    assert oracle.noop(K.NOOP[0]) == B(K.NOOP[1])
    enc = oracle.noop(K.NOOP_OPTIMIZE[0])
    assert list(oracle.decompress(oracle.NOOP, enc, 4)) == [float(v) for v in K.NOOP_OPTIMIZE[1]]
    v = [1.0, 2.0, 3.0, 4.0, 1.0]  # noop.rs:103-108
    assert list(oracle.decompress(oracle.NOOP, oracle.noop(v), 5)) == v


@pytest.mark.parametrize("name", ["RLE_CONSTANT", "RLE_SIMPLE", "RLE_U8", "RLE_F64"])
def test_rle_bytes_roundtrip(oracle, name):
    data, want = getattr(K, name)
    enc = oracle.rle(data)
    assert enc == B(want)
    assert list(oracle.decompress(oracle.RLE, enc, len(data))) == data


def test_rle_sparse(oracle):
    enc = oracle.rle(K.RLE_SPARSE)
    assert len(enc) < 16
    assert list(oracle.decompress(oracle.RLE, enc, len(K.RLE_SPARSE))) == K.RLE_SPARSE


@pytest.mark.parametrize("name,idw", [("POLY_U8", 0), ("POLY_I16", 0), ("POLY_I32", 0),
                                      ("POLY_F64", 0), ("IDW_U8", 1), ("POLY_LINE", 0),
                                      ("IDW_LINE", 1)])
def test_polynomial_bytes(oracle, name, idw):
    data, want = getattr(K, name)
    assert oracle.polynomial(data, idw=bool(idw)) == B(want)


@pytest.mark.parametrize("name,idw", [("POLY_CR_OUT", 0), ("POLY_LINEAR_OUT", 0), ("IDW_OUT", 1),
                                      ("IDW_LINEAR_OUT", 1)])
def test_polynomial_values(oracle, name, idw):
    data, want = getattr(K, name)
    enc = oracle.polynomial(data, idw=bool(idw))
    out = oracle.decompress(oracle.IDW if idw else oracle.POLYNOMIAL, enc, len(data))
    assert list(out) == want


def test_polynomial_allowed_error(oracle):
    # polynomial.rs:517-526 and :572-581
    enc, err, _ = oracle.polynomial_allowed_error(K.V17, 0.05)
    out = oracle.decompress(oracle.POLYNOMIAL, enc, len(K.V17))
    assert oracle.error_mape(K.V17, out) <= 0.05
    enc, err, _ = oracle.polynomial_allowed_error(K.V17, 0.02, idw=True)
    out = oracle.decompress(oracle.IDW, enc, len(K.V17))
    assert oracle.error_mape(K.V17, out) <= 0.02


def test_fft_set_bytes(oracle):
    """fft.rs:551-560.  K, positions, order, max/min are exact; the f32 bins are rustfft
    arithmetic [3P] whose butterfly order is CPU dependent -> pinned to 1 ulp."""
    data, want = K.FFT_SET2
    got = oracle.fft_set(data, 2)
    assert len(got) == len(want)
    fg, mxg, mng = H.parse_fft_payload(got)
    fw, mxw, mnw = H.parse_fft_payload(B(want))
    assert (mxg, mng) == (mxw, mnw)
    assert [f[0] for f in fg] == [f[0] for f in fw]
    for (_, reg, img), (_, rew, imw) in zip(fg, fw):
        for a, b in ((reg, rew), (img, imw)):
            ia = struct.unpack("<i", struct.pack("<f", a))[0]
            ib = struct.unpack("<i", struct.pack("<f", b))[0]
            assert abs(ia - ib) <= 1, (a, b)


def test_fft_lossless(oracle):
    # fft.rs:563-568
    enc = oracle.fft_set(K.V12, 12)
    assert list(oracle.decompress(oracle.FFT, enc, 12)) == K.V12


def test_fft_lossy_values(oracle):
    # fft.rs:571-579
    data, want = K.FFT_LOSSY_OUT
    out = oracle.decompress(oracle.FFT, oracle.fft(data), len(data))
    assert list(out) == want


def test_fft_allowed_error(oracle):
    # fft.rs:582-589
    enc, err, _ = oracle.fft_allowed_error(K.V12, 0.01)
    out = oracle.decompress(oracle.FFT, enc, 12)
    assert oracle.error_mape(K.V12, out) <= 0.01


def test_fft_static(oracle):
    # fft.rs:603-626: constant input -> zero frequencies, decodes to the constant
    v = [1.0] * 1024
    enc = oracle.fft(v)
    freqs, _, _ = H.parse_fft_payload(enc)
    assert len(freqs) == 0
    assert list(oracle.decompress(oracle.FFT, enc, 1024)) == v


# ---- stream level ----------------------------------------------------------

def test_stream_constant_bytes(oracle):
    v = [1.0] * 1024
    bro, chosen, _ = oracle.stream_compress(v, [0, 1024], oracle.CONSTANT, bounded=False)
    assert bro == B(K.STREAM_CONSTANT_1024)
    assert list(oracle.decompress_data(bro)) == v  # data.rs:167-176


def test_stream_higher_version_rejected(oracle):
    bro = bytearray(B(K.STREAM_CONSTANT_1024))
    bro[4] = 9  # header.rs:103-113
    with pytest.raises(RuntimeError):
        oracle.decompress_data(bytes(bro))
    bad = bytearray(B(K.STREAM_CONSTANT_1024))
    bad[0] = 0
    with pytest.raises(RuntimeError):
        oracle.decompress_data(bytes(bad))


def test_csv_constant_cli_kat(oracle, golden_dir):
    """BASELINE.json configs[0]: csv sample, --compressor constant (SURVEY 8(c) hand-derived)."""
    vals = H.read_csv_values(os.path.join(golden_dir, "csv", "cpu_utilization.csv"))
    assert len(vals) == 2854
    bro = oracle.compress_data(vals, oracle.CONSTANT)
    assert bro.hex() == K.CSV_CONSTANT_BRO_HEX
    out = oracle.decompress_data(bro)
    assert len(out) == 2854
    assert out[0] == 13.85002983491348 and out[2048] == 14.39248627554732


# ---- e2e.rs behaviours on the reference's fixture --------------------------

@pytest.fixture(scope="module")
def heap(golden_dir):
    d = H.read_wbro(os.path.join(golden_dir, "wbros", "go_gc_heap_goal_bytes.wbro"))
    assert len(d) == 2953
    return d


@pytest.mark.parametrize("comp", ["IDW", "POLYNOMIAL", "NOOP", "RLE", "AUTO"])
def test_e2e_lossless(oracle, heap, comp):
    # e2e.rs:12-49,158-160 : --error 0 must round-trip bit-exactly
    bro = oracle.compress_data(heap, getattr(oracle, comp), cli_error=0)
    out = oracle.decompress_data(bro)
    assert np.array_equal(out, heap)


@pytest.mark.parametrize("comp", ["IDW", "POLYNOMIAL", "FFT", "AUTO"])
def test_e2e_lossy(oracle, heap, comp):
    # e2e.rs:17-54,162-164,234-248 : --error 5 -> MAPE <= 0.05
    bro = oracle.compress_data(heap, getattr(oracle, comp), cli_error=5)
    out = oracle.decompress_data(bro)
    assert len(out) == len(heap)
    assert oracle.error_mape(heap, out) <= 0.05


def test_e2e_uptime_noop(oracle, golden_dir):
    # e2e.rs:166-185
    d = H.read_wbro(os.path.join(golden_dir, "wbros", "uptime.wbro"))
    out = oracle.decompress_data(oracle.compress_data(d, oracle.NOOP))
    assert np.array_equal(out, d)


def test_memory_used_nan_dropped(oracle, golden_dir):
    # optimizer/mod.rs:64-71: the trailing NaN is removed before chunking
    d = H.read_wbro(os.path.join(golden_dir, "wbros", "memory_used.wbro"))
    assert np.isnan(d[-1]) and len(d) == 2301
    out = oracle.decompress_data(oracle.compress_data(d, oracle.AUTO, cli_error=5))
    assert len(out) == 2300


def test_all_compressors_all_levels_run(oracle, heap):
    # integration_test.rs:59-106
    for comp in ("AUTO", "NOOP", "FFT", "CONSTANT", "POLYNOMIAL", "IDW", "RLE"):
        fc, frames = H.parse_bro(oracle.compress_data(heap, getattr(oracle, comp), cli_error=3))
        assert fc == 3 and [f[1] for f in frames] == [2048, 512, 393]
    for level in range(7):
        fc, frames = H.parse_bro(oracle.compress_data(heap, oracle.AUTO, cli_error=3, level=level))
        assert fc == 3


def test_wbro_fixture_layout(golden_dir):
    # wavbrro.rs:223-233 one-sample archive parses with the test-side reader
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".wbro", delete=False) as f:
        f.write(b"WBRO0000WBRO" + bytes(K.WBRO_ONE_SAMPLE))
        p = f.name
    try:
        assert list(H.read_wbro(p)) == [1.0]
    finally:
        os.unlink(p)
