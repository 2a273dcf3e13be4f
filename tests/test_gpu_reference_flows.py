"""The reference's own stream / end-to-end / integration tests, re-run against the GPU path:
atsc/src/data.rs:113-176, atsc/src/header.rs:94-114, atsc/tests/e2e.rs, atsc/tests/integration_test.rs.
The e2e and integration flows drive the `atsc` binary (atsc_amd/bin/atsc) exactly as the reference's
tests drive CARGO_BIN_EXE_atsc."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import kat as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    return atsc_amd


@pytest.fixture(scope="module")
def ctx(A):
    c = A.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def atsc_bin(A):
    p = os.path.join(os.path.dirname(A.__file__), "bin", "atsc")
    assert os.path.exists(p), "atsc CLI was not built"
    return p


# ---- atsc/src/data.rs tests -----------------------------------------------------------------
def test_compress_chunk(A, ctx):
    cs = A.CompressedStream(ctx)
    cs.compress_chunk([1.0] * 5)
    assert cs.frame_count == 1


def test_compress_chunk_with(A, ctx):
    cs = A.CompressedStream(ctx)
    cs.compress_chunk_with([1.0] * 5, A.CONSTANT)
    assert cs.frame_count == 1


def test_to_bytes(A, ctx):
    cs = A.CompressedStream(ctx)
    cs.compress_chunk_with([1.0] * 1024, A.CONSTANT)
    assert cs.to_bytes() == bytes(K.STREAM_CONSTANT_1024)


def test_from_bytes_and_constant_decompression(A, ctx):
    cs = A.CompressedStream(ctx)
    cs.compress_chunk_with([1.0] * 1024, A.CONSTANT)
    b = cs.to_bytes()
    cs2 = A.CompressedStream.from_bytes(ctx, b)
    assert cs2.frame_count == 1
    assert list(cs2.decompress()) == [1.0] * 1024


def test_header_versions(A, ctx):
    # header.rs:94-114
    b = bytearray(K.STREAM_CONSTANT_1024)
    assert int.from_bytes(b[4:8], "little") == 1
    b[4] = 9
    with pytest.raises(A.AtscError) as ei:
        A.CompressedStream.from_bytes(ctx, bytes(b))
    assert ei.value.rc == A.capi.E_VERSION


def test_auto_without_bound_is_rejected(A, ctx):
    # Compressor::Auto => todo!() on the unbounded path (compressor/mod.rs:72)
    cs = A.CompressedStream(ctx)
    with pytest.raises(A.AtscError):
        cs.compress_chunk_with([1.0, 2.0], A.AUTO)


def test_mixed_stream_matches_oracle(A, ctx, oracle):
    """One stream, chunks with different compressors and bounds, call order kept."""
    x = H.synth_series(77, 4096, block=512)
    cs = A.CompressedStream(ctx)
    parts = []
    plan = [(0, 512, A.AUTO, True, 0.05), (512, 1024, A.RLE, False, 0.0), (1024, 1536, A.POLYNOMIAL, True, 0.01),
            (1536, 2048, A.NOOP, False, 0.0), (2048, 3072, A.AUTO, True, 0.05), (3072, 4096, A.CONSTANT, False, 0.0)]
    for a, b, comp, bounded, me in plan:
        if bounded:
            cs.compress_chunk_bounded_with(x[a:b], comp, me, 0)
        else:
            cs.compress_chunk_with(x[a:b], comp)
        bro, _, _ = oracle.stream_compress(x[a:b], [0, b - a], comp, bounded, me, 0)
        parts.append(bro[10:])  # strip "BRRO" ver count + varint(1)
    got = cs.to_bytes()
    fg = H.parse_bro(got)[1]
    fo = [H.parse_bro_body(p, with_count=False)[0] for p in parts]
    assert [f[:3] for f in fg] == [f[:3] for f in fo]
    for g, o in zip(fg, fo):
        if g[2] != oracle.FFT:
            assert g == o
    out = A.CompressedStream.from_bytes(ctx, got).decompress()
    assert len(out) == 4096


def test_compress_data_matches_oracle_on_fixtures(A, ctx, oracle, golden_dir):
    for name in ("go_gc_heap_goal_bytes", "memory_used", "uptime"):
        d = A.wbro_read(os.path.join(golden_dir, "wbros", name + ".wbro"))
        for comp, e in ((A.AUTO, 5), (A.AUTO, 0), (A.NOOP, 3), (A.RLE, 3), (A.CONSTANT, 3), (A.POLYNOMIAL, 3),
                        (A.IDW, 3)):
            got = A.compress_data(ctx, d, comp, e)
            ref = oracle.compress_data(d, comp, cli_error=e)
            fg, fo = H.parse_bro(got), H.parse_bro(ref)
            assert fg[0] == fo[0] and [f[:3] for f in fg[1]] == [f[:3] for f in fo[1]], (name, comp, e)
            if all(f[2] != oracle.FFT for f in fg[1]):
                assert got == ref, (name, comp, e)
            out = A.decompress_data(ctx, got)
            assert np.array_equal(out, oracle.decompress_data(got)) or comp in (A.AUTO,)


# ---- atsc/tests/e2e.rs ----------------------------------------------------------------------
def _run(atsc_bin, *args):
    r = subprocess.run([atsc_bin] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (args, r.stdout[-500:], r.stderr[-500:])
    return r


def _flow(A, atsc_bin, golden_dir, tmp_path, compressor, error):
    src = os.path.join(golden_dir, "wbros", "go_gc_heap_goal_bytes.wbro")
    work = tmp_path / "go_gc_heap_goal_bytes.wbro"
    shutil.copy(src, work)
    _run(atsc_bin, "--compressor", compressor, "--error", error, work)
    _run(atsc_bin, "-u", tmp_path / "go_gc_heap_goal_bytes.bro")
    return A.wbro_read(src), A.wbro_read(work)


@pytest.mark.parametrize("compressor", ["idw", "polynomial", "noop", "rle", "auto"])
def test_e2e_lossless(A, atsc_bin, golden_dir, tmp_path, compressor):
    # e2e.rs:12-49,158-160: --error 0 round-trips bit-exactly
    orig, got = _flow(A, atsc_bin, golden_dir, tmp_path, compressor, 0)
    assert np.array_equal(orig, got)


@pytest.mark.parametrize("compressor", ["idw", "polynomial", "fft", "auto"])
def test_e2e_lossy(A, atsc_bin, golden_dir, tmp_path, compressor):
    # e2e.rs:17-54,162-164,234-248: --error 5 keeps MAPE <= 0.05
    orig, got = _flow(A, atsc_bin, golden_dir, tmp_path, compressor, 5)
    assert len(orig) == len(got)
    assert H.mape(orig, got) <= 0.05


def test_e2e_uptime_noop(A, atsc_bin, golden_dir, tmp_path):
    # e2e.rs:166-185
    shutil.copy(os.path.join(golden_dir, "wbros", "uptime.wbro"), tmp_path / "uptime.wbro")
    _run(atsc_bin, "--compressor", "noop", tmp_path / "uptime.wbro")
    _run(atsc_bin, "-u", tmp_path / "uptime.bro")
    assert np.array_equal(A.wbro_read(os.path.join(golden_dir, "wbros", "uptime.wbro")),
                          A.wbro_read(tmp_path / "uptime.wbro"))


@pytest.mark.parametrize("name,extra", [("cpu_utilization.csv", []),
                                        ("cpu_utilization_no_headers_only_values.csv", ["--no-header"])])
def test_e2e_csv_noop(A, atsc_bin, golden_dir, tmp_path, name, extra):
    # e2e.rs:57-156: csv input, noop, decompressed samples equal round(value)
    shutil.copy(os.path.join(golden_dir, "csv", name), tmp_path / name)
    _run(atsc_bin, "--csv", *extra, "--compressor", "noop", tmp_path / name)
    stem = name[:-4]
    _run(atsc_bin, "-u", tmp_path / (stem + ".bro"))
    vals = H.read_csv_values(os.path.join(golden_dir, "csv", name), header=not extra)
    got = A.wbro_read(tmp_path / (stem + ".wbro"))
    assert np.array_equal(got, np.sign(vals) * np.floor(np.abs(vals) + 0.5))  # noop rounds to i64


def test_cli_csv_constant_kat(A, atsc_bin, golden_dir, tmp_path):
    # BASELINE.json configs[0]: the hand-derived 58-byte .bro
    shutil.copy(os.path.join(golden_dir, "csv", "cpu_utilization.csv"), tmp_path / "cpu_utilization.csv")
    _run(atsc_bin, "--csv", "--compressor", "constant", tmp_path / "cpu_utilization.csv")
    assert (tmp_path / "cpu_utilization.bro").read_bytes().hex() == K.CSV_CONSTANT_BRO_HEX


# ---- atsc/tests/integration_test.rs ---------------------------------------------------------
@pytest.mark.parametrize("compressor", ["auto", "noop", "fft", "constant", "polynomial", "idw", "rle"])
def test_integration_dir_and_file(A, atsc_bin, golden_dir, tmp_path, compressor):
    # integration_test.rs:59-94: a directory with 1.wbro, 2.wbro and a single file
    d = tmp_path / "dir"
    d.mkdir()
    src = os.path.join(golden_dir, "wbros", "memory_used.wbro")
    shutil.copy(src, d / "1.wbro")
    shutil.copy(src, d / "2.wbro")
    _run(atsc_bin, "--compressor", compressor, d)
    assert (d / "1.bro").exists() and (d / "2.bro").exists()
    shutil.copy(src, tmp_path / "3.wbro")
    _run(atsc_bin, "--compressor", compressor, tmp_path / "3.wbro")
    assert (tmp_path / "3.bro").exists()
    assert (d / "1.bro").read_bytes() == (tmp_path / "3.bro").read_bytes()


@pytest.mark.parametrize("level", range(7))
def test_integration_sample_levels(A, atsc_bin, golden_dir, tmp_path, level):
    # integration_test.rs:96-106
    shutil.copy(os.path.join(golden_dir, "wbros", "go_gc_heap_goal_bytes.wbro"), tmp_path / "1.wbro")
    _run(atsc_bin, "--compressor", "auto", "-c", level, tmp_path / "1.wbro")
    assert (tmp_path / "1.bro").exists()


def test_cli_rejects_bad_flags(atsc_bin, tmp_path):
    for args in (["--compressor", "lz4", "x"], ["-e", "51", "x"], ["-c", "7", "x"], []):
        r = subprocess.run([atsc_bin] + args, capture_output=True, text=True, timeout=60)
        assert r.returncode == 2
    r = subprocess.run([atsc_bin, str(tmp_path / "missing.wbro")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1  # main.rs:239-242


# ---- csv-compressor/src/main.rs --------------------------------------------
@pytest.fixture(scope="module")
def csvc_bin(A):
    p = os.path.join(os.path.dirname(A.__file__), "bin", "csv-compressor")
    assert os.path.exists(p), "csv-compressor CLI was not built"
    return p


def _metric_csv(path, n=3000, klass=0):
    """`timestamp,value` file: millisecond timestamps every 15 s from 00:00:30 UTC with one gap."""
    from oracle import vsri_oracle as VO

    vals = H.synth_series(321, n, klass=klass)
    day = 1700006400
    ts, t = [], (day + 30) * 1000
    for i in range(n):
        ts.append(t + (i * 37) % 1000)
        t += 15000 if i != n // 2 else 15000 * 20
    with open(path, "w") as f:
        f.write(VO.samples_to_csv_text(ts, vals))
    return np.array(ts, dtype=np.int64), vals


@pytest.mark.parametrize("compressor,error", [("noop", 5), ("constant", 5), ("auto", 0), ("auto", 5),
                                              ("fft", 5), ("polynomial", 3), ("idw", 3)])
def test_csv_compressor_round_trip(A, ctx, oracle, csvc_bin, tmp_path, compressor, error):
    from oracle import vsri_oracle as VO

    src = tmp_path / "metric.csv"
    ts, vals = _metric_csv(src)
    _run(csvc_bin, "--output-vsri", "--output-wavbrro", "--compressor", compressor, "-e", error, src)
    # the three outputs of the compress leg (main.rs:174-207)
    index = VO.metric_from_samples(ts)
    assert (tmp_path / "metric.vsri").read_text() == index.to_text()
    assert (tmp_path / "metric.wavbro").read_bytes() == A.wbro_to_bytes(vals)
    bro = (tmp_path / "metric.bro").read_bytes()
    cid = getattr(A, compressor.upper())
    ref = oracle.compress_data(vals, cid, cli_error=error)
    fg, fo = H.parse_bro(bro), H.parse_bro(ref)
    assert fg[0] == fo[0] and [f[:3] for f in fg[1]] == [f[:3] for f in fo[1]]
    if all(f[2] != oracle.FFT for f in fg[1]):
        assert bro == ref
    # -u: .bro + .vsri -> .wbro + .csv (main.rs:139-173)
    out = tmp_path / "restored"
    _run(csvc_bin, "-u", "-o", out, tmp_path / "metric.bro")
    dec = A.wbro_read(tmp_path / "restored.wbro")
    assert len(dec) == len(vals)
    if compressor == "noop":
        assert np.array_equal(dec, np.sign(vals) * np.floor(np.abs(vals) + 0.5))
    elif error == 0:
        # "lossless" keeps every point, and the decoder still rounds to 5 decimals (utils/mod.rs:61-74)
        assert np.max(np.abs(dec - vals)) <= 5.0001e-6
        assert np.array_equal(dec, oracle.decompress_data(bro))
    elif compressor != "constant":
        assert H.mape(vals, dec) <= error / 100.0
    assert (tmp_path / "restored.csv").read_text() == VO.samples_to_csv_text(
        VO.metric_sample_times(index, len(vals)), dec)


def test_csv_compressor_flags_and_failures(A, csvc_bin, tmp_path):
    src = tmp_path / "m.csv"
    _metric_csv(src, n=300)
    # --no-compression writes only what was asked for
    _run(csvc_bin, "--no-compression", "--output-vsri", "-o", tmp_path / "idx.any", src)
    assert (tmp_path / "idx.vsri").exists() and not (tmp_path / "idx.bro").exists()
    # clap rejections: exit status 2
    for args in (["--compressor", "rle", str(src)], ["-e", "51", str(src)], ["-c", "7", str(src)], []):
        r = subprocess.run([csvc_bin] + args, capture_output=True, text=True, timeout=60)
        assert r.returncode == 2, args
    # panics of the reference: exit status 101
    r = subprocess.run([csvc_bin, str(tmp_path / "missing.csv")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 101
    r = subprocess.run([csvc_bin, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 101  # "Input is not a file"
    bad = tmp_path / "bad.csv"
    bad.write_text("timestamp,value\n1000,1.0\nabc,2.0\n")
    r = subprocess.run([csvc_bin, str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 101
    back = tmp_path / "back.csv"  # the second sample is earlier in the day than the first
    back.write_text("timestamp,value\n1700006500000,1.0\n1700006400000,2.0\n")
    r = subprocess.run([csvc_bin, str(back)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 101
    # -u without the .vsri next to the .bro: "failed to read vsri"
    _run(csvc_bin, "--compressor", "noop", src)
    r = subprocess.run([csvc_bin, "-u", str(tmp_path / "m.bro")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 101
    # -u on something that is not a BRO file does nothing (bro_reader.rs:31-46)
    r = subprocess.run([csvc_bin, "-u", str(src)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and not (tmp_path / "m.wbro").exists()
