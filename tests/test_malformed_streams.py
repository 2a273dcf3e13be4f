"""Untrusted bytes.  The reference panics on malformed input (bincode decode `.unwrap()`, data.rs:98;
header.rs:34-37,72-74; `step_by(0)`, polynomial.rs:329-340); the library must answer ATSC_E_FORMAT (or
another negative code) and keep working.

CPU part (no GPU): the host parsers -- BRO record walk (atsc_bro_scan, the same walk atsc_dplan_create and
atsc_stream_from_bytes use), WBRO, VSRI, CSV -- on mutated inputs, once through the shipped library and once
in an AddressSanitizer + UBSan build of the host sources (tests/asan, g++), which also drives the host half
of atsc_dplan_create (atsc_internal_dplan_parse).
GPU part: mutated streams of every codec through all three decoder tiers; the context must decode a
good stream afterwards."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ME5 = float(np.float32(5) / np.float32(100))


@pytest.fixture(scope="module")
def A():
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    return atsc_amd


def _scan(A, b):
    import ctypes as C

    arr = np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)
    nf, ns = C.c_uint64(), C.c_uint64()
    rc = A.capi.lib().atsc_bro_scan(arr.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), C.byref(nf), C.byref(ns))
    return rc, nf.value, ns.value


def _streams(oracle, lengths=(12, 64, 256, 1000)):
    """Valid .bro images from the oracle: every codec, several frame lengths, a few frames each."""
    out = {}
    for n in lengths:
        x = np.concatenate([H.synth_series(11, n, klass=k) for k in (0, 1, 2, 3)])
        off = H.frame_offsets(len(x), n)
        for name, comp, bounded in (("auto", oracle.AUTO, True), ("fft", oracle.FFT, True),
                                    ("poly", oracle.POLYNOMIAL, True), ("idw", oracle.IDW, True),
                                    ("rle", oracle.RLE, False), ("noop", oracle.NOOP, False),
                                    ("constant", oracle.CONSTANT, False)):
            bro, _, _ = oracle.stream_compress(x, off, comp, bounded, ME5, 0)
            out["%s_%d" % (name, n)] = bytes(bro)
    return out


def _varint(v):
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return bytes([251]) + struct.pack("<H", v)
    if v < 1 << 32:
        return bytes([252]) + struct.pack("<I", v)
    return bytes([253]) + struct.pack("<Q", v)


def test_scan_accepts_oracle_streams(A, oracle):
    for name, bro in _streams(oracle).items():
        rc, nf, ns = _scan(A, bro)
        n = int(name.split("_")[1])
        assert (rc, nf, ns) == (0, 4, 4 * n), name


def test_scan_rejects_wrapping_lengths(A):
    """The advisor's round-1 reproducer: a payload length of 2^64 - 20 wrapped `pos + len` past the bound
    check (std::length_error through the ABI); 2^64 - 12 was accepted as a bogus record."""
    head = b"BRRO" + struct.pack("<I", 1) + bytes([1]) + bytes([1]) + bytes([41]) + _varint(1024) + bytes([3])
    for evil in (2 ** 64 - 20, 2 ** 64 - 12, 2 ** 64 - 1, 2 ** 63, 2 ** 32, 2 ** 32 - 1, 1000):
        b = head + _varint(evil)
        b += bytes([30]) * max(0, 30 - len(b))
        rc, _, _ = _scan(A, b)
        assert rc == A.capi.E_FORMAT, hex(evil)
    # a frame count the bytes cannot hold, and one that runs off the end
    good = head + _varint(3) + bytes([30, 3, 1])
    assert _scan(A, good)[0] == 0
    for cnt in (2, 250, 2 ** 16, 2 ** 40, 2 ** 64 - 1):
        b = good[:9] + _varint(cnt) + good[10:]
        assert _scan(A, b)[0] == A.capi.E_FORMAT, cnt
    assert _scan(A, good[:9] + bytes([253, 1, 0]))[0] == A.capi.E_FORMAT


def test_scan_survives_mutations(A, oracle):
    rng = np.random.default_rng(5)
    codes = {0, A.capi.E_FORMAT, A.capi.E_VERSION}
    for name, bro in _streams(oracle, lengths=(12, 256)).items():
        for cut in range(len(bro)):
            rc, _, _ = _scan(A, bro[:cut])
            assert rc in codes and rc != 0, (name, cut)   # every strict prefix is short of a record
        for _ in range(300):
            m = bytearray(bro)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            assert _scan(A, bytes(m))[0] in codes, name


def test_wbro_survives_mutations(A, golden_dir):
    rng = np.random.default_rng(6)
    raw = open(os.path.join(golden_dir, "wbros", "uptime.wbro"), "rb").read()
    for cut in list(range(0, 64)) + list(range(len(raw) - 64, len(raw))):
        try:
            A.wbro_from_bytes(raw[:cut])
        except A.AtscError as e:
            assert e.rc == A.capi.E_FORMAT
    for _ in range(500):
        m = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(len(m) - 64, len(m)) if rng.integers(0, 2) else rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        try:
            A.wbro_from_bytes(bytes(m))
        except A.AtscError as e:
            assert e.rc == A.capi.E_FORMAT
    # 2 GiB archives cannot be expressed in rkyv's i32 offsets: the writer says so instead of wrapping
    import ctypes as C

    out, ln = C.POINTER(C.c_uint8)(), C.c_uint64()
    one = np.zeros(1)
    rc = A.capi.lib().atsc_wbro_to_bytes(one.ctypes.data_as(C.POINTER(C.c_double)), 2 ** 28, C.byref(out), C.byref(ln))
    assert rc == A.capi.E_UNSUPPORTED


def test_host_parsers_under_sanitizers(A, oracle, golden_dir, tmp_path):
    """ASan + UBSan build (g++) of atsc_host.cpp / atsc_stream.cpp / atsc_vsri.cpp; tests/asan/host_fuzz.cpp feeds
    the corpus written here through the parsers: as is, truncated, bit-flipped and with hostile varints."""
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no g++ / ROCm headers")
    asan = os.path.join(ROOT, "tests", "asan")
    subprocess.check_call(["make", "-s", "-j4", "-C", asan], stdout=subprocess.DEVNULL)
    corpus = tmp_path / "corpus"
    scratch = tmp_path / "scratch"
    corpus.mkdir()
    scratch.mkdir()
    for name, bro in _streams(oracle, lengths=(12, 256, 1000)).items():
        (corpus / (name + ".bro")).write_bytes(bro)
    # a frame of the large tier (its own plan-table entries: 2^a 3^b splits) and a stream of many frames
    x = H.synth_series(2, 16384 + 3 * 2048)
    bro, _, _ = oracle.stream_compress(x, np.array([0, 16384, 18432, 20480, 22528], dtype=np.uint64), oracle.AUTO, True, ME5, 0)
    (corpus / "large_16384.bro").write_bytes(bytes(bro))
    for f in ("uptime.wbro", "go_gc_heap_goal_bytes.wbro"):
        shutil.copy(os.path.join(golden_dir, "wbros", f), corpus / f)
    shutil.copy(os.path.join(golden_dir, "csv", "cpu_utilization.csv"), corpus / "cpu.csv")
    (corpus / "samples.csv").write_text("timestamp,value\n1700000000000,1.5\n1700000015000,2.5\n1700000030000,-3e5\n")
    (corpus / "index.vsri").write_text("55745\n59435\n15,0,55745,166\n30,166,58250,40\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([os.path.join(asan, "_build", "host_fuzz"), str(corpus), str(scratch)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-6000:])
    assert "parser calls" in p.stdout, p.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "host_fuzz.log"), "a") as f:
        f.write(p.stdout)


# ---------------------------------------------------------------------------------------------------------
# GPU: mutated streams through the decoders
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx(A):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    c = A.Context(0)
    yield c
    c.close()


def _decode_or_reject(A, ctx, records, has_count=False):
    try:
        return ctx.decompress_host(records, has_count)
    except A.AtscError as e:
        assert e.rc in (A.capi.E_FORMAT, A.capi.E_UNSUPPORTED), e
        return None


@pytest.mark.gpu
@pytest.mark.parametrize("n", [40, 256, 2048, 16384])   # direct-DFT, one-wavefront, multi-wavefront, large tier
def test_mutated_streams_through_the_decoders(A, ctx, oracle, n):
    rng = np.random.default_rng(100 + n)
    x = np.concatenate([H.synth_series(5, n, klass=k) for k in (0, 1, 2, 3)])
    off = H.frame_offsets(len(x), n)
    rounds = 150 if n <= 2048 else 40
    n_rejected = n_decoded = 0
    for name, comp, bounded in (("auto", oracle.AUTO, True), ("fft", oracle.FFT, True), ("poly", oracle.POLYNOMIAL, True),
                                ("idw", oracle.IDW, True), ("rle", oracle.RLE, False), ("noop", oracle.NOOP, False),
                                ("constant", oracle.CONSTANT, False)):
        if comp == oracle.IDW and n > 2048:
            continue  # (O(n^2) on the CPU oracle)
        bro, _, _ = oracle.stream_compress(x, off, comp, bounded, ME5, 0)
        bro = bytes(bro)
        body = bro[10:]  # records without header and count (4 frames: a one-byte count)
        good = ctx.decompress_host(body)
        frames = H.parse_bro_body(body, with_count=False)
        muts = []
        for _ in range(rounds):  # random stomps anywhere, and in the first bytes of a record (headers, counts)
            m = bytearray(body)
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, len(m))) if rng.integers(0, 2) else int(rng.integers(0, min(len(m), 24)))
                m[at] = int(rng.integers(0, 256)) if rng.integers(0, 2) else m[at] ^ (1 << int(rng.integers(0, 8)))
            muts.append(bytes(m))
        for cut in sorted(set(int(c) for c in rng.integers(0, len(body), size=20))):
            muts.append(body[:cut])
        # targeted: inflated / wrapped payload lengths, counts and sample counts in the first record;
        # polynomial step = 0 (step_by(0) panic in the reference), positions beyond the transform length
        fs, sc, tag, payload = frames[0]
        rest = body[len(_varint(fs)) + len(_varint(sc)) + len(_varint(tag)) + len(_varint(len(payload))) + len(payload):]
        for evil_len in (2 ** 64 - 20, 2 ** 64 - 12, 2 ** 32 - 12, len(payload) + 1, len(body) * 2, 0):
            muts.append(_varint(fs) + _varint(sc) + _varint(tag) + _varint(evil_len) + payload + rest)
        for evil_n in (0, 1, n + 1, 131072, 131073, 2 ** 32, 2 ** 64 - 1):
            muts.append(_varint(fs) + _varint(evil_n) + _varint(tag) + _varint(len(payload)) + payload + rest)
        if tag in (oracle.POLYNOMIAL, oracle.IDW):
            for step in (0, 1, 255):
                p2 = payload[:-1] + bytes([step])
                muts.append(_varint(fs) + _varint(sc) + _varint(tag) + _varint(len(p2)) + p2 + rest)
        if tag in (oracle.FFT, oracle.NOOP, oracle.RLE, oracle.POLYNOMIAL, oracle.IDW):
            at = 1 if tag in (oracle.FFT, oracle.NOOP) else 2  # the element count of the payload
            for cnt in (250, 65535, 2 ** 31, 2 ** 64 - 1):
                p2 = payload[:at] + _varint(cnt) + payload[at + 1:]
                muts.append(_varint(fs) + _varint(sc) + _varint(tag) + _varint(len(p2)) + p2 + rest)
        for m in muts:
            got = _decode_or_reject(A, ctx, m)
            if got is None:
                n_rejected += 1
            else:
                n_decoded += 1
        # the context is intact: the good stream still decodes to the same samples
        again = ctx.decompress_host(body)
        assert np.array_equal(again, good, equal_nan=True), name
    assert n_rejected > 0 and n_decoded > 0, (n_rejected, n_decoded)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity.log"), "a") as f:
        f.write("mutated streams n=%d: %d rejected, %d decoded (still well-formed)\n" % (n, n_rejected, n_decoded))
