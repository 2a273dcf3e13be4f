"""GPU parity tests: the HIP path (through the C ABI in include/atsc_hip.h) against the CPU
oracle on the same seeded inputs, the reference's fixtures, and -- at BASELINE.json's full
sizes -- size-independent properties (round trips, error bound, stream structure)."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests import parity as P

pytestmark = pytest.mark.gpu

ME5 = float(np.float32(5) / np.float32(100))
ME1 = float(np.float32(1) / np.float32(100))
ME3 = float(np.float32(3) / np.float32(100))


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    import __graft_entry__ as G

    G.build()
    import atsc_amd

    c = atsc_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def A():
    import atsc_amd

    return atsc_amd


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity.log", "a") as f:
        f.write(msg + "\n")


# ---------------------------------------------------------------------------------------
# auto selector on the synthetic classes of SURVEY 8(d), F256 framing (configs[1], configs[2])
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("klass", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("me", [ME5, ME1])
def test_auto_f256_classes(ctx, A, oracle, klass, me):
    nfr = 96
    x = H.synth_series(3, 256 * nfr, klass=klass)
    off = H.frame_offsets(len(x), 256)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, me)
    _log(P.assert_summary(s, nfr, "auto f256 class %d me %.3f codecs %s" % (klass, me, s["codecs"])))


def test_auto_mixed_block(ctx, A, oracle):
    # classes cycled per 1024-sample block so one launch sees every selector branch
    x = H.synth_series(11, 256 * 200, block=1024)
    off = H.frame_offsets(len(x), 256)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, ME5)
    assert len(s["codecs"]) >= 4, s["codecs"]
    _log(P.assert_summary(s, 200, "auto mixed codecs %s" % s["codecs"]))
    # compression ratio within +-5 % of the oracle on identical framing
    assert abs(s["bytes"] - s["oracle_bytes"]) <= 0.05 * s["oracle_bytes"]


@pytest.mark.parametrize("comp", ["FFT", "POLYNOMIAL", "IDW", "RLE", "CONSTANT", "NOOP"])
def test_forced_codecs(ctx, A, oracle, comp):
    cid = getattr(A, comp)
    bounded = comp in ("FFT", "POLYNOMIAL", "IDW")  # main.rs:150-162
    x = np.concatenate([H.synth_series(5, 256 * 12, klass=k) for k in range(5)])
    off = H.frame_offsets(len(x), 256)
    s = P.compare_batch(oracle, ctx, x, off, cid, bounded, ME5)
    _log(P.assert_summary(s, len(off) - 1, "forced %s" % comp))
    if comp != "FFT":
        assert s["tol"] == 0 and s["boundary"] == 0


# ---------------------------------------------------------------------------------------
# frame lengths: tails, primes, every kernel class
# ---------------------------------------------------------------------------------------
SIZES = [1, 2, 3, 5, 17, 64, 100, 127, 128, 129, 200, 255, 256, 300, 393, 511, 512, 513, 1000, 1024,
         1500, 2048, 3000, 4096]


@pytest.mark.parametrize("klass", [0, 2, 3])
def test_auto_frame_lengths(ctx, A, oracle, klass):
    xs, offs = [], [0]
    for n in SIZES:
        xs.append(H.synth_series(100 + n, n, klass=klass))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), A.AUTO, True, ME5)
    _log(P.assert_summary(s, len(SIZES), "auto lengths class %d codecs %s" % (klass, s["codecs"])))


def test_reference_chunker_on_fixture(ctx, A, oracle, golden_dir):
    # e2e.rs flow on go_gc_heap_goal_bytes.wbro: clean -> chunk (2048, 512, 393) -> auto
    d = H.read_wbro(os.path.join(golden_dir, "wbros", "go_gc_heap_goal_bytes.wbro"))
    d = A.clean_data(d)
    sizes = A.chunk_sizes(len(d))
    assert sizes == [2048, 512, 393]
    off = np.cumsum([0] + sizes).astype(np.uint64)
    for e in (0, 1, 3, 5):
        me = float(np.float32(e) / np.float32(100))
        s = P.compare_batch(oracle, ctx, d, off, A.AUTO, True, me)
        _log(P.assert_summary(s, 3, "fixture heap e=%d codecs %s" % (e, s["codecs"])))
        bro = A.bro_prefix(3) + s["records"]
        if s["boundary"] == 0 and s["tol"] == 0:
            assert bro == oracle.compress_data(d, oracle.AUTO, cli_error=e)
        out = ctx.decompress_host(s["records"])
        assert len(out) == len(d)
        if e == 0:
            assert np.array_equal(out, d)  # e2e.rs:158-160
        else:
            assert H.mape(d, out) <= me + 1e-3  # e2e.rs:234-248 (+ fft.rs:334 truncation)


def test_csv_constant_kat(ctx, A, oracle, golden_dir):
    # BASELINE.json configs[0]; hand-derived 58-byte stream of SURVEY 8(c)
    from tests.golden import kat as K

    vals = H.read_csv_values(os.path.join(golden_dir, "csv", "cpu_utilization.csv"))
    sizes = A.chunk_sizes(len(vals))
    off = np.cumsum([0] + sizes).astype(np.uint64)
    rec, _, chosen, _ = ctx.compress_host(vals, off, A.CONSTANT, False, 0.0, 0)
    assert (A.bro_prefix(len(sizes)) + rec).hex() == K.CSV_CONSTANT_BRO_HEX
    out = ctx.decompress_host(rec)
    assert out[0] == 13.85002983491348 and len(out) == 2854


# ---------------------------------------------------------------------------------------
# edge cases the reference's behaviour depends on (D7, D9, D10; App. C)
# ---------------------------------------------------------------------------------------
def _edge_frames():
    rng = np.random.default_rng(42)
    fr = []
    fr.append(np.zeros(256))                                   # all zero -> constant
    z = H.synth_series(1, 256, klass=0); z[17] = 0.0; fr.append(z)   # a zero poisons MAPE (D10)
    z = H.synth_series(1, 256, klass=0); z[0] = 0.0; fr.append(z)
    fr.append(-H.synth_series(2, 256, klass=0))                # negative F64
    fr.append(-np.floor(H.synth_series(2, 256, klass=2)))      # negative integers -> I16
    fr.append(np.floor(H.synth_series(2, 256, klass=2)) * 100)  # I32 range
    fr.append(np.floor(H.synth_series(2, 256, klass=2)) * 1e7)  # beyond i32 -> F64 bitdepth
    fr.append(np.where(np.arange(256) % 2 == 0, 5.0, 250.0))   # alternating two values: RLE R = n
    fr.append(np.repeat(rng.integers(0, 255, 16), 16).astype(np.float64))  # U8 runs
    fr.append(rng.integers(0, 255, 256).astype(np.float64))    # U8 noise
    fr.append(rng.normal(0, 1, 256))                           # sign changes, tiny magnitudes
    fr.append(np.full(256, 1.23456))                           # F64 constant
    a = np.full(256, 7.0); a[255] = 8.0; fr.append(a)          # step at the end
    a = np.full(256, 7.0); a[0] = 8.0; fr.append(a)
    a = np.full(256, 1e-3); a[100:] = 2e-3; fr.append(a)
    a = np.full(256, 1000.0); a[::2] += 1e-4; fr.append(a)    # min != max in f64, equal in f32
    a = np.arange(256, dtype=np.float64); a[0] = -0.0; fr.append(a)
    fr.append(np.arange(256, dtype=np.float64) + 0.5)
    fr.append(1e12 + np.arange(256, dtype=np.float64) * 1e6)
    fr.append(H.synth_series(9, 256, klass=1) * 1e-6)
    return fr


def test_edge_frames_auto(ctx, A, oracle):
    fr = _edge_frames()
    x = np.concatenate(fr)
    off = H.frame_offsets(len(x), 256)
    for me in (ME5, ME1, 0.0):
        s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, me)
        _log(P.assert_summary(s, len(fr), "edge frames me=%r codecs %s" % (me, s["codecs"])))


@pytest.mark.parametrize("comp", ["FFT", "POLYNOMIAL", "IDW", "RLE", "NOOP", "CONSTANT"])
def test_edge_frames_forced(ctx, A, oracle, comp):
    fr = _edge_frames()
    x = np.concatenate(fr)
    off = H.frame_offsets(len(x), 256)
    s = P.compare_batch(oracle, ctx, x, off, getattr(A, comp), comp in ("FFT", "POLYNOMIAL", "IDW"), ME5)
    _log(P.assert_summary(s, len(fr), "edge forced %s" % comp))


# ---------------------------------------------------------------------------------------
# decompression parity (configs[4] path)
# ---------------------------------------------------------------------------------------
def test_decompress_matches_oracle(ctx, A, oracle):
    x = np.concatenate([H.synth_series(21, 256 * 16, klass=k) for k in range(5)] + _edge_frames())
    off = H.frame_offsets(len(x), 256)
    nf = len(off) - 1
    for comp, bounded in ((A.AUTO, True), (A.FFT, True), (A.POLYNOMIAL, True), (A.IDW, True),
                          (A.RLE, False), (A.NOOP, False), (A.CONSTANT, False), (A.POLYNOMIAL, False),
                          (A.IDW, False)):
        # decode the ORACLE's stream on the GPU and compare with the oracle's own decode
        bro, chosen, _ = oracle.stream_compress(x, off, comp, bounded, ME5, 0)
        ref = oracle.decompress_data(bro)
        body_off, nfr = A.bro_open(bro)
        assert nfr == nf
        out = ctx.decompress_host(bro[body_off:])
        assert len(out) == len(ref)
        for i in range(nf):
            seg = slice(int(off[i]), int(off[i + 1]))
            if chosen[i] == oracle.FFT:
                # f32 inverse transform: within 4 f32 ulp of the frame's magnitude (+ 1e-5 grid)
                scale = max(np.max(np.abs(ref[seg])), 1e-30)
                assert np.max(np.abs(out[seg] - ref[seg])) <= 4 * scale * 2.0 ** -23 + 1.00001e-5, (
                    comp, i, np.max(np.abs(out[seg] - ref[seg])))
            else:
                assert np.array_equal(out[seg], ref[seg]), (comp, i, chosen[i])


def test_decompress_rejects_garbage(ctx, A):
    with pytest.raises(A.AtscError):
        ctx.decompress_host(bytes([41, 251, 0, 1, 1, 4, 15, 200, 0, 0]))
    with pytest.raises(A.AtscError):
        A.bro_open(b"XXXX" + bytes(10))
    with pytest.raises(A.AtscError):
        A.bro_open(b"BRRO" + bytes([9, 0, 0, 0, 1, 1]))


# ---------------------------------------------------------------------------------------
# BASELINE.json full sizes: properties that do not need the oracle
# ---------------------------------------------------------------------------------------
def _full_size_roundtrip(ctx, A, n_samples, me, seed):
    import torch

    x = H.synth_series(seed, n_samples)  # classes cycled per 65536-sample block
    off = H.frame_offsets(n_samples, 256)
    nf = len(off) - 1
    dev = torch.device("cuda:0")
    plan = ctx.plan(off)
    d_x = torch.from_numpy(x).to(dev)
    outs = plan.alloc_outputs(torch, dev)
    plan.compress(d_x, outs, A.AUTO, True, me, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec_off = outs["rec_off"].cpu().numpy().astype(np.int64)
    total = int(rec_off[-1])
    assert np.all(np.diff(rec_off) > 0) and total <= plan.body_bound
    body = outs["body"][:total].cpu().numpy().tobytes()
    chosen = outs["chosen"].cpu().numpy()
    err = outs["err"].cpu().numpy()
    # determinism: a second run produces the same bytes
    plan.compress(d_x, outs, A.AUTO, True, me, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert outs["body"][:total].cpu().numpy().tobytes() == body
    # device-resident decompress
    dp = A.DPlan(ctx, body)
    assert dp.n_frames == nf and dp.n_samples == n_samples
    d_body = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n_samples, dtype=torch.float64, device=dev)
    dp.decompress(d_body, d_out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    xo = x.reshape(nf, 256)
    oo = out.reshape(nf, 256)
    with np.errstate(divide="ignore", invalid="ignore"):
        fm = np.sum(np.abs((oo - xo) / xo), axis=1) / 256.0
    lossless = np.isin(chosen, [A.CONSTANT, A.RLE])
    assert np.array_equal(oo[lossless], xo[lossless])
    # every lossy frame honours the bound the selector applied (err <= max_error exactly) and the
    # decoded error equals the reported one up to f32 reconstruction noise
    lossy = ~lossless
    assert np.all(err[lossy] <= me)
    # Polynomial: the decoder repeats the encoder's arithmetic, so the decoded MAPE is the reported one
    poly = chosen == A.POLYNOMIAL
    assert np.all(np.abs(fm[poly] - err[poly]) <= 1e-9)
    # FFT: the reported error is over the Gibbs-padded signal of L = 288 samples (fft.rs:345); over
    # the 256 real samples it is at most L/n times that
    fft = chosen == A.FFT
    assert np.all(fm[fft] <= me * 288.0 / 256.0 + 2e-6)
    return {"ratio": 8.0 * n_samples / (total + 10), "codecs": {int(c): int(np.sum(chosen == c))
                                                                 for c in np.unique(chosen)}}


@pytest.mark.parametrize("klass", [1, 0])
def test_full_size_config1_1m(ctx, A, oracle, klass):
    """BASELINE.json configs[1] at full size: 2^20 samples in 4096 frames of 256, `--compressor fft`, e = 5 %,
    once all-C1 (deep ladder) and once all-C0.  Every frame goes through the
    oracle comparison (trips, K, bin order, coefficients, reported error); then the device-resident round
    trip: a forced codec is accepted whatever error it reached, so the decoded error is checked against
    the one the encoder reported."""
    import torch

    n = 1 << 20
    x = H.synth_series(0, n, klass=klass)
    off = H.frame_offsets(n, 256)
    nf = len(off) - 1
    s = P.compare_batch(oracle, ctx, x, off, A.FFT, True, ME5)
    _log(P.assert_summary(s, nf, "configs[1] 1M forced fft class %d" % klass))
    assert s["codecs"] == {A.FFT: nf}
    dev = torch.device("cuda:0")
    plan = ctx.plan(off)
    d_x = torch.from_numpy(x).to(dev)
    outs = plan.alloc_outputs(torch, dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.compress(d_x, outs, A.FFT, True, ME5, 0, st)
    torch.cuda.synchronize()
    total = int(outs["rec_off"][-1].item())
    body = outs["body"][:total].cpu().numpy().tobytes()
    assert body == s["records"]  # the device-resident call and the host-pointer call agree
    err = outs["err"].cpu().numpy()
    plan.compress(d_x, outs, A.FFT, True, ME5, 0, st)
    torch.cuda.synchronize()
    assert outs["body"][:total].cpu().numpy().tobytes() == body  # deterministic
    dp = A.DPlan(ctx, body)
    d_body = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().reshape(nf, 256)
    xo = x.reshape(nf, 256)
    fm = np.sum(np.abs((out - xo) / xo), axis=1) / 256.0
    # the reported error is the mean over the L = 288 Gibbs-padded samples (fft.rs:345): over the 256 real
    # samples the sum can only be smaller, the divisor is 256 instead of 288
    assert np.all(fm <= err * 288.0 / 256.0 + 2e-6)
    ks = [len(H.parse_fft_payload(f[3])[0]) for f in H.parse_bro_body(body, with_count=False)]
    assert max(ks) <= 25 and min(ks) >= 3  # mf = 3 ... 3 + 17 + 5 (fft.rs:298-302,348-352)
    _log("configs[1] class %d: ratio %.2f, K %d..%d, worst decoded MAPE %.4f" % (klass, 8.0 * n / (total + 12),
                                                                                min(ks), max(ks), float(fm.max())))


def test_full_size_config2_10m(ctx, A):
    r = _full_size_roundtrip(ctx, A, 10485760, ME5, seed=0)
    _log("config 10M roundtrip %r" % r)
    assert len(r["codecs"]) >= 4


def test_many_small_frames_two_level_pack(ctx, A, oracle):
    """131072 frames of 8 samples: the direct-DFT class (n < 128) and the record packer's two-level
    offset scan (more than 64 chunks of 1024 frames)."""
    nf, fl = 131072, 8
    x = H.synth_series(5, nf * fl, block=4096)
    off = H.frame_offsets(len(x), fl)
    rec, rec_off, chosen, err = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
    bro, cho, _ = oracle.stream_compress(x, off, oracle.AUTO, True, ME5, 0)
    assert np.array_equal(chosen, cho)
    body_off, n = A.bro_open(bro)
    assert n == nf
    ref = bro[body_off:]
    assert len(rec) == len(ref)
    fr_g = H.parse_bro_body(rec, with_count=False)
    fr_o = H.parse_bro_body(ref, with_count=False)
    verdicts = {}
    bad = []
    for i in range(nf):
        if fr_g[i] == fr_o[i]:
            v = "exact"
        else:  # every frame that is not byte-identical goes through the frame comparison (FFT tolerances)
            v = P.compare_frame(oracle, x[i * fl:(i + 1) * fl], ME5, fr_g[i][2], fr_g[i][3], fr_o[i][2], fr_o[i][3],
                                None, i, None)
        if v.startswith("FAIL"):
            bad.append((i, v))
        verdicts[v if not v.startswith("FAIL") else "fail"] = verdicts.get(v if not v.startswith("FAIL") else "fail", 0) + 1
    assert not bad, bad[:5]
    nfft = sum(1 for f in fr_g if f[2] == oracle.FFT)
    _log("small frames: %d frames, %d fft, verdicts %r" % (nf, nfft, verdicts))
    assert verdicts.get("tie", 0) <= max(2, int(P.TIE_FRAC * nf)), verdicts
    assert verdicts.get("boundary", 0) <= max(1, int(P.BOUNDARY_FRAC * nf)), verdicts
    assert np.all(np.diff(rec_off.astype(np.int64)) > 0)


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5, 6])
def test_sample_levels(ctx, A, oracle, level):
    """-c / --compression-selection-sample-level (frame/mod.rs:89-111): frames at least
    COMPRESSION_SPEED[level] long take the codec chosen on their prefix; shorter ones run the full
    selector.  Mixed lengths so both branches and the constant shortcut are hit."""
    sizes = [4096, 2048, 1024, 1000, 512, 300, 256, 200, 128, 100, 64]
    xs, offs = [], [0]
    for k, n in enumerate(sizes * 3):
        klass = (k * 7) % 5
        v = H.synth_series(300 + k, n, klass=klass)
        if k % 11 == 3:  # constant prefix, non-constant tail: the trial must not take the shortcut
            v[: n // 2] = v[0]
        xs.append(v)
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), A.AUTO, True, ME5, level=level)
    _log(P.assert_summary(s, len(sizes) * 3, "sample level %d codecs %s" % (level, s["codecs"])))


@pytest.mark.parametrize("comp", ["POLYNOMIAL", "IDW"])
def test_unbounded_compress(ctx, A, oracle, comp):
    """CompressedStream::compress_chunk_with (data.rs:47-53) -> Compressor::compress
    (compressor/mod.rs:63-74): polynomial()/idw store max(3, n/100) points, no error loop."""
    xs, offs = [], [0]
    for k, n in enumerate([12, 17, 100, 256, 300, 1024, 4096, 4, 256, 512]):
        v = H.synth_series(400 + k, n, klass=k % 5)
        xs.append(v)
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), getattr(A, comp), False, 0.0)
    _log(P.assert_summary(s, 10, "unbounded %s" % comp))
    assert s["exact"] == 10


def test_reference_kat_vectors_on_gpu(ctx, A):
    """The reference's own byte-level known-answer vectors, straight through the C ABI."""
    from tests.golden import kat as K

    def run(data, comp, bounded=False):
        rec, _, _, _ = ctx.compress_host(np.array(data), [0, len(data)], comp, bounded, 0.0, 0)
        (fs, sc, tag, payload), = H.parse_bro_body(rec, with_count=False)
        assert (fs, sc, tag) == (41, len(data), comp)
        return payload

    for name in ("CONSTANT_U8", "CONSTANT_F64"):
        d, want = getattr(K, name)
        assert run(d, A.CONSTANT) == bytes(want)
    assert run(K.NOOP[0], A.NOOP) == bytes(K.NOOP[1])
    for name in ("RLE_CONSTANT", "RLE_SIMPLE", "RLE_U8", "RLE_F64"):
        d, want = getattr(K, name)
        assert run(d, A.RLE) == bytes(want)
    for name in ("POLY_U8", "POLY_I16", "POLY_I32", "POLY_F64", "POLY_LINE"):
        d, want = getattr(K, name)
        assert run(d, A.POLYNOMIAL) == bytes(want)
    for name in ("IDW_U8", "IDW_LINE"):
        d, want = getattr(K, name)
        assert run(d, A.IDW) == bytes(want)
    # decoded values: polynomial.rs:486-514, :538-569
    for name, comp in (("POLY_CR_OUT", A.POLYNOMIAL), ("POLY_LINEAR_OUT", A.POLYNOMIAL),
                       ("IDW_OUT", A.IDW), ("IDW_LINEAR_OUT", A.IDW)):
        d, want = getattr(K, name)
        rec, _, _, _ = ctx.compress_host(np.array(d), [0, len(d)], comp, False, 0.0, 0)
        assert list(ctx.decompress_host(rec)) == want
    # data.rs:146-154 whole stream
    rec, _, _, _ = ctx.compress_host(np.ones(1024), [0, 1024], A.CONSTANT, False, 0.0, 0)
    assert A.bro_prefix(1) + rec == bytes(K.STREAM_CONSTANT_1024)


def test_unbounded_fft(ctx, A, oracle):
    """Compressor::compress(FFT) -> fft() (fft.rs:466-484, :366-388): no Gibbs padding, transform
    length = frame length (2^a 3^b through the Stockham stages, anything else through the direct DFT),
    the max(3, n/100) largest bins, no error loop."""
    from tests.golden import kat as K

    sizes = [12, 100, 127, 128, 256, 300, 1000, 1024, 2187, 4096, 4095, 3, 7]
    xs, offs = [np.array(K.V12)], [0, 12]
    for k, n in enumerate(sizes[1:]):
        xs.append(H.synth_series(500 + k, n, klass=(0, 1, 2)[k % 3]))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.FFT, False, 0.0)
    _log(P.assert_summary(s, len(sizes), "unbounded FFT"))
    # fft.rs:571-579: the decoded values of the 12-sample vector are pinned to 5 decimals
    out = ctx.decompress_host(s["records"])
    assert list(out[:12]) == K.FFT_LOSSY_OUT[1]
    ref = oracle.decompress_data(A.bro_prefix(len(sizes)) + s["records"])
    assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-5


# ---------------------------------------------------------------------------------------
# large frames: what the reference chunker emits for long series (optimizer/mod.rs:78-98)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("sizes,klasses", [
    ([8192, 5000, 6500, 16384, 4097], (0, 2, 3, 1)),   # 6500 -> L = 6561 = 3^8 (odd: full complex FFT)
    ([32768, 65536], (0, 3)),
    # every power-of-two chunk from 8192 samples up has M = 243 x 9 P: the grid path of atsc_large_fast.h, all classes
    ([8192, 16384, 32768, 65536], (0, 1, 2, 3, 4)),
    ([131072], (0,)),
    ([131072], (1,)),                                    # busy signal: bins >= 65536 are admitted (u16 wrap)
])
def test_auto_large_frames(ctx, A, oracle, sizes, klasses):
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in klasses:
            xs.append(H.synth_series(700 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), A.AUTO, True, ME5)
    _log(P.assert_summary(s, len(offs) - 1, "large frames %s classes %s codecs %s" % (sizes, klasses, s["codecs"])))


@pytest.mark.parametrize("me", [ME1, 0.0, ME3], ids=["e1", "e0", "e3"])
@pytest.mark.parametrize("sizes,klasses", [
    ([8192, 16384, 32768, 65536], (0, 1, 2, 3, 4)),     # the grid path's lengths, every class
    ([131072], (0, 1, 2, 3)),                            # configs[3]'s frame in the CLI's framing
    ([5000, 6500, 20000, 4097], (0, 1, 3)),              # no 243 x 9 P split: the general kernels
], ids=["pow2", "131072", "other"])
def test_auto_large_frames_error_bounds(ctx, A, oracle, sizes, klasses, me):
    """The auto selector on large frames at the error bounds round 3 never held against the oracle's ENCODER:
    e = 1 % (configs[3]: FFT ladders that go on past their first trip, polynomial trips woven in between,
    `poly_next_lb` pruning), e = 0 (every ladder runs to its end; Noop is never a candidate, so the selector's pick
    among failing candidates -- frame/mod.rs:113-147 -- is what is compared) and e = 3 %."""
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in klasses:
            xs.append(H.synth_series(1700 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), A.AUTO, True, me)
    _log(P.assert_summary(s, len(offs) - 1, "large frames %s classes %s me %.3f codecs %s" % (sizes, klasses, me, s["codecs"])))


def test_config3_in_chunker_framing_against_the_oracle(ctx, A, oracle):
    """BASELINE.json configs[3] as the `atsc` CLI frames it: series of 262144 samples, class = series % 5, each cut
    into two 131072-sample frames (optimizer/mod.rs:78-98), --compressor auto, e = 1 %.  Twelve series = 24 frames,
    every one against the oracle's encoder (codec, K, bin order, coefficients, reported error); then the stream is
    decoded on the GPU and held against the oracle's decode of the same bytes."""
    NS, PER, F = 12, 262144, 131072
    x = np.concatenate([H.synth_series(s, PER, klass=s % 5) for s in range(NS)])
    sizes = []
    for s in range(NS):
        sizes += A.chunk_sizes(PER)
    assert sizes == [F] * (2 * NS)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, ME1)
    _log(P.assert_summary(s, 2 * NS, "configs[3] chunker framing, e = 1 %%, codecs %s" % s["codecs"]))
    assert len(s["codecs"]) >= 3
    out = ctx.decompress_host(s["records"])
    ref = oracle.decompress_data(A.bro_prefix(2 * NS) + s["records"])
    for i in range(2 * NS):
        seg = slice(i * F, (i + 1) * F)
        if s["chosen"][i] == oracle.FFT:
            scale = max(np.max(np.abs(ref[seg])), 1e-30)
            tol = (4 + np.log2(F)) * scale * 2.0 ** -23 + 1.00001e-5
            assert np.max(np.abs(out[seg] - ref[seg])) <= tol, i
        else:
            assert np.array_equal(out[seg], ref[seg]), (i, s["chosen"][i])
        if s["chosen"][i] in (oracle.CONSTANT, oracle.RLE):
            assert np.array_equal(out[seg], x[seg]), i


def _run_frames(seed):
    """Frames of the five power-of-two chunk lengths made of runs (rle.rs:142-189): few to thousands of runs, integer and
    fractional values, negative values, run starts on both sides of 251 and of 65536 (1-, 3- and 5-byte index varints),
    one value repeated in far-apart runs, two distinct values only, a single run change at the very end."""
    rng = np.random.default_rng(seed)
    frames = []
    for n in (8192, 16384, 32768, 65536, 131072):
        for kind in range(6):
            if kind == 0:    # gauge: ~n / 136 runs of 41 integer values
                lens = rng.integers(16, 256, size=n // 16)
                vals = 100.0 + rng.integers(0, 41, size=len(lens))
            elif kind == 1:  # long runs, fractional values (F64 points)
                lens = rng.integers(200, 3000, size=n // 200 + 2)
                vals = np.round(rng.normal(50.0, 20.0, size=len(lens)), 3)
            elif kind == 2:  # two values alternating in short and long runs
                lens = rng.integers(1, 400, size=n // 8)
                vals = np.where(np.arange(len(lens)) % 2 == 0, 7.0, -3.0)
            elif kind == 3:  # negative integers beyond i16: I32 points
                lens = rng.integers(30, 600, size=n // 30 + 2)
                vals = -40000.0 - rng.integers(0, 9, size=len(lens)) * 1000.0
            elif kind == 4:  # one change in the last sample
                lens = np.array([n - 1, 1])
                vals = np.array([5.0, 6.0])
            else:            # many runs: close to what the grid path sorts in LDS, and beyond it for the longest frames
                lens = rng.integers(20, 60, size=n // 20 + 2)
                vals = 1000.0 + rng.integers(0, 300, size=len(lens))
            x = np.repeat(vals, lens)[:n]
            assert len(x) == n
            frames.append(x.astype(np.float64))
    return frames


@pytest.mark.parametrize("me", [ME1, 0.0, ME5], ids=["e1", "e0", "e5"])
def test_rle_frames_on_the_large_grid_path(ctx, A, oracle, me):
    """Run-structured frames of 8192 ... 131072 samples under the auto selector: k_large_decide1 sizes RLE exactly from
    the run-start bit map of the polynomial pieces and, where it wins, emits it -- byte-identical to the oracle -- and
    the decoder's grid path (k_large_dparse + k_large_trip243<true>, status 4) expands the GPU's and the oracle's RLE
    streams to the same samples.  Frames whose runs outgrow the LDS sort take the general kernels: same bars."""
    frames = _run_frames(5)
    x = np.concatenate(frames)
    off = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    nf = len(frames)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, me)
    _log(P.assert_summary(s, nf, "run frames on the large grid path me %.3f codecs %s" % (me, s["codecs"])))
    assert s["codecs"].get(A.RLE, 0) >= nf // 3, s["codecs"]
    out = ctx.decompress_host(s["records"])
    ref = oracle.decompress_data(A.bro_prefix(nf) + s["records"])
    for i in range(nf):
        seg = slice(int(off[i]), int(off[i + 1]))
        if s["chosen"][i] in (A.RLE, A.CONSTANT, A.POLYNOMIAL):
            assert np.array_equal(out[seg], ref[seg]), (i, s["chosen"][i])
        if s["chosen"][i] == A.RLE:
            assert np.array_equal(out[seg], x[seg]), i
    # the oracle's forced-RLE stream of the same frames through the GPU decoder
    bro, chosen, _ = oracle.stream_compress(x, off, A.RLE, False, 0.0, 0)
    body_off, nfr = A.bro_open(bro)
    assert nfr == nf
    assert np.array_equal(ctx.decompress_host(bro[body_off:]), x)
    # and the GPU's forced-RLE stream (general kernel) against the oracle's bytes
    rec, _, _, _ = ctx.compress_host(x, off, A.RLE, False, 0.0, 0)
    assert rec == bro[body_off:]


def test_hand_built_rle_streams_on_the_large_decoder(ctx, A, oracle):
    """RLE payloads (rle.rs:204-236) no encoder writes, on the chunker's power-of-two chunks, decoded like the oracle
    decodes them: two groups naming one start (an empty run: the later group wins -- the grid decoder ranks runs by a
    bit map of the starts and falls back to its sort when a start repeats), a group's starts in descending order and a
    small start in the three-byte form (the group walk's marker scans do not apply: the general decoder's), groups
    of one start and of hundreds, values of every width."""
    import struct
    rng = np.random.default_rng(4242)

    def vi(v):
        return P.H_varint(int(v))

    def val(bd, v):
        if bd == oracle.BD_U8:
            return bytes([int(v)])
        if bd == oracle.BD_F64:
            return struct.pack("<d", float(v))
        z = (int(v) << 1) ^ (int(v) >> 63)
        return vi(z)

    id_byte = oracle.compress(A.RLE, np.repeat([1.0, 2.0], [5, 5]), False, 0.0)[0][:1]
    recs, want = [], []
    for n in (8192, 131072):
        for case in ("empty", "descending", "wide-small", "plain"):
            for bd, pool in ((oracle.BD_U8, np.arange(0, 200.0)), (oracle.BD_I16, np.arange(-300.0, 300.0, 7)),
                             (oracle.BD_I32, np.arange(-90000.0, 90000.0, 4001)), (oracle.BD_F64, np.arange(0.5, 90.0, 1.25))):
                ng = int(rng.integers(2, 40))
                vals = rng.choice(pool, size=ng, replace=False)
                nruns = int(rng.integers(ng, 900))
                starts = np.unique(np.concatenate([[0], rng.integers(1, n, size=nruns - 1),
                                                   rng.integers(1, 251, size=3), rng.integers(251, min(n, 65536), size=3)]))
                owner = rng.integers(0, ng, size=len(starts))
                owner[:ng] = np.arange(ng)  # every group holds a run
                groups = [sorted(starts[owner == g].tolist()) for g in range(ng)]
                if case == "empty":  # the later groups repeat some starts of the earlier ones
                    for g in range(1, ng):
                        groups[g] = sorted(set(groups[g]) | set(rng.choice(groups[g - 1], size=min(2, len(groups[g - 1])),
                                                                           replace=False).tolist()))
                body = b""
                for g in range(ng):
                    st = groups[g][::-1] if case == "descending" else groups[g]
                    body += val(bd, vals[g]) + vi(len(st))
                    for k, i0 in enumerate(st):
                        if case == "wide-small" and g == 1 and k == 0 and i0 < 65536:
                            body += bytes([251]) + struct.pack("<H", i0)
                        else:
                            body += vi(i0)
                pay = id_byte + vi(bd) + vi(ng) + body
                recs.append(_record(A.RLE, n, pay))
                want.append(np.array(oracle.decompress(A.RLE, pay, n)))
    out = ctx.decompress_host(b"".join(recs))
    ref = np.concatenate(want)
    assert len(out) == len(ref)
    pos = 0
    for k, w in enumerate(want):
        assert np.array_equal(out[pos:pos + len(w)], w), k
        pos += len(w)


@pytest.mark.parametrize("comp,me", [("FFT", ME5), ("FFT", ME1), ("POLYNOMIAL", ME5), ("POLYNOMIAL", ME1)],
                         ids=["fft-e5", "fft-e1", "poly-e5", "poly-e1"])
def test_forced_fft_and_polynomial_on_the_large_grid_path(ctx, A, oracle, comp, me):
    """`--compressor fft` / `--compressor polynomial` (main.rs:150-162) on the chunker's power-of-two chunks: the grid
    path takes the ladders' first trips (k_large_decide1 / k_large_trip243 / k_large_decide2 with FastState::forced),
    a ladder that goes on is the general kernel's -- either way the oracle's trips, K, bin order and bytes."""
    sizes = [8192, 16384, 32768, 65536, 131072]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in (0, 1, 2, 3):
            xs.append(H.synth_series(1900 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, getattr(A, comp), True, me)
    _log(P.assert_summary(s, len(offs) - 1, "forced %s on power-of-two large frames, me %.3f" % (comp, me)))
    out = ctx.decompress_host(s["records"])
    ref = oracle.decompress_data(A.bro_prefix(len(offs) - 1) + s["records"])
    for i in range(len(offs) - 1):
        seg = slice(int(off[i]), int(off[i + 1]))
        if comp == "POLYNOMIAL":
            assert np.array_equal(out[seg], ref[seg]), i
        else:
            n = int(off[i + 1] - off[i])
            scale = max(np.max(np.abs(ref[seg])), 1e-30)
            assert np.max(np.abs(out[seg] - ref[seg])) <= (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5, i


def _config3_device_series(torch, dev, n_series, per):
    """configs[3] on the device: series s has class s % 5; classes 0-2 are generated by torch on the
    GPU, the gauge and constant classes on the host."""
    d_x = torch.empty(n_series * per, dtype=torch.float64, device=dev)
    for s in range(n_series):
        d_x[s * per:(s + 1) * per] = H.synth_series_torch(torch, dev, s, per, s % 5)
    return d_x


@pytest.mark.parametrize("framing", ["f256", "chunker"])
def test_full_size_config3_config4_one_gpu(ctx, A, framing):
    """BASELINE.json configs[3] and configs[4] at FULL size on one GPU: 4096 series x 262144 samples = 2^30 samples
    (8.6 GB of f64), auto, e = 1 %, once in 256-sample frames (bench.py's framing: 4 194 304 frames) and once in the
    reference chunker's 131072-sample frames (8192 frames); then the whole stream is decoded (configs[4]).  No oracle
    at this size: determinism, record offsets, the codec mix, bit-exact round trip of the lossless codecs, every lossy
    frame within the bound the selector applied, decoded error against the reported one.  Everything stays on the
    device except the record bytes the decode plan is built from."""
    import torch

    dev = torch.device("cuda:0")
    NS, PER = 4096, 262144
    n = NS * PER
    F = 256 if framing == "f256" else 131072
    L_over_n = 288.0 / 256.0 if F == 256 else 139968.0 / 131072.0
    d_x = _config3_device_series(torch, dev, NS, PER)
    off = np.arange(0, n + 1, F, dtype=np.uint64)
    nf = len(off) - 1
    plan = ctx.plan(off)
    st = torch.cuda.current_stream().cuda_stream
    outs = plan.alloc_outputs(torch, dev)
    plan.compress(d_x, outs, A.AUTO, True, ME1, 0, st)
    torch.cuda.synchronize()
    rec_off = outs["rec_off"]
    total = int(rec_off[-1].item())
    assert total <= plan.body_bound
    assert bool((rec_off[1:] > rec_off[:-1]).all())
    body1 = outs["body"][:total].clone()
    chosen = outs["chosen"].clone()
    err = outs["err"].clone()
    plan.compress(d_x, outs, A.AUTO, True, ME1, 0, st)  # determinism
    torch.cuda.synchronize()
    assert int(outs["rec_off"][-1].item()) == total and torch.equal(outs["body"][:total], body1)
    assert torch.equal(outs["chosen"], chosen)
    del outs
    plan.close()
    # a series' class decides its codec family: constant series -> Constant, the gauge -> RLE (or a lossy codec that
    # is smaller within 1 %), the others lossy
    ch = chosen.reshape(NS, nf // NS)
    cls = torch.arange(NS, device=dev) % 5
    assert bool((ch[cls == 4] == A.CONSTANT).all())
    assert not bool((ch[cls <= 2] == A.CONSTANT).any())  # (a 256-sample frame of the gauge may sit inside equal runs)
    assert bool(((ch[cls == 3] == A.CONSTANT) | (ch[cls == 3] == A.RLE)).float().mean() > 0.9)
    assert not bool((chosen == A.NOOP).any())
    # configs[4]: decode the whole stream
    body = body1.cpu().numpy()
    dp = A.DPlan(ctx, body)
    assert dp.n_frames == nf and dp.n_samples == n
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    dp.decompress(body1, d_out, st)
    torch.cuda.synchronize()
    dp.close()
    xo = d_x.reshape(nf, F)
    oo = d_out.reshape(nf, F)
    lossless = (chosen == A.CONSTANT) | (chosen == A.RLE)
    assert bool((oo[lossless] == xo[lossless]).all())
    assert bool((err[lossless] == 0).all())
    lossy = ~lossless
    assert bool((err[lossy] <= ME1).all())
    fm = ((oo - xo).abs() / xo.abs()).sum(dim=1) / float(F)  # (no zero samples in these classes)
    # (a ladder that ends with every sample stored reports 0.0, polynomial.rs:264-269, while the decoder still rounds
    # to five decimals: 5e-6 / |x| per sample at most)
    poly = (chosen == A.POLYNOMIAL) & (err > 0)
    assert bool(((fm[poly] - err[poly]).abs() <= 1e-9).all())
    poly0 = (chosen == A.POLYNOMIAL) & (err == 0)
    assert bool((fm[poly0] <= 1e-7).all())
    fft = chosen == A.FFT
    # the reported error is the mean over the L Gibbs-padded samples (fft.rs:345): over the n real samples the sum
    # can only be smaller, the divisor is n instead of L
    assert bool((fm[fft] <= ME1 * L_over_n + 2e-6).all())
    codecs = {int(c): int((chosen == c).sum().item()) for c in torch.unique(chosen).tolist()}
    _log("configs[3]/[4] full size on one GPU, %s framing: %d frames, ratio %.3f, codecs %s, worst lossy MAPE %.5f"
         % (framing, nf, 8.0 * n / (total + 12), codecs, float(fm[lossy].max().item())))
    del d_x, d_out, xo, oo


def test_decode_in_parts_with_records_longer_than_a_part(ctx, A):
    """atsc_decompress_frames into registered memory walks the records in parts; a record longer than a part's stride
    (131072-sample Noop / RLE frames: ~0.2-1.2 MB each) ends behind the next part's limit, and that part is then
    empty -- not an error.  Same samples as the pageable (one-plan) path."""
    import ctypes as C

    from atsc_amd import capi

    lib = capi.lib()
    F = 131072
    for comp, klass, nfr in ((A.NOOP, 0, 3), (A.RLE, 3, 5), (A.FFT, 1, 4)):
        x = np.concatenate([H.synth_series(60 + k, F, klass=klass) for k in range(nfr)])
        off = H.frame_offsets(len(x), F)
        rec, _, _, _ = ctx.compress_host(x, off, comp, comp == A.FFT, 0.0, 0)
        if len(rec) < (1 << 20):
            # (the parts form starts at 1 MB of records: pad the stream with more frames of the same kind)
            reps = (1 << 20) // len(rec) + 1
            x = np.tile(x, reps)
            off = H.frame_offsets(len(x), F)
            rec, _, _, _ = ctx.compress_host(x, off, comp, comp == A.FFT, 0.0, 0)
        assert len(rec) >= (1 << 20)
        ref = ctx.decompress_host(rec)  # pageable destination: one plan
        dec = np.full(len(x), -1.0)
        recarr = np.frombuffer(rec, dtype=np.uint8).copy()
        done = []
        try:
            for a in (dec, recarr):
                capi.check(lib.atsc_host_register(C.c_void_p(a.ctypes.data), a.nbytes), ctx._h)
                done.append(a)
            on = C.c_uint64()
            rc = lib.atsc_decompress_frames(ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(rec), 0,
                                            dec.ctypes.data_as(C.POINTER(C.c_double)), len(dec), C.byref(on))
            capi.check(rc, ctx._h)
            assert on.value == len(x)
            assert np.array_equal(dec, ref), comp
        finally:
            for a in done:
                lib.atsc_host_unregister(C.c_void_p(a.ctypes.data))


def test_host_calls_on_registered_memory_and_after_a_trim(ctx, A):
    """atsc_host_register page-locks the caller's buffers: atsc_compress_frames then uploads by DMA and copies the
    records back part by part beside the later uploads, atsc_decompress_frames enqueues the records' copy before the
    host walks them -- the bytes and the decoded samples must be those of the same calls on pageable memory.
    atsc_ctx_trim / atsc_release_caches give the retained memory back: the next calls rebuild what they need and
    return the same bytes."""
    import ctypes as C

    from atsc_amd import capi

    lib = capi.lib()
    n = 1 << 23  # 32768 frames of 256: four parts in atsc_compress_frames
    x = np.ascontiguousarray(H.synth_series(77, n))
    off = H.frame_offsets(n, 256)
    rec0, ro0, ch0, err0 = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
    out0 = ctx.decompress_host(rec0)
    offc = np.ascontiguousarray(off, dtype=np.uint64)
    nf = len(off) - 1
    body = np.empty(len(rec0) + 4096, dtype=np.uint8)
    dec = np.empty(n, dtype=np.float64)
    recarr = np.frombuffer(rec0, dtype=np.uint8).copy()
    bufs = [x, body, dec, recarr]
    done = []
    try:
        for a in bufs:
            capi.check(lib.atsc_host_register(C.c_void_p(a.ctypes.data), a.nbytes), ctx._h)
            done.append(a)
        for round_ in range(3):
            blen = C.c_uint64()
            capi.check(lib.atsc_compress_frames(
                ctx._h, x.ctypes.data_as(C.POINTER(C.c_double)), offc.ctypes.data_as(C.POINTER(C.c_uint64)), nf, A.AUTO, 1,
                C.c_float(np.float32(ME5)), 0, body.ctypes.data_as(C.POINTER(C.c_uint8)), body.size, C.byref(blen),
                None, None, None), ctx._h)
            assert bytes(body[: blen.value]) == rec0, round_
            on = C.c_uint64()
            capi.check(lib.atsc_decompress_frames(
                ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(rec0), 0,
                dec.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(on)), ctx._h)
            assert on.value == n and np.array_equal(dec, out0), round_
            if round_ == 0:
                capi.check(lib.atsc_ctx_trim(ctx._h), ctx._h)
            elif round_ == 1:
                lib.atsc_release_caches()
        # a capacity one byte short is reported, not overrun (the part-by-part copy checks every part's end)
        blen = C.c_uint64()
        rc = lib.atsc_compress_frames(
            ctx._h, x.ctypes.data_as(C.POINTER(C.c_double)), offc.ctypes.data_as(C.POINTER(C.c_uint64)), nf, A.AUTO, 1,
            C.c_float(np.float32(ME5)), 0, body.ctypes.data_as(C.POINTER(C.c_uint8)), len(rec0) - 1, C.byref(blen),
            None, None, None)
        assert rc == capi.E_CAPACITY, rc
    finally:
        for a in done:
            lib.atsc_host_unregister(C.c_void_p(a.ctypes.data))


def test_decode_in_parts_reports_a_bad_record_in_a_later_part(ctx, A):
    """atsc_decompress_frames into registered memory works in parts (the next part is walked and decoded while the one
    before travels back).  A record that is broken in a later part -- an unknown codec tag, a length that runs past the
    end, a payload the kernel rejects -- must end the call with the error the one-plan form gives, with every stream
    drained: the same context then decodes the intact stream again, into the same buffers, to the same samples."""
    import ctypes as C

    from atsc_amd import capi

    lib = capi.lib()
    n = 1 << 22
    x = np.ascontiguousarray(H.synth_series(91, n))
    off = H.frame_offsets(n, 256)
    rec0, ro0, _, _ = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
    assert len(rec0) > (1 << 20)
    out0 = ctx.decompress_host(rec0)
    ro = np.asarray(ro0, dtype=np.uint64)
    dec = np.empty(n, dtype=np.float64)
    recarr = np.frombuffer(rec0, dtype=np.uint8).copy()
    done = []
    try:
        for a in (dec, recarr):
            capi.check(lib.atsc_host_register(C.c_void_p(a.ctypes.data), a.nbytes), ctx._h)
            done.append(a)

        def call(length):
            on = C.c_uint64(12345)
            rc = lib.atsc_decompress_frames(ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), length, 0,
                                            dec.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(on))
            return rc, on.value

        nf = len(ro) - 1
        p_late = int(ro[int(nf * 0.9)])  # a record in the last part
        saved = recarr.copy()
        for mutate in ("length", "truncate"):
            recarr[:] = saved
            if mutate == "length":
                recarr[p_late] = 0xFF; recarr[p_late + 1] = 0xFF; recarr[p_late + 2] = 0xFF; recarr[p_late + 3] = 0x7F
                rc, onv = call(len(rec0))
            else:
                rc, onv = call(len(rec0) - 3)  # the last record is cut short
            assert rc != 0 and onv == 0, (mutate, rc, onv)  # (no result: *out_n = 0 even when parts were copied)
            recarr[:] = saved
            dec[:] = -1.0
            rc, onv = call(len(rec0))
            assert rc == 0 and onv == n and np.array_equal(dec, out0), mutate
    finally:
        for a in done:
            lib.atsc_host_unregister(C.c_void_p(a.ctypes.data))


def test_large_grid_path_keeps_the_frames_it_can_decide():
    """The grid path hands a frame back to the general kernel without a trace in the output -- the bytes are the same, the
    call is 100 us to 2 ms slower -- so the hand-backs are counted here: under ATSC_DEBUG_STOP=-3 k_compress_large<0>
    prints a reason code per frame it is left with (tools/fast_left_probe.py).  On the mixed workload at e = 5 % nothing
    may come back for a reason other than 7 (a passing polynomial the RLE lower bound could still beat) or 11-14 (a ladder
    that goes on), and codes 8-10 -- a select that misses its own count -- never."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for flen, nf in ((131072, 96), (16384, 256)):
        env = dict(os.environ, NF=str(nf), FLEN=str(flen), RAW="1")
        env.pop("ATSC_LARGE_NO_FAST", None)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fast_left_probe.py")], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-400:]
        head = [l for l in r.stdout.splitlines() if l.startswith("frames ")]
        assert head and int(head[0].split()[1]) == nf, r.stdout[-400:]
        whys = [int(l.split()[4]) for l in r.stdout.splitlines() if l.startswith("FASTLEFT")]
        assert all(w in (7, 11, 12, 13, 14) for w in whys), (flen, whys)
        assert len(whys) <= nf // 16, (flen, whys)


def test_forced_fft_large_frames_with_folded_positions(ctx, A, oracle):
    """Forced FFT on 131072-sample frames of a smooth class: the ladder stores a bin p and its `pos as u16` alias
    p + 65536 (fft.rs:242), so a stored position occurs twice.  Found by tools/fuzz_soak.py (FUZZ_LARGE=2): the
    comparison has to match entries by (position, occurrence) -- by position alone it held an entry against its alias."""
    F = 131072
    xs = [H.synth_series(900 + k, F, klass=c) for k, c in enumerate((2, 0, 2))]
    x = np.concatenate(xs)
    off = H.frame_offsets(len(x), F)
    s = P.compare_batch(oracle, ctx, x, off, A.FFT, True, ME5)
    dup = 0
    for fs, sc, tag, payload in H.parse_bro_body(s["records"], with_count=False):
        pos = [f[0] for f in H.parse_fft_payload(payload)[0]]
        dup += len(pos) - len(set(pos))
    assert dup >= 1, "the case this test is for: a folded position stored twice"
    _log(P.assert_summary(s, 3, "forced FFT, 131072-sample frames with folded positions (%d duplicates)" % dup))


@pytest.mark.parametrize("me", [ME5, ME1, 0.0, ME3], ids=["e5", "e1", "e0", "e3"])
def test_reference_chunker_long_series(ctx, A, oracle, me):
    """compress_data flow (main.rs:130-165) on a 300000-sample series: 131072, 131072, 32768,
    4096, 512, 480 -- every kernel tier in one batch, byte-compared at the stream level."""
    x = H.synth_series(42, 300000, block=50000)
    x = A.clean_data(x)
    sizes = A.chunk_sizes(len(x))
    assert sizes == [131072, 131072, 32768, 4096, 512, 480]
    off = np.cumsum([0] + sizes).astype(np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, me)
    _log(P.assert_summary(s, len(sizes), "chunker long series me %.3f codecs %s" % (me, s["codecs"])))


def test_decompress_large_frames(ctx, A, oracle):
    """Decode the ORACLE's stream of large frames on the GPU and compare with the oracle's decode."""
    sizes = [8192, 6500, 16384, 131072]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in ((0, 1, 2, 3) if n < 100000 else (0, 1)):
            xs.append(H.synth_series(800 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    nf = len(off) - 1
    for comp, bounded in ((A.AUTO, True), (A.FFT, True), (A.POLYNOMIAL, True), (A.RLE, False),
                          (A.NOOP, False), (A.CONSTANT, False)):
        bro, chosen, _ = oracle.stream_compress(x, off, comp, bounded, ME5, 0)
        ref = oracle.decompress_data(bro)
        body_off, nfr = A.bro_open(bro)
        out = ctx.decompress_host(bro[body_off:])
        assert len(out) == len(ref)
        for i in range(nf):
            seg = slice(int(off[i]), int(off[i + 1]))
            if chosen[i] == oracle.FFT:
                n = int(off[i + 1] - off[i])
                scale = max(np.max(np.abs(ref[seg])), 1e-30)
                tol = (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5  # f32 FFT of length ~n
                assert np.max(np.abs(out[seg] - ref[seg])) <= tol, (comp, i, np.max(np.abs(out[seg] - ref[seg])), tol)
            else:
                assert np.array_equal(out[seg], ref[seg]), (comp, i, chosen[i])


@pytest.mark.parametrize("me", [ME5, ME1])
def test_decompress_power_of_two_large_frames(ctx, A, oracle, me):
    """The ORACLE's stream of 8192 ... 131072-sample frames (all of them M = 243 x 9 P: k_large_dparse +
    k_large_trip243<true> per row dimension) decoded on the GPU against the oracle's decode."""
    sizes = [8192, 16384, 32768, 65536, 131072]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in ((0, 1, 2, 3, 4) if n < 100000 else (0, 2)):
            xs.append(H.synth_series(900 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    nf = len(off) - 1
    for comp in (A.AUTO, A.FFT, A.POLYNOMIAL):
        bro, chosen, _ = oracle.stream_compress(x, off, comp, True, me, 0)
        ref = oracle.decompress_data(bro)
        body_off, nfr = A.bro_open(bro)
        out = ctx.decompress_host(bro[body_off:])
        assert len(out) == len(ref)
        for i in range(nf):
            seg = slice(int(off[i]), int(off[i + 1]))
            if chosen[i] == oracle.FFT:
                n = int(off[i + 1] - off[i])
                scale = max(np.max(np.abs(ref[seg])), 1e-30)
                tol = (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5
                assert np.max(np.abs(out[seg] - ref[seg])) <= tol, (comp, i, np.max(np.abs(out[seg] - ref[seg])), tol)
            else:
                assert np.array_equal(out[seg], ref[seg]), (comp, i, chosen[i])


def test_roundtrip_long_series_e2e(ctx, A, oracle):
    """compress_data / decompress_data (main.rs:130-172) on a long series: GPU stream decoded by the
    GPU and by the oracle; MAPE bound as e2e.rs:234-248."""
    x = H.synth_series(43, 200000, klass=0)
    sizes = A.chunk_sizes(len(x))
    off = np.cumsum([0] + sizes).astype(np.uint64)
    rec, _, chosen, err = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
    out = ctx.decompress_host(rec)
    ref = oracle.decompress_data(A.bro_prefix(len(sizes)) + rec)
    assert len(out) == len(x) == len(ref)
    assert H.mape(x, out) <= ME5 and H.mape(x, ref) <= ME5
    assert np.max(np.abs(out - ref)) <= 30 * 1300 * 2.0 ** -23


# ---------------------------------------------------------------------------------------
# randomized sweep: many generators x lengths x error bounds
# ---------------------------------------------------------------------------------------
def _fuzz_frame(rng, n):
    kind = rng.integers(0, 12)
    t = np.arange(n, dtype=np.float64)
    if kind == 0:   # random walk
        v = 500.0 + np.cumsum(rng.normal(0, 1, n))
    elif kind == 1:  # steps / plateaus
        v = np.repeat(rng.integers(1, 2000, max(n // 7, 1) + 1), 7)[:n].astype(np.float64)
    elif kind == 2:  # sparse spikes on a constant
        v = np.full(n, 42.0); v[rng.integers(0, n, max(n // 20, 1))] = rng.uniform(1, 1e4)
    elif kind == 3:  # quantised sine (2 decimals)
        v = np.round(100 + 30 * np.sin(t / rng.uniform(2, 40)), 2)
    elif kind == 4:  # mixed sign
        v = rng.normal(0, 100, n)
    elif kind == 5:  # tiny magnitudes
        v = rng.uniform(1e-9, 1e-6, n)
    elif kind == 6:  # huge magnitudes
        v = rng.uniform(1e12, 1e15, n)
    elif kind == 7:  # small integers (u8)
        v = rng.integers(0, 4, n).astype(np.float64) + 1
    elif kind == 8:  # i16 integers with trend
        v = np.floor(1000 + 3 * t + rng.normal(0, 2, n))
    elif kind == 9:  # i32 integers
        v = np.floor(rng.uniform(1e5, 2e9, n))
    elif kind == 10:  # exact multi-tone (few FFT bins)
        v = 1000 + 100 * np.cos(2 * np.pi * 3 * t / max(n, 2)) + 50 * np.sin(2 * np.pi * 7 * t / max(n, 2))
    else:  # saw tooth with occasional zeros
        v = (t % 17) * 3.0
    return np.asarray(v, dtype=np.float64)


@pytest.mark.parametrize("seed,e", [(1, 5), (2, 1), (3, 0), (4, 3), (5, 10), (6, 50)])
def test_fuzz_auto(ctx, A, oracle, seed, e):
    rng = np.random.default_rng(seed)
    xs, offs = [], [0]
    for _ in range(150):
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), 256, 128, 512, rng.integers(600, 4097)],
                           p=[0.15, 0.45, 0.15, 0.05, 0.05, 0.15]))
        xs.append(_fuzz_frame(rng, n))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    me = float(np.float32(e) / np.float32(100))
    s = P.compare_batch(oracle, ctx, x, np.array(offs, dtype=np.uint64), A.AUTO, True, me)
    _log(P.assert_summary(s, 150, "fuzz seed %d e=%d codecs %s" % (seed, e, s["codecs"])))
    out = ctx.decompress_host(s["records"])
    ref = oracle.decompress_data(A.bro_prefix(150) + s["records"])
    assert len(out) == len(ref) == len(x)
    ch = s["chosen"]
    for i in range(150):
        seg = slice(int(offs[i]), int(offs[i + 1]))
        if ch[i] != oracle.FFT:
            assert np.array_equal(out[seg], ref[seg], equal_nan=True), (i, ch[i])


# ---------------------------------------------------------------------------------------
# pipelined entry point: packing on the context's stream, two scratch sets
# ---------------------------------------------------------------------------------------
def test_pipelined_batches_match_plain_calls(ctx, A):
    """Five different batches pushed back-to-back through atsc_compress_plan_dev_pipelined (with a
    plain call mixed in) give, batch by batch, the bytes of the single-stream call."""
    import torch

    dev = torch.device("cuda", 0)
    nf, F = 2048, 256
    off = H.frame_offsets(nf * F, F)
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream
    batches = [torch.from_numpy(H.synth_series(100 + b, nf * F)).to(dev) for b in range(5)]

    def fetch(o):
        total = int(o["rec_off"][-1].item())
        return (o["body"][:total].cpu().numpy().tobytes(), o["rec_off"].cpu().numpy().copy(),
                o["chosen"].cpu().numpy().copy(), o["err"].cpu().numpy().copy())

    ref = []
    o = plan.alloc_outputs(torch, dev)
    for d_x in batches:
        plan.compress(d_x, o, A.AUTO, True, ME5, 0, stream)
        torch.cuda.synchronize()
        ref.append(fetch(o))
    assert len({r[0] for r in ref}) == len(ref)  # the batches really differ

    outs = [plan.alloc_outputs(torch, dev) for _ in batches]
    for b, d_x in enumerate(batches):
        if b == 3:  # a plain call in the middle orders itself after the pending packing
            plan.compress(d_x, outs[b], A.AUTO, True, ME5, 0, stream)
        else:
            plan.compress(d_x, outs[b], A.AUTO, True, ME5, 0, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.current_stream().synchronize()  # only the caller's stream: join must cover the packing
    for b in range(len(batches)):
        got = fetch(outs[b])
        assert got[0] == ref[b][0], "batch %d bytes differ" % b
        assert np.array_equal(got[1], ref[b][1]) and np.array_equal(got[2], ref[b][2])
        assert np.array_equal(got[3], ref[b][3], equal_nan=True)
    plan.close() if hasattr(plan, "close") else None


_FAIL_SET_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import atsc_amd
from tests import helpers as H
me = float(np.float32(5) / np.float32(100))
ctx = atsc_amd.Context(0)
ctx.set_chains(4)
dev = torch.device("cuda:0")
for F, nf in ((256, 2048), (131072, 6)):
    xs = [H.synth_series(300 + b, F * nf, block=F * max(nf // 6, 1), class_shift=b) for b in range(3)]
    d_xs = [torch.from_numpy(v).to(dev) for v in xs]
    off = H.frame_offsets(F * nf, F)
    plan = ctx.plan(off)
    st = torch.cuda.current_stream().cuda_stream
    ref = []
    o = plan.alloc_outputs(torch, dev)
    for b in range(3):
        plan.compress(d_xs[b], o, atsc_amd.AUTO, True, me, 0, st)
        torch.cuda.synchronize()
        t = int(o["rec_off"][-1].item())
        ref.append(o["body"][:t].cpu().numpy().tobytes())
    outs = [plan.alloc_outputs(torch, dev) for _ in range(8)]
    for i in range(24):
        plan.compress(d_xs[i % 3], outs[i % 8], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
    plan.join(st)
    torch.cuda.synchronize()
    for i in range(16, 24):
        t = int(outs[i % 8]["rec_off"][-1].item())
        assert outs[i % 8]["body"][:t].cpu().numpy().tobytes() == ref[i % 3], (F, i)
    plan.close()
print("FAILSET-OK")
"""


@pytest.mark.parametrize("fail_from", [1, 2, 3, 5])
def test_pipelined_calls_fall_back_to_fewer_chains_when_a_scratch_set_does_not_fit(fail_from):
    """The first pipelined call builds two scratch sets per chain.  When set q cannot be allocated (ATSC_DEBUG_FAIL_SET=q
    lets its third block fail, as an out-of-memory would) the blocks it did get go back to the pool and the plan goes on
    with the sets it has: q = 1 -> one set, every call behind its predecessor; q = 2, 3 -> one chain; q = 5 -> two.
    The records are those of plain calls either way."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ATSC_DEBUG_FAIL_SET=str(fail_from), GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", _FAIL_SET_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAILSET-OK" in r.stdout, (r.stdout[-300:], r.stderr[-800:])


_TRIP_WIDTH_SCRIPT = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
import atsc_amd
from tests import helpers as H
ctx = atsc_amd.Context(0)
for me in (float(np.float32(5) / np.float32(100)), float(np.float32(1) / np.float32(100))):
    for n, nf in ((131072, 40), (32768, 24), (8192, 30)):
        x = np.concatenate([H.synth_series(640 + k, n, klass=k % 5) for k in range(nf)])
        off = H.frame_offsets(n * nf, n)
        rec, _, chosen, err = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
        out = ctx.decompress_host(rec)
        print("TRIPW", n, me, hashlib.sha256(rec).hexdigest(), hashlib.sha256(out.tobytes()).hexdigest(),
              hashlib.sha256(np.asarray(err).tobytes()).hexdigest())
"""


def test_the_two_tile_widths_of_the_first_fft_trip_agree_bit_for_bit():
    """k_large_trip243 comes in a 192- and a 512-thread form, picked by the size of the launch (atsc_large_fast.h).  A
    bucket's sum runs in list order whichever group of 16 lanes it falls to, so the two forms must give the same bits:
    records, reported errors and decoded samples of the same batches with either form forced (ATSC_TRIP_WIDE_MAX=0 /
    1000000, read once per process)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = []
    for wm in ("0", "1000000"):
        env = dict(os.environ, ATSC_TRIP_WIDE_MAX=wm)
        r = subprocess.run([sys.executable, "-c", _TRIP_WIDTH_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("TRIPW")]
        assert r.returncode == 0 and len(lines) == 6, (r.stdout[-300:], r.stderr[-800:])
        got.append(lines)
    assert got[0] == got[1]


@pytest.mark.parametrize("F,nf,chains", [(131072, 12, 2), (256, 4096, 4), (131072, 12, 1)])
def test_pipelined_input_release_lets_the_caller_refill_one_buffer(ctx, A, F, nf, chains):
    """The kernels of a pipelined call read d_samples on the context's streams, not in the caller's stream order
    (include/atsc_hip.h): a caller that keeps ONE input buffer and refills it batch after batch orders every refill
    behind atsc_plan_input_release.  Six batches through one buffer -- large frames (the fast path's kernels, on two
    chains or one) and small ones (four chains) -- give the bytes of plain calls on private buffers."""
    import torch

    dev = torch.device("cuda", 0)
    off = H.frame_offsets(nf * F, F)
    ctx.set_chains(chains)
    try:
        plan = ctx.plan(off)
        side = torch.cuda.Stream(device=dev)  # the caller's stream: refills and calls are enqueued here
        batches = [torch.from_numpy(H.synth_series(300 + b, nf * F, class_shift=b)).to(dev) for b in range(6)]

        def fetch(o):
            total = int(o["rec_off"][-1].item())
            return o["body"][:total].cpu().numpy().tobytes()

        o = plan.alloc_outputs(torch, dev)
        ref = []
        for d_x in batches:
            plan.compress(d_x, o, A.AUTO, True, ME5, 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ref.append(fetch(o))
        assert len(set(ref)) == len(ref)
        buf = torch.empty(nf * F, dtype=torch.float64, device=dev)
        outs = [plan.alloc_outputs(torch, dev) for _ in batches]
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for b, d_x in enumerate(batches):
                plan.input_release(side.cuda_stream)  # the refill below waits for the previous calls' last read
                buf.copy_(d_x, non_blocking=True)
                plan.compress(buf, outs[b], A.AUTO, True, ME5, 0, side.cuda_stream, pipelined=True)
            plan.join(side.cuda_stream)
        side.synchronize()
        for b in range(len(batches)):
            assert fetch(outs[b]) == ref[b], "batch %d differs (F=%d, %d chains)" % (b, F, chains)
        plan.close()
    finally:
        ctx.set_chains(2)


# ---------------------------------------------------------------------------------------
# forced Idw on large frames (polynomial.rs:375-393 at 4097 .. 131072 samples)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("bounded", [True, False])
def test_idw_large_frames(ctx, A, oracle, bounded):
    sizes = [4097, 5000, 8192, 12000, 20000]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in (0, 1, 2, 3):
            xs.append(H.synth_series(900 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.IDW, bounded, ME5)
    _log(P.assert_summary(s, len(off) - 1, "idw large frames bounded=%s" % bounded))
    assert s["tol"] == 0 and s["boundary"] == 0
    # decode: the oracle's stream through the GPU decoder, bit for bit
    bro, chosen, _ = oracle.stream_compress(x, off, A.IDW, bounded, ME5, 0)
    ref = oracle.decompress_data(bro)
    body_off, _ = A.bro_open(bro)
    out = ctx.decompress_host(bro[body_off:])
    assert np.array_equal(out, ref)


def test_idw_one_131072_frame(ctx, A, oracle):
    """The reference chunker's largest frame through forced Idw (O(n K) per ladder trip)."""
    x = H.synth_series(77, 131072, klass=0)
    off = np.array([0, len(x)], dtype=np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.IDW, True, ME5)
    _log(P.assert_summary(s, 1, "idw 131072"))
    assert s["tol"] == 0 and s["boundary"] == 0


@pytest.mark.parametrize("chains", [1, 2, 4])
def test_pipelined_chunker_framing_of_an_odd_length_series(ctx, A, chains):
    """A series whose length is no power of two, cut by the reference chunker (optimizer/mod.rs:78-98): large frames of
    several row dimensions (131072, 65536, ..., 8192 samples: the grid path per P), medium and small ones behind them.
    Batch after batch through the pipelined entry point, over 1, 2 and 4 chains: every batch's bytes are those of
    a plain call."""
    import torch

    dev = torch.device("cuda", 0)
    n = 131072 + 65536 + 32768 + 16384 + 8192 + 4096 + 1024 + 256 + 77
    sizes = A.chunk_sizes(n)
    assert sizes[0] == 131072 and 8192 in sizes and len(sizes) >= 8, sizes
    off = np.cumsum([0] + sizes).astype(np.uint64)
    batches = [torch.from_numpy(H.synth_series(1700 + b, n, klass=b % 4)).to(dev) for b in range(5)]
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream
    ref = []
    o = plan.alloc_outputs(torch, dev)
    for d_x in batches:
        plan.compress(d_x, o, A.AUTO, True, ME5, 0, stream)
        torch.cuda.synchronize()
        total = int(o["rec_off"][-1].item())
        ref.append(o["body"][:total].cpu().numpy().tobytes())
    ctx.set_chains(chains)
    try:
        rounds = 12
        outs = [plan.alloc_outputs(torch, dev) for _ in range(rounds)]
        for r in range(rounds):
            plan.compress(batches[r % len(batches)], outs[r], A.AUTO, True, ME5, 0, stream, pipelined=True)
        plan.join(stream)
        torch.cuda.synchronize()
        for r in range(rounds):
            total = int(outs[r]["rec_off"][-1].item())
            assert outs[r]["body"][:total].cpu().numpy().tobytes() == ref[r % len(batches)], (chains, r)
    finally:
        ctx.set_chains(2)


def test_independent_chains_on_one_gpu(ctx, A):
    """Three chains -- a context, a plan and a stream each -- push batches through the pipelined entry point at the
    same time (bench.py's `value_chains`, INTEGRATION.md "Several chains on one GPU"): every batch's bytes are those
    of a plain call on one context, whatever the other chains' kernels were doing on the same CUs."""
    import torch
    import atsc_amd

    dev = torch.device("cuda", 0)
    nf, F = 4096, 256
    off = H.frame_offsets(nf * F, F)
    batches = [torch.from_numpy(H.synth_series(700 + b, nf * F)).to(dev) for b in range(4)]
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream
    ref = []
    o = plan.alloc_outputs(torch, dev)
    for d_x in batches:
        plan.compress(d_x, o, A.AUTO, True, ME5, 0, stream)
        torch.cuda.synchronize()
        total = int(o["rec_off"][-1].item())
        ref.append(o["body"][:total].cpu().numpy().tobytes())

    C, rounds = 3, 4
    ctxs = [atsc_amd.Context(0) for _ in range(C)]
    plans = [c.plan(off) for c in ctxs]
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
    outs = [[p.alloc_outputs(torch, dev) for _ in range(rounds)] for p in plans]
    for r in range(rounds):
        for c in range(C):
            plans[c].compress(batches[(r + c) % len(batches)], outs[c][r], A.AUTO, True, ME5, 0,
                              streams[c].cuda_stream, pipelined=True)
    for c in range(C):
        plans[c].join(streams[c].cuda_stream)
    torch.cuda.synchronize()
    for r in range(rounds):
        for c in range(C):
            o = outs[c][r]
            total = int(o["rec_off"][-1].item())
            assert o["body"][:total].cpu().numpy().tobytes() == ref[(r + c) % len(batches)], (c, r)


def test_pipelined_adaptive_order_mixed_lengths(ctx, A):
    """Cost-ordered launches (atsc_ctx_set_adaptive_order) over a plan with several frame-length
    classes and a large frame: eight pipelined batches, alternating between two data sets, give the
    bytes of plain calls; the launch order never shows in the results."""
    import torch

    dev = torch.device("cuda", 0)
    lens = [256] * 700 + [64] * 90 + [300] * 40 + [512] * 60 + [1000] * 30 + [2048] * 12 + [4096] * 6 + [8192] + [100] * 50
    rng = np.random.default_rng(5)
    rng.shuffle(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(off[-1])
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream
    data = [torch.from_numpy(H.synth_series(300 + b, n, block=4096)).to(dev) for b in range(2)]

    def fetch(o):
        total = int(o["rec_off"][-1].item())
        return o["body"][:total].cpu().numpy().tobytes(), o["chosen"].cpu().numpy().copy()

    ref = []
    o = plan.alloc_outputs(torch, dev)
    for d_x in data:
        plan.compress(d_x, o, A.AUTO, True, ME5, 0, stream)
        torch.cuda.synchronize()
        ref.append(fetch(o))
    ctx.set_adaptive_order(True)
    outs = [plan.alloc_outputs(torch, dev) for _ in range(8)]
    for b in range(8):
        plan.compress(data[(b // 3) % 2], outs[b], A.AUTO, True, ME5, 0, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.current_stream().synchronize()
    for b in range(8):
        got = fetch(outs[b])
        want = ref[(b // 3) % 2]
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), b


# ---------------------------------------------------------------------------------------
# RLE with few runs (one run per lane on the GPU): grouping, ordering and byte layout
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rle_few_runs_fuzz(ctx, A, oracle, seed):
    rng = np.random.default_rng(seed)
    pools = [np.array([0.0, -0.0, 1.0, -1.0, 2.5, 1e300, -1e-300, 255.0, 256.0, 65535.0, 70000.0, -3.0]),
             np.arange(0, 9, dtype=np.float64), np.array([7.0, 7.5]), np.array([100.0, 200.0, 300.0, 100000.0])]
    xs, offs = [], [0]
    for _ in range(260):
        n = int(rng.choice([1, 2, 3, 17, 64, 100, 128, 200, 256, 300, 512]))
        runs = int(rng.integers(1, min(n, 20) + 1))
        pool = pools[int(rng.integers(0, len(pools)))]
        cuts = np.sort(rng.choice(np.arange(1, n), size=runs - 1, replace=False)) if runs > 1 else np.array([], dtype=int)
        bounds = np.concatenate([[0], cuts, [n]]).astype(int)
        f = np.empty(n)
        prev = None
        for a, b in zip(bounds[:-1], bounds[1:]):
            v = pool[int(rng.integers(0, len(pool)))]
            while prev is not None and v == prev and len(pool) > 1:  # adjacent runs must differ (== compare)
                v = pool[int(rng.integers(0, len(pool)))]
            f[a:b] = v
            prev = v
        xs.append(f)
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    for comp, bounded in ((A.RLE, False), (A.AUTO, True)):
        s = P.compare_batch(oracle, ctx, x, off, comp, bounded, ME5)
        _log(P.assert_summary(s, len(off) - 1, "rle few runs seed %d comp %d codecs %s" % (seed, comp, s["codecs"])))
        if comp == A.RLE:
            assert s["tol"] == 0 and s["boundary"] == 0
            out = ctx.decompress_host(s["records"])
            # not compared with x: at integer bitdepths the reference loses the sign of -0.0 and
            # saturates integer-valued samples beyond i32; the decode has to equal the oracle's
            ref = oracle.decompress_data(A.bro_prefix(len(off) - 1) + s["records"])
            assert np.array_equal(out.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("flen", [128, 256, 512, 1024, 2048, 4096])
def test_fixed_length_instantiation_matches_generic(ctx, A, monkeypatch, flen):
    """Uniform batches of 128 / 256 / 512 / 1024 / 2048 / 4096-sample frames run k_compress<.., FN> (geometry
    and FFT stage list folded at compile time); ATSC_NO_UNIFORM forces the table-driven instantiation.
    Same bytes, same errors."""
    import torch

    dev = torch.device("cuda", 0)
    ctx.enable_diag(False)  # (a diagnostics record would route the uniform batch to the table-driven kernels too)
    nf = (1 << 20) // flen
    x = H.synth_series(11, nf * flen, block=8192)
    off = H.frame_offsets(len(x), flen)
    d_x = torch.from_numpy(x).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    got = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("ATSC_NO_UNIFORM", "1")
        plan = ctx.plan(off)
        for me in (ME5, ME1):
            o = plan.alloc_outputs(torch, dev)
            plan.compress(d_x, o, A.AUTO, True, me, 0, stream)
            torch.cuda.synchronize()
            total = int(o["rec_off"][-1].item())
            got.append((o["body"][:total].cpu().numpy().tobytes(), o["err"].cpu().numpy().copy()))
    assert got[0][0] == got[2][0] and got[1][0] == got[3][0]
    assert np.array_equal(got[0][1], got[2][1], equal_nan=True) and np.array_equal(got[1][1], got[3][1], equal_nan=True)


@pytest.mark.parametrize("form", ["tiled", "stages"])
def test_large_frames_both_transform_forms(ctx, A, oracle, monkeypatch, form):
    """The large tier picks its transform by batch size (LDS-tiled two-pass for many frames, stage by
    stage for few); both forms, forced through ATSC_LARGE_FFT, meet the same parity bars, encode and decode."""
    monkeypatch.setenv("ATSC_LARGE_FFT", form)
    sizes = [4097, 6500, 8192, 20000, 59000, 131072]   # 6500, 59000 -> odd transform lengths 3^8, 3^10
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in ((0, 1) if n > 50000 else (0, 1, 2, 3)):
            xs.append(H.synth_series(950 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    s = P.compare_batch(oracle, ctx, x, off, A.AUTO, True, ME5)
    _log(P.assert_summary(s, len(off) - 1, "large frames, %s transform, codecs %s" % (form, s["codecs"])))
    s = P.compare_batch(oracle, ctx, x, off, A.FFT, True, ME1)
    _log(P.assert_summary(s, len(off) - 1, "large frames, %s transform, forced fft e=1%%" % form))
    bro, chosen, _ = oracle.stream_compress(x, off, A.FFT, True, ME5, 0)
    ref = oracle.decompress_data(bro)
    body_off, _ = A.bro_open(bro)
    out = ctx.decompress_host(bro[body_off:])
    for i in range(len(off) - 1):
        seg = slice(int(off[i]), int(off[i + 1]))
        n = int(off[i + 1] - off[i])
        scale = max(np.max(np.abs(ref[seg])), 1e-30)
        tol = (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5
        assert np.max(np.abs(out[seg] - ref[seg])) <= tol, (form, i)


def test_compress_is_graph_capturable(ctx, A):
    """atsc_compress_plan_dev only enqueues kernels on the caller's stream: it can be captured into a
    HIP graph and replayed on new data in the same buffers."""
    import torch

    dev = torch.device("cuda", 0)
    nf, F = 4096, 256
    off = H.frame_offsets(nf * F, F)
    plan = ctx.plan(off)
    d_x = torch.empty(nf * F, dtype=torch.float64, device=dev)
    outs = plan.alloc_outputs(torch, dev)
    batches = [torch.from_numpy(H.synth_series(500 + b, nf * F, block=4096)).to(dev) for b in range(3)]

    def fetch():
        total = int(outs["rec_off"][-1].item())
        return outs["body"][:total].cpu().numpy().tobytes()

    want = []
    for b in batches:
        d_x.copy_(b)
        plan.compress(d_x, outs, A.AUTO, True, ME5, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want.append(fetch())
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        plan.compress(d_x, outs, A.AUTO, True, ME5, 0, side.cuda_stream)  # warm-up outside the capture
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            plan.compress(d_x, outs, A.AUTO, True, ME5, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i, b in enumerate(batches):
        d_x.copy_(b)
        outs["body"].zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert fetch() == want[i], i


@pytest.mark.parametrize("comp,bounded,level", [("AUTO", True, 3), ("AUTO", True, 6), ("FFT", False, 0),
                                                ("RLE", False, 0), ("POLYNOMIAL", True, 0), ("IDW", True, 0)])
def test_pipelined_modes_match_plain_calls(ctx, A, comp, bounded, level):
    """The pipelined entry point with the sample-level trial launch, the unpadded-FFT sub-plan and
    forced codecs: four batches back-to-back give the bytes of plain calls."""
    import torch

    dev = torch.device("cuda", 0)
    lens = [256] * 300 + [1024] * 24 + [4096] * 6 + [8192] * 2 + [200] * 30
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(off[-1])
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream
    cid = getattr(A, comp)
    data = [torch.from_numpy(H.synth_series(600 + b, n, block=2048)).to(dev) for b in range(2)]

    def fetch(o):
        total = int(o["rec_off"][-1].item())
        return o["body"][:total].cpu().numpy().tobytes()

    ref = []
    o = plan.alloc_outputs(torch, dev)
    for d_x in data:
        plan.compress(d_x, o, cid, bounded, ME5, level, stream)
        torch.cuda.synchronize()
        ref.append(fetch(o))
    outs = [plan.alloc_outputs(torch, dev) for _ in range(4)]
    for b in range(4):
        plan.compress(data[b % 2], outs[b], cid, bounded, ME5, level, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.current_stream().synchronize()
    for b in range(4):
        assert fetch(outs[b]) == ref[b % 2], (comp, b)


def test_host_entry_points_reuse_device_memory(ctx, A):
    """The host-pointer entry points build a plan and their buffers per call; the context's pool hands
    the same device blocks out again, so repeated calls do not grow the footprint, and results repeat."""
    import torch

    x = H.synth_series(21, 600000)
    off = H.frame_offsets(len(x), 256)
    first = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
    bro = A.compress_data(ctx, x, A.AUTO, 5)
    A.decompress_data(ctx, bro)  # (the context keeps the plans of recent layouts: every entry point once before measuring)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        again = ctx.compress_host(x, off, A.AUTO, True, ME5, 0)
        assert again[0] == first[0]
        assert A.compress_data(ctx, x, A.AUTO, 5) == bro
        out = A.decompress_data(ctx, bro)
        assert len(out) == len(x)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)


def _record(tag, n, payload):
    """One frame record of a .bro body (frame/mod.rs:57-66 via bincode): size, samples, codec, payload."""
    return P.H_varint(41) + P.H_varint(n) + P.H_varint(tag) + P.H_varint(len(payload)) + payload


def test_decompress_foreign_fft_streams(ctx, A, oracle):
    """FFT payloads this library's encoder never writes, decoded like the oracle decodes them: more entries
    than the ladder can store (the decoder's serial parse + dense transform), duplicate positions (later
    entries overwrite), positions above L/2 (mirrored), entries of both widths in one stream."""
    import struct
    rng = np.random.default_rng(77)
    # (513 ... 4096 samples with 16 or more entries: the decoder's inverse transform in LDS, even and odd L; fewer
    # entries, or more than fit beside the spectrum: the direct sum)
    # (power-of-two chunks from 8192 samples with at most 1344 entries: the grid decoder -- k_large_dparse's block walks over
    # entries of both widths, its list of repeated positions (up to 64 of them) and the walk over every later entry
    # beyond that; positions are u16: below 65536 whatever the transform length)
    for n, cnt, dups in ((4097, 700, 20), (8192, 3000, 20), (6561, 60, 20), (20000, 1500, 20), (256, 40, 20), (1000, 300, 20),
                         (100, 30, 20), (40, 12, 20), (2048, 500, 20), (4096, 1200, 20), (4096, 1700, 20), (3000, 90, 20),
                         (600, 17, 20), (1024, 15, 20), (2048, 16, 20), (131072, 1300, 20), (131072, 1000, 200),
                         (131072, 1344, 64), (131072, 9, 2), (65536, 600, 70), (32768, 327, 5), (16384, 160, 40), (8192, 81, 3)):
        L = int(oracle.next_size(n)) if n >= 128 else n  # fft.rs:432-444: no padding below 128 samples
        pos = rng.integers(0, min(L, 65536), size=cnt)
        pos[: cnt // 3] = rng.integers(0, min(251, L), size=cnt // 3)  # one-byte positions among the wide ones
        rng.shuffle(pos)
        ndup = min(dups, cnt // 4)
        pos[cnt // 2: cnt // 2 + ndup] = pos[:ndup]                      # duplicates: the later entry wins
        re = rng.normal(0, 50.0, size=cnt).astype(np.float32)
        im = rng.normal(0, 50.0, size=cnt).astype(np.float32)
        pay = bytes([15]) + P.H_varint(cnt)
        for p, r, i in zip(pos, re, im):
            pay += P.H_varint(int(p)) + struct.pack("<ff", r, i)
        pay += struct.pack("<ff", 40.0, -40.0)
        ref = np.array(oracle.decompress(oracle.FFT, pay, n))
        out = ctx.decompress_host(_record(oracle.FFT, n, pay))
        assert len(out) == n
        scale = max(float(np.max(np.abs(re))), 1.0) * cnt / L
        tol = (4 + np.log2(n)) * max(scale, 40.0) * 2.0 ** -23 + 1.00001e-5
        assert np.max(np.abs(out - ref)) <= tol, (n, cnt, float(np.max(np.abs(out - ref))), tol)


def test_decompress_mixed_width_varints(ctx, A, oracle):
    """Polynomial points and Noop values whose zigzag varints mix 1-, 3-, 5- and 9-byte widths at random:
    the 64-at-a-time parse has to settle lane by lane; decode must equal the oracle's bit for bit."""
    rng = np.random.default_rng(78)
    for n in (4097, 9000, 40000, 64, 300, 2000):
        # Noop: integers around the 251 marker and the 16/32-bit boundaries
        pool = np.array([0, 1, -1, 124, 125, 126, -125, -126, -127, 300, -300, 32767, -32768, 32768, 70000, -70000,
                         2 ** 31 - 1, -2 ** 31, 2 ** 31, 2 ** 40, -2 ** 40], dtype=np.float64)
        x = rng.choice(pool, size=n)
        po, _ = oracle.compress(A.NOOP, x, False, 0.0)
        ref = np.array(oracle.decompress(A.NOOP, po, n))
        out = ctx.decompress_host(_record(A.NOOP, n, po))
        assert np.array_equal(out, ref), ("noop", n)
        # Polynomial with i16 / i32 points of mixed widths
        for lim in (200, 40000):
            x = np.round(rng.uniform(-lim, lim, size=n))
            x[rng.integers(0, n, size=n // 7)] = rng.integers(-3, 4, size=n // 7)
            po, _ = oracle.compress(A.POLYNOMIAL, x, True, 0.5)
            ref = np.array(oracle.decompress(A.POLYNOMIAL, po, n))
            out = ctx.decompress_host(_record(A.POLYNOMIAL, n, po))
            assert np.array_equal(out, ref), ("poly", n, lim)


def test_compress_data_long_input_with_nonfinite_samples(ctx, A):
    """OptimizerPlan::plan drops NaN / infinite samples (optimizer/mod.rs:47-49).  On a long input the scan
    runs beside the GPU work and a hit sends the cleaned copy through again: the stream has to equal the one
    made from the cleaned series, and a clean long input has to equal itself compressed in two halves' framing
    (same chunker) -- i.e. the speculative first pass leaves nothing behind."""
    x = H.synth_series(31, 1500000)
    clean_bro = A.compress_data(ctx, x, A.AUTO, 5)
    y = x.copy()
    ys = []
    cut = [7, 131072, 700001, 1499999]
    prev = 0
    for k, c in enumerate(cut):
        ys.append(y[prev:c])
        ys.append(np.array([np.nan, np.inf, -np.inf][k % 3: k % 3 + 1]))
        prev = c
    ys.append(y[prev:])
    dirty = np.concatenate(ys)
    assert len(dirty) == len(x) + len(cut)
    assert A.compress_data(ctx, dirty, A.AUTO, 5) == clean_bro
    assert A.compress_data(ctx, x, A.AUTO, 5) == clean_bro
    out = A.decompress_data(ctx, clean_bro)
    assert len(out) == len(x)


def test_large_frames_sparse_and_dense_inverse_agree(ctx, A, oracle, monkeypatch):
    """The large tier's ladder reconstructs from the sparse bin list (sparse_inverse); ATSC_LARGE_DENSE keeps
    the dense transforms through the workspace.  Same forward transform, so the stored coefficients are the
    same numbers: the two must pick the same codecs and write the same bytes unless a frame sits on a decision
    threshold (counted, rare), and both decoders must reproduce the oracle's decode of the sparse stream."""
    sizes = [4097, 6561, 8192, 20000, 65536, 131072]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in ((0, 1) if n > 50000 else (0, 1, 2, 3)):
            xs.append(H.synth_series(1200 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    nf = len(off) - 1
    for comp, me in ((A.AUTO, ME5), (A.FFT, ME1)):
        monkeypatch.delenv("ATSC_LARGE_DENSE", raising=False)
        rec_s, off_s, ch_s, err_s = ctx.compress_host(x, off, comp, True, me, 0)
        out_s = ctx.decompress_host(rec_s)
        monkeypatch.setenv("ATSC_LARGE_DENSE", "1")
        rec_d, off_d, ch_d, err_d = ctx.compress_host(x, off, comp, True, me, 0)
        out_d = ctx.decompress_host(rec_s)
        monkeypatch.delenv("ATSC_LARGE_DENSE", raising=False)
        assert np.array_equal(ch_s, ch_d)
        fs = H.parse_bro_body(rec_s, with_count=False)
        fd = H.parse_bro_body(rec_d, with_count=False)
        differ = sum(1 for a, b in zip(fs, fd) if a != b)
        assert differ <= 1, (comp, differ)
        tol = P.FFT_ERR_ATOL + P.FFT_ERR_RTOL * np.abs(err_d) + np.array(
            [P.fft_err_noise(x[int(off[i]):int(off[i + 1])]) for i in range(nf)])
        assert np.all(np.abs(err_s - err_d) <= tol) or differ
        ref = np.array(oracle.decompress_data(A.bro_prefix(nf) + rec_s))
        for i in range(nf):
            seg = slice(int(off[i]), int(off[i + 1]))
            n = seg.stop - seg.start
            if ch_s[i] == oracle.FFT:
                scale = max(np.max(np.abs(ref[seg])), 1e-30)
                t = (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5
                assert np.max(np.abs(out_s[seg] - ref[seg])) <= t and np.max(np.abs(out_d[seg] - ref[seg])) <= t, (comp, i)
            else:
                assert np.array_equal(out_s[seg], ref[seg]) and np.array_equal(out_d[seg], ref[seg]), (comp, i)


def test_large_frames_grid_and_single_kernel_forms_agree(ctx, A, monkeypatch):
    """A batch of few large frames runs its sample walks and the first FFT trip's tiles as grids over the whole
    GPU (k_large_stats, k_large_poly1, k_large_trip_tiles, k_decompress_large_tiles); ATSC_LARGE_NO_TRIP_TILES /
    ATSC_LARGE_DECODE_ONE_KERNEL keep everything of a frame on its own workgroup.  Same arithmetic per sample,
    different summation order of the error sums: same codecs and bytes (unless a frame sits on a decision
    threshold), errors equal to summation-order accuracy, decoded samples identical."""
    sizes = [4097, 6561, 8192, 20000, 65536, 131072, 131072]
    xs, offs = [], [0]
    for k, n in enumerate(sizes):
        for c in (0, 1, 2, 3):
            xs.append(H.synth_series(1300 + k, n, klass=c))
            offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    for comp, me in ((A.AUTO, ME5), (A.AUTO, ME1), (A.FFT, ME5)):
        rec_g, _, ch_g, err_g = ctx.compress_host(x, off, comp, True, me, 0)
        out_g = ctx.decompress_host(rec_g)
        monkeypatch.setenv("ATSC_LARGE_NO_TRIP_TILES", "1")
        monkeypatch.setenv("ATSC_LARGE_DECODE_ONE_KERNEL", "1")
        rec_1, _, ch_1, err_1 = ctx.compress_host(x, off, comp, True, me, 0)
        out_1 = ctx.decompress_host(rec_g)
        monkeypatch.delenv("ATSC_LARGE_NO_TRIP_TILES")
        monkeypatch.delenv("ATSC_LARGE_DECODE_ONE_KERNEL")
        assert np.array_equal(ch_g, ch_1)
        fg = H.parse_bro_body(rec_g, with_count=False)
        f1 = H.parse_bro_body(rec_1, with_count=False)
        differ = sum(1 for a, b in zip(fg, f1) if a != b)
        assert differ <= 1, (comp, me, differ)
        if not differ:
            assert np.allclose(err_g, err_1, rtol=1e-9, atol=1e-15, equal_nan=True)
        assert np.array_equal(out_g, out_1, equal_nan=True)


def test_many_frames_left_to_the_general_kernels_are_shared_out(ctx, A, oracle):
    """Behind the grid paths the general kernels run 256 workgroups at most, which share out the LIST of frames left to
    them (k_compress_large<0> / k_decompress_large<0>, fb_append): with more listed frames than workgroups a workgroup
    takes several frames in turn.  Encoder: 640 frames of 8192 samples at e = 0.1 % -- ladders that go on well past
    their first trip, i.e. left by k_large_decide2 -- forced FFT and auto, against the oracle's encoder.  Decoder: 640 records, Noop frames (left
    alone by k_large_dparse) between FFT and polynomial ones (taken by it), against the oracle's decode."""
    n, nf = 8192, 640
    x = np.concatenate([H.synth_series(7100 + k, n, klass=k % 3) for k in range(nf)])
    off = H.frame_offsets(n * nf, n)
    me = float(np.float32(0.1) / np.float32(100))
    for comp in (A.FFT, A.AUTO):  # (--compressor fft: every ladder runs to its end; auto: the polynomial wins most frames)
        s = P.compare_batch(oracle, ctx, x, off, comp, True, me)
        _log(P.assert_summary(s, nf, "640 frames of 8192 samples at e = 0.1 %%, compressor %d, codecs %s" % (comp, s["codecs"])))
        if comp == A.FFT:
            ks = [len(H.parse_fft_payload(fr[3])[0]) for fr in H.parse_bro_body(s["records"], with_count=False)]
            assert sum(1 for k in ks if k > n // 100 + 1) > 300, "the ladders were meant to go on past their first trip"
    # decoder: the oracle's Noop records of the odd frames between its auto records of the even ones
    bro_a, _, _ = oracle.stream_compress(x, off, A.AUTO, True, ME5, 0)
    bro_n, _, _ = oracle.stream_compress(x, off, A.NOOP, False, 0.0, 0)
    fa = H.parse_bro_body(bytes(bro_a)[A.bro_open(bro_a)[0]:], with_count=False)
    fn = H.parse_bro_body(bytes(bro_n)[A.bro_open(bro_n)[0]:], with_count=False)
    recs = b"".join(_record(f[2], f[1], f[3]) for f in (fn[i] if i % 2 else fa[i] for i in range(nf)))
    ref = np.array(oracle.decompress_data(A.bro_prefix(nf) + recs))
    out = ctx.decompress_host(recs)
    for i in range(nf):
        seg = slice(i * n, (i + 1) * n)
        if i % 2 == 0 and fa[i][2] == A.FFT:
            scale = max(np.max(np.abs(ref[seg])), 1e-30)
            assert np.max(np.abs(out[seg] - ref[seg])) <= (4 + np.log2(n)) * scale * 2.0 ** -23 + 1.00001e-5, i
        else:
            assert np.array_equal(out[seg], ref[seg]), (i, fa[i][2])


def test_large_fast_path_matches_general_kernel(ctx, A, oracle, monkeypatch):
    """131072-sample frames under the auto selector take the grid path of atsc_large_fast.h (column / row transforms
    in registers, per-frame decisions as their own launches, the tile kernel for the first ladder trip and for the
    decoder); ATSC_LARGE_NO_FAST keeps the general per-frame kernels.  Two forward transforms with different
    butterfly orders: the coefficients agree to f32 accuracy, so the two forms must choose the same codecs with
    the same K, report the same error to the decode bar's accuracy, and both decoders must reproduce the oracle's
    decode of either stream (polynomial / RLE / constant frames bit for bit)."""
    xs, offs = [], [0]
    for F in (131072, 131072, 65536, 32768, 16384, 8192):
        for c in (0, 1, 2, 3, 4):
            xs.append(H.synth_series(1500 + c + F % 97, F, klass=c))
            offs.append(offs[-1] + F)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    nf = len(off) - 1
    for me in (ME5, ME1):
        monkeypatch.delenv("ATSC_LARGE_NO_FAST", raising=False)
        rec_f, _, ch_f, err_f = ctx.compress_host(x, off, A.AUTO, True, me, 0)
        out_ff = ctx.decompress_host(rec_f)
        monkeypatch.setenv("ATSC_LARGE_NO_FAST", "1")
        rec_g, _, ch_g, err_g = ctx.compress_host(x, off, A.AUTO, True, me, 0)
        out_fg = ctx.decompress_host(rec_f)  # the fast path's stream through the general decoder
        monkeypatch.delenv("ATSC_LARGE_NO_FAST", raising=False)
        assert np.array_equal(ch_f, ch_g), (me, ch_f, ch_g)
        ff = H.parse_bro_body(rec_f, with_count=False)
        fg = H.parse_bro_body(rec_g, with_count=False)
        ref = np.array(oracle.decompress_data(A.bro_prefix(nf) + rec_f))
        for i in range(nf):
            seg = slice(int(off[i]), int(off[i + 1]))
            if ch_f[i] == oracle.FFT:
                ka, kb = H.parse_fft_payload(ff[i][3])[0], H.parse_fft_payload(fg[i][3])[0]
                assert len(ka) == len(kb), (me, i, len(ka), len(kb))
                tol = P.FFT_ERR_ATOL + P.FFT_ERR_RTOL * abs(err_g[i]) + P.fft_err_noise(x[seg])
                assert abs(err_f[i] - err_g[i]) <= tol, (me, i, err_f[i], err_g[i])
                scale = max(np.max(np.abs(ref[seg])), 1e-30)
                t = (4 + np.log2(seg.stop - seg.start)) * scale * 2.0 ** -23 + 1.00001e-5
                assert np.max(np.abs(out_ff[seg] - ref[seg])) <= t and np.max(np.abs(out_fg[seg] - ref[seg])) <= t, (me, i)
            else:
                assert ff[i] == fg[i], (me, i)
                assert np.array_equal(out_ff[seg], ref[seg]) and np.array_equal(out_fg[seg], ref[seg]), (me, i)


# ---------------------------------------------------------------------------------------
# Bins of equal norm: the admission order is the reference's BinaryHeap pop order (fft.rs:231-257)
# ---------------------------------------------------------------------------------------
def _gpu_heap_order(A, norms, k):
    import ctypes as C

    a = np.ascontiguousarray(np.asarray(norms, dtype=np.float32))
    out = np.zeros(max(k, 1), dtype=np.uint32)
    fn = A.capi.lib().atsc_internal_heap_order  # test hook, atsc_internal.h
    fn.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    fn.restype = C.c_int
    rc = fn(a.ctypes.data_as(C.POINTER(C.c_float)), len(a), k, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    return out[:k]


def test_fft_tie_order_heap_replay_matches_binaryheap(ctx, A, oracle):
    """The kernels' heap replay (hp_*, atsc_device.h) against the oracle's restatement of
    std::collections::BinaryHeap (Rust 1.81) on norm arrays full of ties: a few distinct values, all
    equal, two values, sorted, reversed, with zeros -- 7 ... 2188 bins (n = 12 ... 4096), every K."""
    rng = np.random.default_rng(12)
    cases = 0
    for bins in (7, 13, 33, 65, 73, 145, 163, 289, 577, 1094, 2188):
        pats = [np.ones(bins), np.arange(bins) % 2 + 1.0, np.arange(bins) % 3 + 1.0, (np.arange(bins) // 4) + 1.0,
                (bins - np.arange(bins)) // 3 + 1.0, rng.integers(1, 4, bins).astype(float),
                rng.integers(1, max(bins // 8, 2), bins).astype(float), rng.integers(0, 3, bins).astype(float),
                np.round(rng.random(bins) * 8) / 8 + 0.125, rng.random(bins)]
        for pat in pats:
            k = bins if bins <= 600 else 450
            g = _gpu_heap_order(A, pat, k)
            o = oracle.heap_order(pat, k)
            assert np.array_equal(g, o), (bins, pat[:16], g[:16], o[:16])
            # and it is a descending-norm order, as any heap's
            assert np.all(np.diff(np.asarray(pat, dtype=np.float32)[g]) <= 0)
            cases += 1
    _log("heap replay == BinaryHeap order on %d tie patterns" % cases)


@pytest.mark.parametrize("bounded", [True, False])
def test_fft_tie_order_impulse_frames(ctx, A, oracle, bounded):
    """Frames whose spectrum ties exactly in both implementations: x = [a, 0, 0, ...] transformed at its own
    length (every bin is (a, +0): the only non-zero input meets twiddle 1 in any butterfly order), i.e. the
    bounded path below 128 samples and FFT::compress (no padding) at any 2^a 3^b length.  All bins tie, so the
    stored order is purely the heap's: the payloads must be byte-identical to the oracle's."""
    lens = [8, 12, 17, 31, 64, 100, 127] if bounded else [12, 64, 127, 128, 144, 256, 288, 512, 1024, 2048, 2187, 4096]
    n_exact = 0
    for n in lens:
        for a in (5.0, -3.25, 1000.0):
            x = np.zeros(n)
            x[0] = a
            off = np.array([0, n], dtype=np.uint64)
            rec, _, chosen, _ = ctx.compress_host(x, off, A.FFT, bounded, ME5, 0)
            po, _ = oracle.compress(oracle.FFT, x, bounded, ME5)
            (fs, sc, tag, payload), = H.parse_bro_body(rec, with_count=False)
            fg, _, _ = H.parse_fft_payload(payload)
            fo, _, _ = H.parse_fft_payload(po)
            assert [f[0] for f in fg] == [f[0] for f in fo], (n, a, [f[0] for f in fg][:12], [f[0] for f in fo][:12])
            assert payload == po, (n, a)
            n_exact += 1
    _log("impulse frames (all bins tie), bounded=%s: %d payloads byte-identical to the oracle" % (bounded, n_exact))


def test_fft_tie_order_mixed_batch(ctx, A, oracle):
    """Tie frames inside an ordinary batch (the one-wavefront class switches to the heap replay in the
    middle of its ladder; its neighbours are unaffected): auto and forced FFT, frames of 64 and 100 samples."""
    for n in (64, 100):
        xs = []
        for k in range(40):
            if k % 4 == 0:
                v = np.zeros(n)
                v[0] = 10.0 + k
            else:
                v = H.synth_series(700 + k, n, klass=k % 4)
            xs.append(v)
        x = np.concatenate(xs)
        off = H.frame_offsets(len(x), n)
        for comp in (A.FFT, A.AUTO):
            s = P.compare_batch(oracle, ctx, x, off, comp, True, ME5)
            _log(P.assert_summary(s, 40, "tie frames in a mixed batch n=%d comp=%d" % (n, comp)))
            assert s["tie"] == 0
