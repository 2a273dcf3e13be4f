"""Decoder fuzz soak (dev aid, GPU box only): fuzz_soak's batches compressed by the ORACLE, decoded on the GPU and
compared with the oracle's decode -- bit for bit on every codec but FFT, within the decode tolerance there."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atsc_amd as A
from oracle import oracle as orc
from tests.test_gpu_parity import _fuzz_frame

orc.build()
ctx = A.Context(0)
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 130)
large = os.environ.get("FUZZ_LARGE") in ("1", "2")  # 2: only the power-of-two chunks 8192 ... 131072 (the decoder's grid path)
large_sizes = [8192, 16384, 32768, 65536, 131072] if os.environ.get("FUZZ_LARGE") == "2" else [4097, 5000, 6561, 8192, 12000, 20000, 32768, 40000]
bad = 0; frames = 0; by_codec = {}
for seed in seeds:
    rng = np.random.default_rng(seed)
    e = int(rng.choice([0, 1, 2, 3, 5, 10, 20, 50])); me = float(np.float32(e) / np.float32(100))
    xs, offs = [], [0]
    for _ in range(30 if large else 200):
        n = int(rng.choice(large_sizes)) if large else int(rng.choice(
            [rng.integers(1, 40), rng.integers(40, 600), 256, 128, 512, 1024, 2048, 4096, rng.integers(600, 4097)],
            p=[0.1, 0.35, 0.15, 0.05, 0.05, 0.05, 0.05, 0.05, 0.15]))
        xs.append(_fuzz_frame(rng, n)); offs.append(offs[-1] + n)
    x = np.concatenate(xs); off = np.array(offs, dtype=np.uint64)
    comp, bounded = [(A.AUTO, True), (A.FFT, True), (A.POLYNOMIAL, True), (A.RLE, False), (A.NOOP, False), (A.IDW, True)][seed % 6]
    if comp == A.IDW and large:
        comp = A.POLYNOMIAL
    bro, chosen, _ = orc.stream_compress(x, off, comp, bounded, me, 0)
    ref = np.array(orc.decompress_data(bro))
    body_off, nfr = A.bro_open(bro)
    out = ctx.decompress_host(bro[body_off:])
    assert len(out) == len(ref)
    nbad = 0
    for i in range(len(off) - 1):
        seg = slice(int(off[i]), int(off[i + 1])); n = seg.stop - seg.start
        by_codec[int(chosen[i])] = by_codec.get(int(chosen[i]), 0) + 1
        if chosen[i] == orc.FFT:
            scale = max(float(np.max(np.abs(ref[seg]))), 1e-30)
            tol = (4 + np.log2(max(n, 2))) * scale * 2.0 ** -23 + 1.00001e-5
            ok = bool(np.all(np.abs(out[seg] - ref[seg]) <= tol) or np.array_equal(out[seg], ref[seg], equal_nan=True))
        else:
            ok = np.array_equal(out[seg], ref[seg], equal_nan=True)
        if not ok:
            nbad += 1
            if nbad <= 3:
                d = np.nonzero(~((out[seg] == ref[seg]) | (np.isnan(out[seg]) & np.isnan(ref[seg]))))[0]
                print("   FAIL seed %d frame %d n=%d codec %d: %d samples differ, first j=%d gpu=%r ref=%r" % (
                    seed, i, n, chosen[i], len(d), d[0], out[seg][d[0]], ref[seg][d[0]]), flush=True)
    frames += len(off) - 1; bad += nbad
    print("seed %d e=%d comp=%d frames=%d fail=%d" % (seed, e, comp, len(off) - 1, nbad), flush=True)
print("TOTAL frames", frames, "codecs", by_codec, "failures", bad)
