"""Wider fuzz soak of the parity comparison than the committed tests run (dev aid, GPU box only):
seeds x error bounds x (uniform 256-sample batches | mixed lengths), auto selector and forced codecs.
Prints one summary line per batch and every failing frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atsc_amd as A
from oracle import oracle as orc
from tests import parity as P
from tests.test_gpu_parity import _fuzz_frame

orc.build()
ctx = A.Context(0)
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 130)
bad = 0
tot = {"exact": 0, "tol": 0, "tie": 0, "boundary": 0, "frames": 0}
for seed in seeds:
    rng = np.random.default_rng(seed)
    e = int(rng.choice([0, 1, 2, 3, 5, 10, 20, 50]))
    me = float(np.float32(e) / np.float32(100))
    uniform = seed % 3 == 0
    xs, offs = [], [0]
    large = os.environ.get("FUZZ_LARGE") in ("1", "2")  # frames of the large tier instead (fewer of them)
    # FUZZ_LARGE=2: only the chunker's power-of-two chunks (M = 243 x 9 P: the grid path of atsc_large_fast.h)
    large_sizes = [8192, 16384, 32768, 65536, 131072] if os.environ.get("FUZZ_LARGE") == "2" else [4097, 5000, 6561, 8192, 12000, 20000, 32768, 40000]
    for _ in range(40 if large else 200):
        n = int(rng.choice(large_sizes)) if large else 256 if uniform else int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), 256, 128, 512, 1024, 2048, 4096,
                                                rng.integers(600, 4097)], p=[0.1, 0.35, 0.15, 0.05, 0.05, 0.05, 0.05, 0.05, 0.15]))
        xs.append(_fuzz_frame(rng, n))
        offs.append(offs[-1] + n)
    x = np.concatenate(xs)
    off = np.array(offs, dtype=np.uint64)
    comp, bounded = [(A.AUTO, True), (A.AUTO, True), (A.FFT, True), (A.POLYNOMIAL, True), (A.RLE, False)][seed % 5]
    s = P.compare_batch(orc, ctx, x, off, comp, bounded, me)
    for k in ("exact", "tol", "tie", "boundary"):
        tot[k] += s[k]
    tot["frames"] += len(offs) - 1
    line = "seed %d e=%d comp=%d uniform=%d exact=%d tol=%d tie=%d boundary=%d fail=%d" % (
        seed, e, comp, uniform, s["exact"], s["tol"], s["tie"], s["boundary"], len(s["fail"]))
    print(line, flush=True)
    for f in s["fail"][:5]:
        print("   FAIL frame %d n=%d: %s" % (f[0], int(off[f[0] + 1] - off[f[0]]), f[1]), flush=True)
    bad += len(s["fail"])
print("TOTAL", tot, "failures", bad)
