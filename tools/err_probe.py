"""Why does a frame's reported FFT error differ from the oracle's?  (dev aid, GPU box only)
usage: err_probe.py SEED FRAME [FRAME ...] -- rebuilds fuzz_soak's batch for SEED and, per frame, compares
the reported errors with the MAPE of each payload decoded by the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atsc_amd as A
from oracle import oracle as orc
from tests import helpers as H
from tests.test_gpu_parity import _fuzz_frame

orc.build()
ctx = A.Context(0)
seed = int(sys.argv[1]); want = [int(a) for a in sys.argv[2:]]
rng = np.random.default_rng(seed)
e = int(rng.choice([0, 1, 2, 3, 5, 10, 20, 50])); me = float(np.float32(e) / np.float32(100))
uniform = seed % 3 == 0
xs, offs = [], [0]
for _ in range(200):
    n = 256 if uniform else int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), 256, 128, 512, 1024, 2048, 4096,
                                            rng.integers(600, 4097)], p=[0.1, 0.35, 0.15, 0.05, 0.05, 0.05, 0.05, 0.05, 0.15]))
    xs.append(_fuzz_frame(rng, n)); offs.append(offs[-1] + n)
x = np.concatenate(xs); off = np.array(offs, dtype=np.uint64)
comp, bounded = [(A.AUTO, True), (A.AUTO, True), (A.FFT, True), (A.POLYNOMIAL, True), (A.RLE, False)][seed % 5]
rec, rec_off, chosen, err = ctx.compress_host(x, off, comp, bounded, me, 0)
frames = H.parse_bro_body(rec, with_count=False)
print("seed", seed, "e", e, "comp", comp)
for i in want:
    fx = x[int(off[i]):int(off[i + 1])]; n = len(fx)
    fs, sc, tag, payload = frames[i]
    if comp == A.AUTO:
        po, cho, eo = orc.compress_best(fx, me, 0)
    else:
        po, eo = orc.compress(comp, fx, bounded, me); cho = comp
    dg = np.array(orc.decompress(tag, payload, n)); do = np.array(orc.decompress(cho, po, n))
    with np.errstate(all="ignore"):
        tg = np.abs((dg - fx) / fx); to = np.abs((do - fx) / fx)
    a = np.abs(fx); nz = a[(a > 0) & np.isfinite(a)]
    print("frame %d n=%d tag gpu/oracle %d/%d  payload bytes %d/%d  min|g| %.3g  max|g| %.3g" % (i, n, tag, cho, len(payload), len(po), nz.min() if nz.size else 0, a.max()))
    print("   reported err gpu %.9g  oracle %.9g   (MAPE over the unpadded decode: gpu payload %.9g, oracle payload %.9g)" % (err[i], eo, tg.mean(), to.mean()))
    d = np.nonzero(dg != do)[0]
    print("   decoded samples that differ: %d of %d" % (len(d), n))
    for j in d[:12]:
        print("      j=%d g=%.9g  gpu %.9g  oracle %.9g   term gpu %.6g oracle %.6g  (delta/n %.3g)" % (j, fx[j], dg[j], do[j], tg[j], to[j], (tg[j] - to[j]) / n))
