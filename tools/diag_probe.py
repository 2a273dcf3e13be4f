"""Ladder statistics per synthetic class from the per-frame diagnostics (dev aid, GPU box only):
how many trips each ladder ran, and what the candidates' sizes were."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

nf = 4096
n = nf * 256
e = int(sys.argv[1]) if len(sys.argv) > 1 else 5
me = float(np.float32(e) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
ctx.enable_diag(True)
off = H.frame_offsets(n, 256)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
NA = 0xFFFFFFFF
for klass in (0, 1, 2, 3):
    x = H.synth_series(0, n, klass=klass)
    d_x = torch.from_numpy(x).to(dev)
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    dg = ctx.last_diag(nf)
    a = np.array([(d.fft_size, d.poly_size, d.rle_size, d.fft_trips, d.fft_k, d.poly_trips, d.poly_step, d.poly_points,
                   d.fft_err, d.poly_err) for d in dg], dtype=np.float64)
    ch = outs["chosen"].cpu().numpy()
    ro = outs["rec_off"].cpu().numpy()
    plen = np.diff(ro)
    print("class", klass, "chosen", {atsc_amd.capi.COMPRESSOR_NAMES[int(c)]: int(np.sum(ch == c)) for c in np.unique(ch)})
    for name, col in (("fft_trips", 3), ("fft_k", 4), ("poly_trips", 5), ("poly_points", 7)):
        v = a[:, col]
        print("   %-11s mean %6.2f  p50 %4.0f  p90 %4.0f  max %4.0f  zero %5d" % (name, v.mean(), np.median(v), np.percentile(v, 90), v.max(), int(np.sum(v == 0))))
    for name, col in (("fft_size", 0), ("poly_size", 1), ("rle_size", 2)):
        v = a[:, col]
        ok = v != NA
        print("   %-11s ran %5d  mean %7.1f  min %5.0f max %6.0f" % (name, int(ok.sum()), v[ok].mean() if ok.any() else -1, v[ok].min() if ok.any() else -1, v[ok].max() if ok.any() else -1))
    print("   record bytes mean %.1f" % plen.mean())
    # how often the ladder that ran first lost
    fft_pass = (a[:, 8] <= me) & (a[:, 0] != NA)
    poly_pass = (a[:, 9] <= me) & (a[:, 1] != NA)
    print("   fft passes %d  poly passes %d  both ran trips %d" % (fft_pass.sum(), poly_pass.sum(), int(np.sum((a[:, 3] > 0) & (a[:, 5] > 0)))))
