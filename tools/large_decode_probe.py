"""Large tier decode: time against the number of 131072-sample frames (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H
F = 131072
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
for nf, klass in ((16, 0), (16, 1), (16, 2), (16, 3), (16, None), (80, None), (256, None), (512, None)):
    n = nf * F
    x = H.synth_series(3, n, klass=klass)
    off = H.frame_offsets(n, F)
    rec, _, chosen, _ = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
    dp = atsc_amd.DPlan(ctx, rec)
    d_body = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("class %-4s frames %4d  decode %8.2f ms  %7.2f Gsamples/s  codecs %s" % (
        klass, nf, dt * 1e3, n / dt / 1e9, {int(c): int(np.sum(chosen == c)) for c in np.unique(chosen)}), flush=True)
    del dp, d_body, d_out
