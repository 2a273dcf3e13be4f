"""k_decompress time per synthetic class, uniform batches of 40960 x 256 samples (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
F = int(os.environ.get("FLEN", "256")); nf = 10485760 // F; n = nf * F
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
off = H.frame_offsets(n, F)
for klass in (0, 1, 2, 3, 4, None):
    x = H.synth_series(3, n, klass=klass)
    rec, _, chosen, _ = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
    dp = atsc_amd.DPlan(ctx, rec)
    d_body = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    u, c = np.unique(chosen, return_counts=True)
    print("class %-4s decode %7.1f us  %7.2f Gsamples/s  bytes %9d  codecs %s" % (klass, dt * 1e6, n / dt / 1e9, len(rec), dict(zip(u.tolist(), c.tolist()))), flush=True)
    dp.close()
