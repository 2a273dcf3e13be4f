"""Large tier (131072-sample frames): compress / decompress time against the number of frames in
the batch, per synthetic class (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

F = 131072
me = float(np.float32(int(sys.argv[1]) if len(sys.argv) > 1 else 5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
for klass in (0, 1, 2, 3, None):
    for nf in (16, 80, 256, 512):
        n = nf * F
        x = H.synth_series(3, n, klass=klass)
        off = H.frame_offsets(n, F)
        plan = ctx.plan(off)
        outs = plan.alloc_outputs(torch, dev)
        d_x = torch.from_numpy(x).to(dev)
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ch = outs["chosen"].cpu().numpy()
        hist = {atsc_amd.capi.COMPRESSOR_NAMES[int(c)]: int(np.sum(ch == c)) for c in np.unique(ch)}
        print("class %-5s frames %4d  %8.2f ms  %7.2f Gsamples/s  %6.1f us/frame  %s" % (
            klass, nf, dt * 1e3, n / dt / 1e9, dt / nf * 1e6, hist), flush=True)
        del plan, outs, d_x
