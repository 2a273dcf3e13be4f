import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
N = 2560 * 4096
F = int(os.environ.get("F", "4096"))
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
for klass in (0, 1, 2, 3, None):
    x = H.synth_series(0, N, klass=klass)
    d_x = torch.from_numpy(x).to(dev)
    off = H.frame_offsets(N, F)
    plan = ctx.plan(off)
    outs = plan.alloc_outputs(torch, dev)
    for _ in range(2):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("class %s frame %d: %.3f ms %.1f Gsamples/s" % (klass, F, dt * 1e3, N / dt / 1e9), flush=True)
    del plan, outs
