"""Decompress throughput by frame length on the mixed synthetic workload, device resident (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

N = 40960 * 256
me = float(np.float32(float(os.environ.get("ERR", "5"))) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(0, N)
d_x = torch.from_numpy(x).to(dev)
for F in (64, 128, 256, 300, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    off = H.frame_offsets(N, F)
    plan = ctx.plan(off)
    outs = plan.alloc_outputs(torch, dev)
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    nb = int(outs["rec_off"][-1].item())
    dp = atsc_amd.DPlan(ctx, outs["body"][:nb].cpu().numpy())
    d_out = torch.empty(N, dtype=torch.float64, device=dev)
    for _ in range(2):
        dp.decompress(outs["body"], d_out, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dp.decompress(outs["body"], d_out, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("frame %6d  frames %6d  %8.3f ms  %7.2f Gsamples/s" % (F, len(off) - 1, dt * 1e3, N / dt / 1e9), flush=True)
    dp.close(); plan.close()
    del outs, d_out
