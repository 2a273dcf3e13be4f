"""Large tier, mixed classes: plain and pipelined rates for a few batch sizes (dev aid, GPU box only).
usage: python tools/large_quick.py [e] ; env NFS="16,80,256" FLEN=131072"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

F = int(os.environ.get("FLEN", "131072"))
me = float(np.float32(int(sys.argv[1]) if len(sys.argv) > 1 else 5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
for nf in [int(v) for v in os.environ.get("NFS", "16,80,256").split(",")]:
    n = nf * F
    xs = [H.synth_series(3 + b, n, klass=None) for b in range(2)]
    off = H.frame_offsets(n, F)
    plan = ctx.plan(off)
    outs = [plan.alloc_outputs(torch, dev) for _ in range(4)]
    d_xs = [torch.from_numpy(x).to(dev) for x in xs]
    for i in range(3):
        plan.compress(d_xs[i % 2], outs[0], atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % 2], outs[0], atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ref = [None, None]
    for b in range(2):
        plan.compress(d_xs[b], outs[0], atsc_amd.AUTO, True, me, 0, st)
        torch.cuda.synchronize()
        tot = int(outs[0]["rec_off"][-1].item())
        ref[b] = outs[0]["body"][:tot].cpu().numpy().tobytes()
    for i in range(16):
        plan.compress(d_xs[i % 2], outs[i % 4], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
    plan.join(st)
    torch.cuda.synchronize()
    reps = 16
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % 2], outs[i % 4], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
    plan.join(st)
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / reps
    same = True
    for q in range(4):
        b = (reps - 4 + q) % 2
        tot = int(outs[q]["rec_off"][-1].item())
        same = same and outs[q]["body"][:tot].cpu().numpy().tobytes() == ref[b]
    ch = outs[0]["chosen"].cpu().numpy()
    hist = {atsc_amd.capi.COMPRESSOR_NAMES[int(c)]: int(np.sum(ch == c)) for c in np.unique(ch)}
    print("frames %4d x %6d  plain %7.3f ms %6.2f Gs/s   pipelined %7.3f ms %6.2f Gs/s  bytes_equal %s  ratio %.2f  %s" % (
        nf, F, dt * 1e3, n / dt / 1e9, dtp * 1e3, n / dtp / 1e9, same, 8.0 * n / len(ref[0]), hist), flush=True)
    del plan, outs, d_xs
