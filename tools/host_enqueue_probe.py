"""How long does the HOST take to enqueue one compress call (no synchronisation inside the loop)?  Compared with the
GPU's time per batch this tells whether a loop of small batches is bound by the host's launch calls.
(dev aid, GPU box only)   env: FLEN (256), NF frames (40960), PIPE=1 pipelined"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

F = int(os.environ.get("FLEN", "256")); nf = int(os.environ.get("NF", "40960")); n = nf * F
pipe = os.environ.get("PIPE", "1") == "1"
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=None)
off = H.frame_offsets(n, F)
plan = ctx.plan(off)
outs = [plan.alloc_outputs(torch, dev) for _ in range(4)]
d_x = torch.from_numpy(x).to(dev)
for i in range(8):
    plan.compress(d_x, outs[i % 4], atsc_amd.AUTO, True, me, 0, st, pipelined=pipe)
plan.join(st); torch.cuda.synchronize()
K = 12
ts = []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    plan.compress(d_x, outs[i % 4], atsc_amd.AUTO, True, me, 0, st, pipelined=pipe)
    ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
plan.join(st); torch.cuda.synchronize()
t2 = time.perf_counter()
print("frames %d x %d pipelined=%s chains=%s: host enqueue per call: median %.1f us, min %.1f, max %.1f; loop %.1f us/call enqueue-only, %.1f us/call with the final sync" % (
    nf, F, pipe, os.environ.get("ATSC_CHAINS", "4"), 1e6 * float(np.median(ts)), 1e6 * min(ts), 1e6 * max(ts), 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K))
