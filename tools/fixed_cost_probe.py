"""Where the fixed cost of bench.py's chained loop comes from (20 timed steps give 88 Gsamples/s,
200 give 102).  Times loops of K = 5 .. 400 steps back to back (same plan, same rotation as bench.py), twice each, and
the per-call host enqueue time.   usage (GPU box): python tools/fixed_cost_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atsc_amd
from tests import helpers as H

N, F, R = 10485760, 256, 5
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
if os.environ.get("CHAINS"):
    ctx.set_chains(int(os.environ["CHAINS"]))
me = float(np.float32(5) / np.float32(100))
d_xs = [torch.from_numpy(H.synth_series(b, N, class_shift=2 * b)).to(dev) for b in range(R)]
off = np.arange(0, N + 1, F, dtype=np.uint64)
plan = ctx.plan(off)
st = torch.cuda.current_stream().cuda_stream
outs = [plan.alloc_outputs(torch, dev) for _ in range(8)]


def loop(K, pipe=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enq = []
    for i in range(K):
        a = time.perf_counter()
        plan.compress(d_xs[i % R], outs[i % 8], atsc_amd.AUTO, True, me, 0, st, pipelined=pipe)
        enq.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    if pipe:
        plan.join(st)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e3, (t1 - t0) * 1e3, float(np.median(enq)) * 1e6, float(np.max(enq)) * 1e6


for i in range(8):
    plan.compress(d_xs[i % R], outs[i % 8], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
plan.join(st)
torch.cuda.synchronize()
print("pipelined loops (total ms, ms until the last call returned, us per call enqueue median / max)")
for rep in range(2):
    for K in (5, 10, 20, 20, 20, 50, 100, 200, 400):
        tot, sub, med, mx = loop(K)
        print("  K %4d  total %8.3f ms  = %7.2f us/step  (%6.1f Gsamples/s)  submitted after %8.3f ms  enqueue %5.1f / %6.1f us"
              % (K, tot, tot / K * 1e3, N * K / tot / 1e6, sub, med, mx), flush=True)
idle = float(os.environ.get("IDLE", "0.5"))
for gap in (0.0, 0.01, 0.1, idle):
    time.sleep(gap)
    tot, sub, med, mx = loop(20)
    print("  after %.2f s idle: K 20 total %.3f ms = %.2f us/step" % (gap, tot, tot / 20 * 1e3), flush=True)
print("plain single-stream loops")
for K in (5, 20, 20, 100):
    tot, sub, med, mx = loop(K, False)
    print("  K %4d  total %8.3f ms  = %7.2f us/step  enqueue %5.1f / %6.1f us" % (K, tot, tot / K * 1e3, med, mx), flush=True)
