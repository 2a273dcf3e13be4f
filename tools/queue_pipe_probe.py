"""Batches of configs[2] dealt round-robin over C independent chains (a plan and a stream each, pipelined entry
point): nothing crosses between the chains' queues (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H
N = 40960 * 256
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
R = 4
d_xs = [torch.from_numpy(np.roll(H.synth_series(0, N), 65536 * b)).to(dev) for b in range(R)]
off = H.frame_offsets(N, 256)
for C in (1, 2, 3, 4):
    for one_ctx in (True, False):
        ctxs = [atsc_amd.Context(0) for _ in range(1 if one_ctx else C)]
        plans = [ctxs[0 if one_ctx else c].plan(off) for c in range(C)]
        outs = [[p.alloc_outputs(torch, dev) for _ in range(2)] for p in plans]
        streams = [torch.cuda.Stream() for _ in range(C)]
        def go(i):
            c = i % C
            plans[c].compress(d_xs[i % R], outs[c][(i // C) % 2], atsc_amd.AUTO, True, me, 0, streams[c].cuda_stream, pipelined=True)
        for i in range(8 * C):
            go(i)
        torch.cuda.synchronize()
        K = 48
        t0 = time.perf_counter()
        for i in range(K):
            go(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("chains %d (%s): %.1f us per batch  %.1f Gsamples/s" % (C, "one context" if one_ctx else "a context each", dt * 1e6, N / dt / 1e9), flush=True)
        del plans, outs, ctxs
