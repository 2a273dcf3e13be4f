"""Does feeding the frame kernel from two hardware queues fill the slots quicker (dev aid, GPU box only)?
The same 40960 frames of 256 samples as one launch, and as 2 / 4 plans of every 2nd / 4th run of frames on as many streams."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

N = 40960 * 256
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
x = H.synth_series(0, N)
d_x = torch.from_numpy(x).to(dev)
for parts in (1, 2, 4):
    n_part = N // parts
    plans, outs, streams, views = [], [], [], []
    for p in range(parts):
        off = H.frame_offsets(n_part, 256)
        plan = ctx.plan(off)
        plans.append(plan)
        outs.append(plan.alloc_outputs(torch, dev))
        streams.append(torch.cuda.Stream())
        views.append(d_x[p * n_part:(p + 1) * n_part])
    def run():
        for p in range(parts):
            plans[p].compress(views[p], outs[p], atsc_amd.AUTO, True, me, 0, streams[p].cuda_stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%d stream(s): %.1f us per 10.5 M samples  %.1f Gsamples/s" % (parts, dt * 1e6, N / dt / 1e9), flush=True)
