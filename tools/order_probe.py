"""Does frame order matter?  Same 40960 frames of the mixed workload, three block orders (dev aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

n = 40960 * 256
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
off = H.frame_offsets(n, 256)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(0, n).reshape(160, 65536)
cost = {0: 2, 1: 0, 2: 1, 3: 3, 4: 4}  # class -> rank, most expensive first
orders = {
    "natural (classes cycle per block)": np.arange(160),
    "expensive classes first": np.array(sorted(range(160), key=lambda b: (cost[b % 5], b))),
    "cheap classes first": np.array(sorted(range(160), key=lambda b: (-cost[b % 5], b))),
}
for name, perm in orders.items():
    d_x = torch.from_numpy(np.ascontiguousarray(x[perm]).reshape(-1)).to(dev)
    for _ in range(2):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    for _ in range(6):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    ms, cnt = ctx.profile_read()
    ctx.set_profiling(False)
    print("%-36s %7.1f us/launch  bytes %d" % (name, ms / cnt * 1e3, int(outs["rec_off"][-1].item())), flush=True)
