"""k_compress time per synthetic class (uniform batches of 40960 x 256) and for the mixed workload
(dev aid, GPU box only).  Prints microseconds per launch and the codec histogram."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H

n = 40960 * 256
e = int(sys.argv[1]) if len(sys.argv) > 1 else 5
me = float(np.float32(e) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
off = H.frame_offsets(n, 256)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
for klass in (0, 1, 2, 3, 4, None):
    x = H.synth_series(0, n, klass=klass)
    d_x = torch.from_numpy(x).to(dev)
    for _ in range(2):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    for _ in range(5):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    ms, cnt = ctx.profile_read()
    ctx.set_profiling(False)
    ch = outs["chosen"].cpu().numpy()
    hist = {atsc_amd.capi.COMPRESSOR_NAMES[int(c)]: int(np.sum(ch == c)) for c in np.unique(ch)}
    print("class %-5s %7.1f us/launch  %s  bytes %d" % (klass, ms / cnt * 1e3, hist, int(outs["rec_off"][-1].item())), flush=True)
