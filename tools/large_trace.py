import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
F = int(os.environ.get("FLEN", "131072")); nf = int(os.environ.get("NF", "80")); n = nf * F
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=None)
off = H.frame_offsets(n, F)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
d_x = torch.from_numpy(x).to(dev)
for _ in range(3):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
