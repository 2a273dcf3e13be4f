import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = int(os.environ.get('NF', '16')); n = nf * F
kl = os.environ.get('KLASS', '0'); kl = None if kl == 'mix' else int(kl)
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=kl)
off = H.frame_offsets(n, F)
rec, _, chosen, _ = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
dp = atsc_amd.DPlan(ctx, rec)
d_body = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
d_out = torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(3):
    dp.decompress(d_body, d_out, st)
torch.cuda.synchronize()
