"""configs[3]'s shape in the reference chunker's framing (16 series x 262,144 samples, class = series % 5, e = 1 %: 32 frames
of 131072 samples): three compress and three decompress calls, for a rocprofv3 kernel trace (tools/large_c3_kstats.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
F = 131072; NS = int(os.environ.get("NS", "16")); PER = 262144; n = NS * PER
me = float(np.float32(1) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = np.concatenate([H.synth_series(s, PER, klass=s % 5) for s in range(NS)])
off = H.frame_offsets(n, F)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
d_x = torch.from_numpy(x).to(dev)
for _ in range(3):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
nb = int(outs["rec_off"][-1].item())
rec = bytes(outs["body"][:nb].cpu().numpy())
print("chosen", np.bincount(outs["chosen"].cpu().numpy(), minlength=7), "bytes", nb)
dp = atsc_amd.DPlan(ctx, rec)
d_out = torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(3):
    dp.decompress(outs["body"], d_out, st)
torch.cuda.synchronize()
