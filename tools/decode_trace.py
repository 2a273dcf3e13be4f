"""configs[2] in 256-sample frames: three decompress calls of the GPU's own stream, device resident, for a rocprofv3
kernel trace / PMC pass (tools/collect_profiles_r04.sh: HBM traffic of k_decompress<1,5> per launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
n = 10485760
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(0, n)
off = H.frame_offsets(n, 256)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
d_x = torch.from_numpy(x).to(dev)
plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
nb = int(outs["rec_off"][-1].item())
dp = atsc_amd.DPlan(ctx, outs["body"][:nb].cpu().numpy())
d_out = torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(3):
    dp.decompress(outs["body"], d_out, st)
torch.cuda.synchronize()
print("records", nb)
