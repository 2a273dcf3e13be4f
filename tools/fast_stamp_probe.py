"""Phase times (us) of k_large_decide1's workgroup 0 (ATSC_DEBUG_STOP=-3 makes it print them): per class of frame.
usage (GPU box): NF=80 python tools/fast_stamp_probe.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = int(os.environ.get("NF", "80")); n = nf * F
klass = os.environ["KLASS"]; klass = None if klass == "mix" else int(klass)
me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=klass); d_x = torch.from_numpy(x).to(dev)
for _ in range(3):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st); torch.cuda.synchronize()
print("CODEC", int(outs["chosen"][0].item()), flush=True)
''' % ROOT
print("phases: 1 stats/poly/bounds, 2 histogram, 3 digit + collection, 4 radix select, 5 admission, 6 bucketing, 7 list + state out")
for klass in ("0", "1", "2", "3", "mix"):
    env = dict(os.environ, ATSC_DEBUG_STOP="-3", KLASS=klass)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("FSTAMP") or l.startswith("CODEC")]
    print("class", klass, "|", " | ".join(lines[-2:]) if lines else r.stderr[-300:])
