"""Which large frames the grid path leaves to the general kernel, and why (ATSC_DEBUG_STOP=-3 makes k_compress_large<0>
print FastState's reason code): 1 geometry / carve-up, 2 NaN first sample, 3 a zero extreme, 4 polynomial step, 5 flat in
f32, 6 fewer than 8 bins, 7 RLE may beat a passing polynomial, 8-10 select, 11 FFT ladder goes on, 12 polynomial ladder
goes on, 13 RLE can still win, 14 nothing passed.   usage (GPU box): NF=256 FLEN=131072 python tools/fast_left_probe.py"""
import os, sys, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = int(os.environ.get("FLEN", "131072")); nf = int(os.environ.get("NF", "256")); n = nf * F
me = float(np.float32(float(os.environ.get("ERR", "5")))/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(int(os.environ.get("SEED", "3")), n); d_x = torch.from_numpy(x).to(dev)
plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st); torch.cuda.synchronize()
ch = outs["chosen"].cpu().numpy()
print("CHOSEN", " ".join(str(int(c)) for c in ch), flush=True)
''' % ROOT
env = dict(os.environ, ATSC_DEBUG_STOP="-3")
r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
left = {}
chosen = []
for l in r.stdout.splitlines():
    if l.startswith("FASTLEFT"):
        w = l.split(); left[int(w[2])] = int(w[4])
        if os.environ.get("RAW"): print(l)
    elif l.startswith("SELECT"):
        print(l)
    elif l.startswith("CHOSEN"):
        chosen = [int(c) for c in l.split()[1:]]
if not chosen:
    print(r.stderr[-500:])
print("frames", len(chosen), "left to the general kernel", len(left))
by = collections.Counter((why, chosen[f]) for f, why in left.items())
for (why, c), k in sorted(by.items()):
    print("  why %2d  codec chosen %2d  frames %d" % (why, c, k))
print("  frames:", sorted(left)[:40])
