"""Phase table of k_compress_large's workgroup 0 from the LTSTAMP lines (ATSC_DEBUG_STOP=-1): microseconds between
the stamps, per class of frame.  usage (GPU box): NF=80 python tools/large_stamp_probe.py"""
import os, sys, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = int(os.environ.get("NF", "80")); n = nf * F
klass = int(os.environ["KLASS"])
me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=klass); d_x = torch.from_numpy(x).to(dev)
for _ in range(2):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st); torch.cuda.synchronize()
print("CODEC", int(outs["chosen"][0].item()), flush=True)
''' % ROOT
for klass in (0, 1, 2, 3):
    env = dict(os.environ, ATSC_DEBUG_STOP="-1", KLASS=str(klass))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("LTSTAMP")]
    codec = [l for l in r.stdout.splitlines() if l.startswith("CODEC")]
    # last launch only: keep the stamps after the last "part? start" of the lowest part number
    runs, cur = [], []
    for l in lines:
        if l[2] == "start" and l[1] in ("part0", "part1") and cur:
            runs.append(cur); cur = []
        cur.append(l)
    if cur: runs.append(cur)
    last = runs[-1] if runs else []
    print("class %d (%s): frame 0" % (klass, codec[-1] if codec else "?"))
    prev = None
    for l in last:
        t = int(l[-1]); name = " ".join(l[2:-1])
        if prev is not None:
            print("   %-6s %-24s %8.1f us" % (l[1], name, (t - prev) / 100.0))
        else:
            print("   %-6s %-24s" % (l[1], name))
        prev = t
    if not last:
        print(r.stdout[-500:], r.stderr[-500:])
