"""Cumulative phase timing of k_compress through ATSC_DEBUG_STOP (dev aid, GPU box only)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, json
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
n = 40960*256; me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, 256); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for klass in (0, 1, None):
    x = H.synth_series(0, n, klass=klass); d_x = torch.from_numpy(x).to(dev)
    for _ in range(2): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize(); ctx.set_profiling(True)
    for _ in range(5): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize(); ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
    res[str(klass)] = round(ms/cnt*1e3, 1)
print(json.dumps(res))
''' % ROOT
out = {}
for stop in (1, 2, 3, 4, 5, 6, 7, 8, 0):
    env = dict(os.environ, ATSC_DEBUG_STOP=str(stop))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out[stop] = json.loads(line[-1]) if line else r.stderr[-300:]
names = {1: "load+stats", 2: "+const/noop checks", 3: "+g/inv regs", 4: "+fwd FFT", 5: "+norms/Z",
         6: "+ladder", 7: "+poly", 8: "+rle", 0: "all (+select/emit)"}
for k in (1, 2, 3, 4, 5, 6, 7, 8, 0):
    print("%-22s %s" % (names[k], out[k]))
