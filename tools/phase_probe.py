"""Cumulative phase timing of k_compress<1,5> per synthetic class through ATSC_DEBUG_STOP (dev aid,
GPU box only).  The stops cut the kernel short, so later phases run unpruned by what was skipped:
read the differences as indications, not as an exact budget."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, json
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = int(os.environ.get("FLEN", "256")); n = 40960*256; me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for klass in (0, 1, 2, 3, None):
    x = H.synth_series(0, n, klass=klass); d_x = torch.from_numpy(x).to(dev)
    for _ in range(2): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize(); ctx.set_profiling(True)
    for _ in range(4): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize(); ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
    res[str(klass)] = round(ms/cnt*1e3, 1)
print(json.dumps(res))
''' % ROOT
names = [(1, "load+stats"), (3, "+const/noop/trial checks"), (10, "+RLE bound / early sizing"),
         (11, "+g, 1/|g| registers"), (13, "+poly if first"), (4, "+tw load, fwd FFT"), (5, "+norms, zero cut"),
         (14, "+FFT ladder"), (15, "+poly if second"), (8, "+pending RLE sizing"), (0, "all (+select, emit)")]
print("%-28s %s" % ("us per batch of 10.5 M samples, frame length %s" % os.environ.get("FLEN", "256"), "class 0 / 1 / 2 / 3 / mixed"))
for stop, name in names:
    env = dict(os.environ, ATSC_DEBUG_STOP=str(stop))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else None
    print("%-28s %s" % (name, "  ".join("%7.1f" % d[k] for k in ("0", "1", "2", "3", "None")) if d else r.stderr[-300:]), flush=True)
