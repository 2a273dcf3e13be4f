"""Cumulative phase timing of k_compress_large through ATSC_DEBUG_STOP (dev aid, GPU box only)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, json, time
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = int(os.environ.get("NF", "16")); n = nf * F
me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for klass in (0, 1, 2, 3):
    x = H.synth_series(3, n, klass=klass); d_x = torch.from_numpy(x).to(dev)
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize(); res[str(klass)] = round((time.perf_counter() - t0) / 3 * 1e6, 0)
print(json.dumps(res))
''' % ROOT
names = [(1, "stats"), (2, "+RLE bound / few runs"), (4, "+fwd FFT, untangle"), (3, "+norms, radix select"),
         (5, "+sort of admitted keys"), (6, "+FFT ladder"), (7, "+poly ladder"), (0, "all (+RLE exact, select, emit, pack)")]
print("us per batch of %s frames x 131072   class 0 / 1 / 2 / 3" % os.environ.get("NF", "16"))
for stop, name in names:
    env = dict(os.environ, ATSC_DEBUG_STOP=str(stop))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else None
    print("%-38s %s" % (name, "  ".join("%8.0f" % d[k] for k in ("0", "1", "2", "3")) if d else r.stderr[-300:]), flush=True)
