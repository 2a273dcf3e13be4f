"""Where a sparse-inverse ladder trip spends its time (dev aid): ATSC_DEBUG_STOP = 16 + bits, bit 0 skips
the direct sum, bit 1 the LDS transforms, bit 2 the evaluation; all runs stop after the FFT ladder."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, json, time
import numpy as np
sys.path.insert(0, %r)
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = int(os.environ.get("NF", "256")); n = nf * F
me = float(np.float32(5)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); off = H.frame_offsets(n, F); plan = ctx.plan(off); outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
x = H.synth_series(3, n, klass=0); d_x = torch.from_numpy(x).to(dev)
plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize(); print(json.dumps(round((time.perf_counter() - t0) / 3 * 1e6, 0)))
''' % ROOT
for stop, name in [(5, "before the ladder"), (6, "ladder"), (16, "ladder, same stop path"), (17, "- direct sum"), (18, "- LDS transforms"),
                   (20, "- evaluation"), (23, "- all three (bucketing only; one trip)")]:
    env = dict(os.environ, ATSC_DEBUG_STOP=str(stop))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("%-40s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
