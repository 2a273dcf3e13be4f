import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import atsc_amd
from tests import helpers as H
n = 10485760
x = H.synth_series(0, n)
ctx = atsc_amd.Context(0)
off = H.frame_offsets(n, 256)
for i in range(3):
    t0 = time.perf_counter(); r = ctx.compress_host(x, off, atsc_amd.AUTO, True, 0.05, 0); print("compress_host F256 %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
print("---- compress_data", file=sys.stderr)
for i in range(3):
    t0 = time.perf_counter(); b = atsc_amd.compress_data(ctx, x, atsc_amd.AUTO, 5); print("compress_data %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
