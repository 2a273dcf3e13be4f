import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import atsc_amd
from tests import helpers as H
n = 10485760
x = H.synth_series(0, n)
ctx = atsc_amd.Context(0)
bro = atsc_amd.compress_data(ctx, x, atsc_amd.AUTO, 5)
for i in range(3):
    t0 = time.perf_counter(); out = atsc_amd.decompress_data(ctx, bro); print("decompress_data %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
rec = ctx.compress_host(x, H.frame_offsets(n, 256), atsc_amd.AUTO, True, 0.05, 0)[0]
os.environ["X"] = "1"
print("---- F256 decompress_host", file=sys.stderr)
for i in range(2):
    t0 = time.perf_counter(); out = ctx.decompress_host(rec); print("decompress_host %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
