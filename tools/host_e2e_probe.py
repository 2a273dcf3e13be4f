"""End-to-end rate of the host-pointer entry points (plan + H2D + compress + D2H, synchronous):
what the CLI and a host binding see (dev aid, GPU box only)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atsc_amd
from tests import helpers as H

ctx = atsc_amd.Context(0)
me = float(np.float32(5) / np.float32(100))
N = 40960 * 256
x = H.synth_series(0, N)
for F in (256, 131072):
    off = H.frame_offsets(N, F)
    ctx.compress_host(x[: F * 8], off[:9], atsc_amd.AUTO, True, me, 0)
    for rep in range(3):
        t0 = time.perf_counter()
        rec, rec_off, chosen, err = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        out = ctx.decompress_host(rec)
        dt2 = time.perf_counter() - t1
        print("frame %6d: compress_host %7.2f ms (%6.2f Gsamples/s, %d bytes)   decompress_host %7.2f ms (%6.2f Gsamples/s)" % (
            F, dt * 1e3, N / dt / 1e9, len(rec), dt2 * 1e3, N / dt2 / 1e9), flush=True)
t0 = time.perf_counter()
bro = atsc_amd.compress_data(ctx, x, atsc_amd.AUTO, 5)
dt = time.perf_counter() - t0
t1 = time.perf_counter()
y = atsc_amd.decompress_data(ctx, bro)
dt2 = time.perf_counter() - t1
print("compress_data (reference chunker) %7.2f ms (%6.2f Gsamples/s)   decompress_data %7.2f ms (%6.2f Gsamples/s)" % (
    dt * 1e3, N / dt / 1e9, dt2 * 1e3, N / dt2 / 1e9))
