"""Where the host-pointer entry points spend their time (10.5 M samples): pieces timed through the C ABI."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import atsc_amd
from tests import helpers as H
lib = atsc_amd.capi.lib()
n = 10485760
x = H.synth_series(0, n)
ctx = atsc_amd.Context(0)
me = float(np.float32(5) / np.float32(100))
def med(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
sizes = atsc_amd.chunk_sizes(n)
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
print("chunker frames:", len(sizes), sizes[:3], sizes[-3:])
for name, o in (("F256", H.frame_offsets(n, 256)), ("chunker", off), ("F4096", H.frame_offsets(n, 4096))):
    print("compress_host %-8s %.2f ms" % (name, med(lambda: ctx.compress_host(x, o, atsc_amd.AUTO, True, me, 0))), flush=True)
out = np.empty(n)
print("clean_data %.2f ms" % med(lambda: lib.atsc_clean_data(x.ctypes.data_as(C.POINTER(C.c_double)), n, out.ctypes.data_as(C.POINTER(C.c_double)))))
print("compress_data %.2f ms" % med(lambda: atsc_amd.compress_data(ctx, x, atsc_amd.AUTO, 5)), flush=True)
bro = atsc_amd.compress_data(ctx, x, atsc_amd.AUTO, 5)
print("decompress_data %.2f ms" % med(lambda: atsc_amd.decompress_data(ctx, bro)), flush=True)
rec = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)[0]
print("decompress_host chunker %.2f ms" % med(lambda: ctx.decompress_host(rec)), flush=True)
rec = ctx.compress_host(x, H.frame_offsets(n, 256), atsc_amd.AUTO, True, me, 0)[0]
print("decompress_host F256 %.2f ms" % med(lambda: ctx.decompress_host(rec)), flush=True)
