"""C-level timing of atsc_compress_data / atsc_decompress_data (no Python copies in the timed region).  Dev aid."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import atsc_amd
from atsc_amd import capi
from tests import helpers as H
L = capi.lib()
ctx = atsc_amd.Context(0)
n = 10485760
x = H.synth_series(3, n)
xp = x.ctypes.data_as(C.POINTER(C.c_double))
for rep in range(4):
    bro = C.POINTER(C.c_uint8)(); blen = C.c_uint64()
    t0 = time.perf_counter()
    rc = L.atsc_compress_data(ctx._h, xp, n, atsc_amd.AUTO, 5, 0, C.byref(bro), C.byref(blen))
    t1 = time.perf_counter()
    assert rc == 0
    out = C.POINTER(C.c_double)(); on = C.c_uint64()
    t2 = time.perf_counter()
    rc = L.atsc_decompress_data(ctx._h, bro, blen.value, C.byref(out), C.byref(on))
    t3 = time.perf_counter()
    assert rc == 0 and on.value == n
    print("compress_data %6.2f ms (%5.2f Gsamples/s, %d bytes)   decompress_data %6.2f ms (%5.2f Gsamples/s)" % (
        (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, blen.value, (t3 - t2) * 1e3, n / (t3 - t2) / 1e9), flush=True)
    L.atsc_free(out); L.atsc_free(bro)
