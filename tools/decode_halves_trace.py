"""atsc_decompress_frames into registered memory (10 M samples, 256-sample frames, mixed classes) with the host phases
on stderr (ATSC_TRACE_HOST=1): where the two-halves form spends its time.   usage (GPU box): python tools/decode_halves_trace.py"""
import os, sys, time, ctypes as C
import numpy as np
os.environ.setdefault("ATSC_TRACE_HOST", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, atsc_amd
from tests import helpers as H
lib = atsc_amd.capi.lib()
F = 256; nf = 40960; n = nf * F
me = float(np.float32(5) / np.float32(100))
ctx = atsc_amd.Context(0)
x = H.synth_series(3, n); off = H.frame_offsets(n, F)
rec, _, _, _ = ctx.compress_host(x, off, atsc_amd.AUTO, True, me, 0)
recarr = np.frombuffer(bytes(rec), dtype=np.uint8).copy()
dec = np.empty(n, dtype=np.float64); on = C.c_uint64()
for a in (recarr, dec):
    atsc_amd.capi.check(lib.atsc_host_register(C.c_void_p(a.ctypes.data), a.nbytes), ctx._h)
def df():
    rc = lib.atsc_decompress_frames(ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(recarr), 0,
                                    dec.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(on))
    atsc_amd.capi.check(rc, ctx._h)
for i in range(4):
    sys.stderr.write("---- call %d\n" % i); sys.stderr.flush()
    t0 = time.perf_counter(); df(); dt = time.perf_counter() - t0
    sys.stderr.write("call %d: %.3f ms\n" % (i, dt * 1e3)); sys.stderr.flush()
for a in (recarr, dec):
    lib.atsc_host_unregister(C.c_void_p(a.ctypes.data))
