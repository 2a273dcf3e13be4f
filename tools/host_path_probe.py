"""Host-pointer entry points on registered caller memory: time per call against the copy piece size and the
number of parts (ATSC_H2D_PIECE_MB / ATSC_HOST_PARTS).  GPU box only."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import atsc_amd  # noqa: E402
from tests.helpers import synth_series  # noqa: E402

FRAME = 256
N = 10485760


def main():
    lib = atsc_amd.capi.lib()
    ctx = atsc_amd.Context(0)
    x = np.ascontiguousarray(synth_series(0, N), dtype=np.float64)
    off = np.arange(0, N + 1, FRAME, dtype=np.uint64)
    nf = len(off) - 1
    cap = int(nf * (32 + 14 * FRAME + 16))
    body = np.empty(cap, dtype=np.uint8)
    blen = C.c_uint64()

    def cf():
        rc = lib.atsc_compress_frames(ctx._h, x.ctypes.data_as(C.POINTER(C.c_double)),
                                      off.ctypes.data_as(C.POINTER(C.c_uint64)), nf, atsc_amd.AUTO, 1,
                                      C.c_float(np.float32(0.05)), 0, body.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
                                      C.byref(blen), None, None, None)
        atsc_amd.capi.check(rc, ctx._h)

    def timed(fn, reps=7):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    for reg in (False, True):
        if reg:
            atsc_amd.capi.check(lib.atsc_host_register(C.c_void_p(x.ctypes.data), x.nbytes))
            atsc_amd.capi.check(lib.atsc_host_register(C.c_void_p(body.ctypes.data), body.nbytes))
        for piece in (2, 16, 64, 128):
            for parts in (1, 5, 8, 16):
                os.environ["ATSC_H2D_PIECE_MB"] = str(piece)
                os.environ["ATSC_HOST_PARTS"] = str(parts)
                ms = timed(cf)
                print("registered %d  piece %3d MB  parts %2d   %.3f ms  %.2f Gsamples/s" % (reg, piece, parts, ms, N / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
