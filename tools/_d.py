import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, atsc_amd
from tests import helpers as H
F = 131072; nf = 10; n = nf * F
me = float(np.float32(1)/np.float32(100)); dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0); ctx.enable_diag(True)
off = H.frame_offsets(n, F)
x = H.synth_series(3, n, klass=None)
for f in range(nf):
    xf = x[f*F:(f+1)*F]
    o = np.array([0, F], dtype=np.uint64)
    t0 = time.perf_counter()
    rec, ro, ch, err = ctx.compress_host(xf, o, atsc_amd.AUTO, True, me, 0)
    dt = time.perf_counter() - t0
    d = ctx.last_diag(1)[0]
    print("frame %d classes (%d,%d): %.2f ms chosen %d len %d | fft trips %d k %d size %d err %.4f | poly trips %d step %d size %d err %.4f | rle size %d" % (
        f, (2*f) % 5, (2*f+1) % 5, dt*1e3, ch[0], len(rec), d.fft_trips, d.fft_k, d.fft_size, d.fft_err, d.poly_trips, d.poly_step, d.poly_size, d.poly_err, d.rle_size), flush=True)
