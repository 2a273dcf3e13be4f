"""Times k_compress per synthetic class and per forced codec (HIP events around the kernel),
to see where a frame's time goes.  Dev aid; run on the GPU box."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atsc_amd
from tests import helpers as H

def main():
    n = 40960 * 256 // 4
    frame = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    me = float(np.float32(5) / np.float32(100))
    dev = torch.device("cuda:0")
    ctx = atsc_amd.Context(0)
    off = H.frame_offsets(n, frame)
    plan = ctx.plan(off)
    outs = plan.alloc_outputs(torch, dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for klass in (0, 1, 2, 3, 4, None):
        x = H.synth_series(0, n, klass=klass)
        d_x = torch.from_numpy(x).to(dev)
        for mode, name, bounded in ((atsc_amd.AUTO, "auto", True), (atsc_amd.FFT, "fft", True),
                                    (atsc_amd.POLYNOMIAL, "poly", True), (atsc_amd.RLE, "rle", False),
                                    (atsc_amd.CONSTANT, "const", False)):
            for _ in range(2):
                plan.compress(d_x, outs, mode, bounded, me, 0, st)
            torch.cuda.synchronize()
            ctx.set_profiling(True)
            for _ in range(5):
                plan.compress(d_x, outs, mode, bounded, me, 0, st)
            torch.cuda.synchronize()
            ms, cnt = ctx.profile_read()
            ctx.set_profiling(False)
            us_per_kframe = ms / cnt * 1e3 / (plan.n_frames / 1000.0)
            res["class%s/%s" % (klass, name)] = round(ms / cnt * 1e3, 1)
    print(json.dumps({"frame": frame, "frames": plan.n_frames, "kernel_us": res}, indent=1))

main()
