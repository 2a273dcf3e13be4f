"""Where a wavefront of the headline kernel spends its cycles (dev aid, GPU box only).  Needs the
instrumented library: ATSC_BUILD_VARIANT=stamps ATSC_BUILD_DEFS=-DATSC_STAMPS python -m atsc_amd.build, then
ATSC_LIB_VARIANT=stamps python tools/stamp_probe.py.  Prints shader-clock cycles per frame and phase
(k_compress<1,5,false,256>'s PH() stamps, summed by lane 0 over every frame of a launch)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ATSC_LIB_VARIANT", "stamps")
import torch, atsc_amd
from atsc_amd import capi
from tests import helpers as H

NAMES = ["load samples", "stats", "const/noop/rle bound", "g, 1/|g| registers", "poly (if first)", "tw load + fwd FFT",
         "norms, zero cut", "ladder: admit bins", "ladder: evaluate + sum", "ladder exit", "poly (if second)",
         "pending RLE sizing", "select + emit"]
F = 256
n = 40960 * F
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
off = H.frame_offsets(n, F)
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
st = torch.cuda.current_stream().cuda_stream
L = capi.lib()
L.atsc_dev_phase_read.restype = C.c_int
L.atsc_dev_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
for klass in (None, 0, 1, 2, 3):
    x = H.synth_series(0, n, klass=klass)
    d_x = torch.from_numpy(x).to(dev)
    for _ in range(2):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    assert L.atsc_dev_phase_read(buf, 1) == 0
    reps = 4
    for _ in range(reps):
        plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
    torch.cuda.synchronize()
    assert L.atsc_dev_phase_read(buf, 1) == 0
    cyc = [buf[i] / (reps * 40960.0) for i in range(len(NAMES))]
    tot = sum(cyc)
    print("class %s: %.0f cycles per frame" % (klass, tot))
    for nm, c in zip(NAMES, cyc):
        print("   %-26s %8.0f  %5.1f %%" % (nm, c, 100.0 * c / tot))

# occupancy over time of the last launch: frames in flight at every 1 us tick (wall clock, 100 MHz)
L.atsc_dev_span_read.restype = C.c_int
L.atsc_dev_span_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint]
x = H.synth_series(0, n, klass=None)
d_x = torch.from_numpy(x).to(dev)
for _ in range(3):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
sp = (C.c_ulonglong * (2 * 40960))()
assert L.atsc_dev_span_read(sp, 40960) == 0
a = np.frombuffer(sp, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
t0 = a[:, 0].min()
beg, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print("launch spans %.1f us; frame life mean %.2f us, p50 %.2f, p99 %.2f, max %.2f" % (
    end.max(), (end - beg).mean(), np.median(end - beg), np.percentile(end - beg, 99), (end - beg).max()))
print("slot-time used: %.1f us x 5888 slots" % ((end - beg).sum() / 5888))
for t in range(0, int(end.max()) + 1, 4):
    live = int(((beg <= t) & (end > t)).sum())
    started = int((beg <= t).sum())
    print("  t=%3d us  in flight %5d  started %5d" % (t, live, started))
