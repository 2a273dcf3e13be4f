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
F = int(os.environ.get("FLEN", "256"))  # 128, 256, 512, 1024, 2048, 4096: the fixed-length kernels
n = 40960 * 256
NF = n // F
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
for klass in ((None, 0, 1, 2, 3) if F == 256 else (None,)):
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
    cyc = [buf[i] / (reps * float(NF)) for i in range(len(NAMES))]
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
sp = (C.c_ulonglong * (3 * min(NF, 65536)))()
assert L.atsc_dev_span_read(sp, min(NF, 65536)) == 0
a = np.frombuffer(sp, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
ok = a[:, 1] > a[:, 0]  # (Constant frames leave before the end stamp)
a = a[ok]
t0 = a[:, 0].min()
beg, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
hw = a[:, 2]
if os.environ.get("DUMP_LONG"):
    idx = np.nonzero(ok)[0]
    life = end - beg
    o = np.argsort(-life)[:40]
    for k in o:
        print("   long frame: launch slot %6d  class block %d  start %7.1f  end %7.1f  life %6.1f  cu %d" % (
            idx[k], (idx[k] // 256) % 5, beg[k], end[k], life[k], (int(hw[k]) >> 8) & 0xF))
print("launch spans %.1f us; %d frames with both stamps; frame life mean %.2f us, p50 %.2f, p99 %.2f, max %.2f" % (
    end.max(), len(a), (end - beg).mean(), np.median(end - beg), np.percentile(end - beg, 99), (end - beg).max()))
for t in range(0, int(end.max()) + 1, 8):
    print("  t=%3d us  in flight %5d  started %5d" % (t, int(((beg <= t) & (end > t)).sum()), int((beg <= t).sum())))
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC_ID 3:0 of the high word
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (((hw >> 32) & 0xF) << 8)
slot = cu * 64 + ((hw >> 4) & 3) * 16 + (hw & 0xF)
print("distinct CUs %d, distinct (CU, SIMD, wave slot) %d" % (len(np.unique(cu)), len(np.unique(slot))))
mid = (beg > 20) & (end < 90)
# frames in flight per CU in the steady state, sampled every us
per_cu = []
for t in range(30, 80, 5):
    live = (beg <= t) & (end > t)
    c = np.bincount(np.unique(cu, return_inverse=True)[1][live], minlength=len(np.unique(cu)))
    per_cu.append(c)
per_cu = np.array(per_cu)
print("frames in flight per CU (steady state): mean %.2f  min %d  max %d" % (per_cu.mean(), per_cu.min(), per_cu.max()))
# gap between a wave slot's consecutive frames
gaps = []
order = np.lexsort((beg, slot))
s_sorted, b_sorted, e_sorted = slot[order], beg[order], end[order]
same = s_sorted[1:] == s_sorted[:-1]
g = (b_sorted[1:] - e_sorted[:-1])[same]
g = g[(b_sorted[1:][same] > 20) & (b_sorted[1:][same] < 90)]
print("gap between consecutive frames of one wave slot: mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f  (n=%d)" % (
    g.mean(), np.percentile(g, 10), np.median(g), np.percentile(g, 90), len(g)))
