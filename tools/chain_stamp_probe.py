"""Frames in flight over time while pipelined calls overlap (dev aid, GPU box only).  Needs the instrumented library
(ATSC_BUILD_VARIANT=stamps ATSC_BUILD_DEFS=-DATSC_STAMPS python -m atsc_amd.build).  Runs the bench's chained loop for a
few steps and reads the wall-clock start / end stamp of every frame of the last launch on each scratch set."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ATSC_LIB_VARIANT", "stamps")
import torch  # noqa: E402
import atsc_amd  # noqa: E402
from atsc_amd import capi  # noqa: E402
from tests import helpers as H  # noqa: E402

CH = int(os.environ.get("CHAINS", "2"))
STEPS = int(os.environ.get("STEPS", "23"))
n = 40960 * 256
NF = 40960
me = float(np.float32(5) / np.float32(100))
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
ctx.set_chains(CH)
plan = ctx.plan(H.frame_offsets(n, 256))
xs = [torch.from_numpy(H.synth_series(b, n, class_shift=2 * b)).to(dev) for b in range(5)]
outs = [plan.alloc_outputs(torch, dev) for _ in range(8)]
st = torch.cuda.current_stream().cuda_stream
L = capi.lib()
import time
for i in range(12):
    plan.compress(xs[i % 5], outs[i % 8], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
plan.join(st)
torch.cuda.synchronize()
t_0 = time.perf_counter()
for i in range(STEPS):
    plan.compress(xs[i % 5], outs[i % 8], atsc_amd.AUTO, True, me, 0, st, pipelined=True)
plan.join(st)
torch.cuda.synchronize()
print("chains %d: %.1f us per step over %d steps" % (CH, (time.perf_counter() - t_0) / STEPS * 1e6, STEPS))
if os.environ.get("BRIEF"):
    sys.exit(0)
L.atsc_dev_span_read_set.restype = C.c_int
L.atsc_dev_span_read_set.argtypes = [C.POINTER(C.c_ulonglong), C.c_uint, C.c_uint]
sets = []
for q in range(min(4, 2 * CH)):
    sp = (C.c_ulonglong * (3 * NF))()
    assert L.atsc_dev_span_read_set(sp, NF, q) == 0
    a = np.frombuffer(sp, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
    a = a[a[:, 1] > a[:, 0]]
    sets.append(a)
t0 = min(a[:, 0].min() for a in sets)
allb, alle = [], []
for q, a in enumerate(sets):
    b, e = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    allb.append(b)
    alle.append(e)
    print("set %d: frames %5d  first start %8.1f us  last end %8.1f us  span %7.1f us  life mean %.2f p50 %.2f p99 %.2f" % (
        q, len(a), b.min(), e.max(), e.max() - b.min(), (e - b).mean(), np.median(e - b), np.percentile(e - b, 99)))
tmax = max(e.max() for e in alle)
for t in np.arange(0, tmax, 10.0):
    row = [int(((b <= t) & (e > t)).sum()) for b, e in zip(allb, alle)]
    print("  t=%6.0f us  in flight %s  total %5d" % (t, " ".join("%5d" % v for v in row), sum(row)))
