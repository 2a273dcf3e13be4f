"""Throughput of the other BASELINE.json configs on one GPU (dev aid; numbers quoted in DESIGN.md).
usage: python tools/bench_configs.py [quick]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atsc_amd
from tests import helpers as H

dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream


def run_compress(x, off, comp, bounded, me, reps=5, level=0):
    plan = ctx.plan(off)
    d_x = torch.from_numpy(x).to(dev)
    outs = plan.alloc_outputs(torch, dev)
    for _ in range(2):
        plan.compress(d_x, outs, comp, bounded, me, level, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.compress(d_x, outs, comp, bounded, me, level, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    total = int(outs["rec_off"][-1].item())
    body = outs["body"][:total].cpu().numpy().tobytes()
    chosen = outs["chosen"].cpu().numpy()
    return dt, total, body, {int(c): int(np.sum(chosen == c)) for c in np.unique(chosen)}


def run_compress_pipelined(xs_list, off, comp, bounded, me, reps=12, level=0):
    """Batch after batch through atsc_compress_plan_dev_pipelined: rotating resident batches, two output sets.
    Returns (seconds per batch, bytes of the last batch, bytes equal to the plain call's?)."""
    plan = ctx.plan(off)
    d_xs = [torch.from_numpy(v).to(dev) for v in xs_list]
    outs = [plan.alloc_outputs(torch, dev), plan.alloc_outputs(torch, dev)]
    ref = plan.alloc_outputs(torch, dev)
    for i in range(4):
        plan.compress(d_xs[i % len(d_xs)], outs[i % 2], comp, bounded, me, level, st, pipelined=True)
    plan.join(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % len(d_xs)], outs[i % 2], comp, bounded, me, level, st, pipelined=True)
    plan.join(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    last = (reps - 1)
    total = int(outs[last % 2]["rec_off"][-1].item())
    got = outs[last % 2]["body"][:total].cpu().numpy().tobytes()
    plan.compress(d_xs[last % len(d_xs)], ref, comp, bounded, me, level, st)
    torch.cuda.synchronize()
    rt = int(ref["rec_off"][-1].item())
    same = rt == total and ref["body"][:rt].cpu().numpy().tobytes() == got
    return dt, total, same


def run_decompress(body, n_samples, reps=5):
    dp = atsc_amd.DPlan(ctx, body)
    d_body = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n_samples, dtype=torch.float64, device=dev)
    for _ in range(2):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dp.decompress(d_body, d_out, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(name, n, dt, total, codecs):
    print(json.dumps({"config": name, "samples": n, "ms": round(dt * 1e3, 3), "Msamples_s": round(n / dt / 1e6, 1),
                      "ratio": round(8.0 * n / max(total, 1), 2), "codecs": codecs}), flush=True)


me5 = float(np.float32(5) / np.float32(100))
me1 = float(np.float32(1) / np.float32(100))
quick = len(sys.argv) > 1
# configs[1]: 1M samples, 256-pt frames, --compressor fft e=5%
for klass in (1, 0):
    n = 1 << 20
    x = H.synth_series(0, n, klass=klass)
    dt, tot, body, cod = run_compress(x, H.frame_offsets(n, 256), atsc_amd.FFT, True, me5, reps=20)
    report("configs[1] 1M f256 fft e=5%% class C%d" % klass, n, dt, tot, cod)
# configs[2] with the reference chunker's framing (80 x 131072)
n = 10485760
x = H.synth_series(0, n)
sizes = atsc_amd.chunk_sizes(n)
off = np.cumsum([0] + sizes).astype(np.uint64)
dt, tot, body, cod = run_compress(x, off, atsc_amd.AUTO, True, me5, reps=3)
report("configs[2] 10M auto e=5%% reference chunker framing (%d x 131072)" % len(sizes), n, dt, tot, cod)
dtd = run_decompress(body, n, reps=3)
report("configs[4]-like decompress of the above", n, dtd, tot, cod)
# the same framing batch after batch (four different series rotating) through the pipelined entry point: the large
# tier's per-frame kernels of batch i overlap the grids of batch i + 1
xsl = [x] + [H.synth_series(s, n) for s in (1, 2, 3)]
dtp, totp, same = run_compress_pipelined(xsl, off, atsc_amd.AUTO, True, me5)
print(json.dumps({"config": "configs[2] chunker framing, pipelined entry point, 4 batches rotating", "samples": n,
                  "ms": round(dtp * 1e3, 3), "Msamples_s": round(n / dtp / 1e6, 1), "bytes_equal_plain_call": same}), flush=True)
# configs[2] F256 + decompress
dt, tot, body, cod = run_compress(x, H.frame_offsets(n, 256), atsc_amd.AUTO, True, me5, reps=10)
report("configs[2] 10M auto e=5% f256", n, dt, tot, cod)
dtd = run_decompress(body, n, reps=10)
report("configs[4]-like decompress of the above (f256)", n, dtd, tot, cod)
# configs[3] shape at reduced size: S series x 262144, auto e=1%, class = s % 5
S = 16 if quick else 256
xs = np.concatenate([H.synth_series(s, 262144, klass=s % 5) for s in range(S)])
n = len(xs)
dt, tot, body, cod = run_compress(xs, H.frame_offsets(n, 256), atsc_amd.AUTO, True, me1, reps=3)
report("configs[3] shape: %d series x 262144 auto e=1%% f256" % S, n, dt, tot, cod)
dtd = run_decompress(body, n, reps=3)
report("configs[4] decompress of the above (f256)", n, dtd, tot, cod)
off = np.arange(0, n + 1, 131072, dtype=np.uint64)
dt, tot, body, cod = run_compress(xs, off, atsc_amd.AUTO, True, me1, reps=2)
report("configs[3] shape: %d series x 262144 auto e=1%% chunker framing" % S, n, dt, tot, cod)
dtd = run_decompress(body, n, reps=2)
report("configs[4] decompress of the above (chunker framing)", n, dtd, tot, cod)

# The paper's production shape (VLDB'24 industrial track p.8, section 7: 13 950 signals x 5 432 samples): the reference chunker
# cuts every series into 4096 + 1024 + 312 samples (optimizer/mod.rs:78-98: power-of-two chunks, a tail of up to 512 samples
# as it is), so one batch holds three frame lengths -- three k_compress classes.  auto, e = 3 % (the paper's setting);
# class = series % 5.
if not quick or os.environ.get("PAPER"):
    NSER, PER = 13950, 5432
    me3 = float(np.float32(3) / np.float32(100))
    xs = np.concatenate([H.synth_series(s, PER, klass=s % 5) for s in range(NSER)])
    sizes = atsc_amd.chunk_sizes(PER)
    assert sum(sizes) == PER, sizes
    off = np.concatenate([[0], np.cumsum(np.tile(sizes, NSER))]).astype(np.uint64)
    n = len(xs)
    dt, tot, body, cod = run_compress(xs, off, atsc_amd.AUTO, True, me3, reps=5)
    report("paper shape: %d series x %d (%s), auto e=3%%" % (NSER, PER, " + ".join(str(v) for v in sizes)), n, dt, tot, cod)
    dtd = run_decompress(body, n, reps=5)
    report("decompress of the above", n, dtd, tot, cod)
    d_list = [xs, np.roll(xs, PER * 7), np.roll(xs, PER * 13)]
    dtp, totp, same = run_compress_pipelined(d_list, off, atsc_amd.AUTO, True, me3)
    print(json.dumps({"config": "paper shape, pipelined entry point, 3 batches rotating", "samples": n, "ms": round(dtp * 1e3, 3),
                      "Msamples_s": round(n / dtp / 1e6, 1), "bytes_equal_plain_call": same}), flush=True)
