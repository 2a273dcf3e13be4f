"""BASELINE.json configs[3]/[4] at full size on ONE GPU: 4096 series x 262144 samples (2^30, 8 GiB of f64),
auto e=1%, class = series_id % 5, F256 framing; then decompress of the produced stream.
usage: python tools/bench_config3_full.py [n_series]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atsc_amd
from tests import helpers as H

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
PER = 262144
dev = torch.device("cuda:0")
ctx = atsc_amd.Context(0)
st = torch.cuda.current_stream().cuda_stream
me = float(np.float32(1) / np.float32(100))
t0 = time.time()
d_x = torch.empty(S * PER, dtype=torch.float64, device=dev)
for s in range(S):
    d_x[s * PER:(s + 1) * PER] = torch.from_numpy(H.synth_series(s, PER, klass=s % 5)).to(dev)
    if s % 512 == 0:
        print("generated", s, "series", round(time.time() - t0, 1), "s", flush=True)
n = S * PER
off = np.arange(0, n + 1, 256, dtype=np.uint64)
t1 = time.time()
plan = ctx.plan(off)
outs = plan.alloc_outputs(torch, dev)
print("plan + buffers", round(time.time() - t1, 1), "s; frames", plan.n_frames, flush=True)
for _ in range(1):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
reps = 3
t2 = time.perf_counter()
for _ in range(reps):
    plan.compress(d_x, outs, atsc_amd.AUTO, True, me, 0, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t2) / reps
total = int(outs["rec_off"][-1].item())
chosen = outs["chosen"].cpu().numpy()
print(json.dumps({"config": "configs[3] %d series x %d, auto e=1%%, f256, 1 GPU" % (S, PER), "samples": n,
                  "ms": round(dt * 1e3, 2), "Msamples_s": round(n / dt / 1e6, 1), "ratio": round(8.0 * n / total, 3),
                  "encoded_bytes": total,
                  "codecs": {int(c): int(np.sum(chosen == c)) for c in np.unique(chosen)}}), flush=True)
body = outs["body"][:total].cpu().numpy().tobytes()
del outs
t3 = time.time()
dp = atsc_amd.DPlan(ctx, body)
print("decode plan (host parse of %d records)" % dp.n_frames, round(time.time() - t3, 1), "s", flush=True)
d_body = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev)
d_out = torch.empty(n, dtype=torch.float64, device=dev)
dp.decompress(d_body, d_out, st)
torch.cuda.synchronize()
t4 = time.perf_counter()
for _ in range(reps):
    dp.decompress(d_body, d_out, st)
torch.cuda.synchronize()
dtd = (time.perf_counter() - t4) / reps
# round-trip property on the whole 2^30 samples: per-frame MAPE on device
x2 = d_x.view(-1, 256); o2 = d_out.view(-1, 256)
fm = ((o2 - x2) / x2).abs().sum(dim=1) / 256.0
ch = torch.from_numpy(chosen).to(dev)
lossless = (ch == 3) | (ch == 6)
ok_lossless = bool(torch.all(o2[lossless] == x2[lossless]))
worst = float(fm[~lossless].max()) if bool((~lossless).any()) else 0.0
print(json.dumps({"config": "configs[4] decompress of the above, 1 GPU", "samples": n, "ms": round(dtd * 1e3, 2),
                  "Msamples_s": round(n / dtd / 1e6, 1), "lossless_frames_bit_exact": ok_lossless,
                  "worst_lossy_frame_mape": worst, "bound": me * 288 / 256}), flush=True)
