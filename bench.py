#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ATSC compressor path.

Metric (BASELINE.json): Msamples/sec compressed (auto, e=5%) + ratio.
Workload at N = 1: BASELINE.json configs[2] -- 10,485,760 synthetic f64 samples in 40960 frames of 256
(classes C0..C4 cycled per 65536-sample block), `--compressor auto`, max_error =
(float)5/100; four different series are resident and the timed loop rotates over them.  One "step" =
one pass of the hot path (per-frame FFT / Catmull-Rom / RLE / Constant fit + error check + selector +
BRO record packing) over one batch with the samples already resident in HBM.
N > 1: BASELINE.json configs[3] -- 4096 series x 262,144 samples (2^30), auto e=1%, series sharded
contiguously over the ranks (4096 / N each: total work fixed, "strong"), one process per GPU
(torch.distributed, backend nccl = RCCL), no data-path collective; the only exchange is the gather of the
encoded records to rank 0, inside the timed region and overlapped with the next step's codecs.
`--workload config3` runs the same workload on one GPU (the curve's 1-GPU point).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime deals streams over a handful of hardware queues (four by default); a chain of the pipelined entry
# point that shares its queue with another stream's dependent work stalls behind it ("Several
# chains").  Eight queues leave room for the chains beside PyTorch's and the communicator's streams.  Read by the
# runtime when it starts, so it is set before anything imports torch; a value the caller set wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_SAMPLES = 10485760
FRAME = 256
ERROR_PCT = 5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


# one worker of the all-core CPU baseline: frames [f0, f1) of the saved sample through the oracle,
# started at the agreed time; prints "<seconds late> <end time>"
_CPU_WORKER = r"""
import sys, time
import numpy as np
root, path, f0, f1, me, start_at = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
sys.path.insert(0, root)
from oracle import oracle as orc
F = %d
xs = np.ascontiguousarray(np.load(path, mmap_mode="r")[f0 * F:f1 * F])
so = np.arange(0, (f1 - f0) * F + 1, F, dtype=np.uint64)
orc.stream_compress(xs[:F * 8], so[:9], orc.AUTO, True, me, 0)
late = max(0.0, time.time() - start_at)
while time.time() < start_at:
    time.sleep(0.001)
orc.stream_compress(xs, so, orc.AUTO, True, me, 0)
print(late, time.time())
""" % FRAME


def cpu_baseline(x, off, me, frames_per_block=256, blocks=160):
    """Times the CPU oracle (C restatement of the reference algorithm, single thread) on a bounded
    sample of the same workload: the first `frames_per_block` frames of each of the first `blocks`
    65536-sample blocks (so every class appears in the workload's proportion).  The defaults take
    the whole batch (40960 frames, about 5 s of CPU work)."""
    from oracle import oracle as orc

    orc.build()
    idx = []
    per_block = 65536 // FRAME
    for b in range(blocks):
        idx.extend(range(b * per_block, b * per_block + frames_per_block))
    xs = np.concatenate([x[int(off[i]):int(off[i + 1])] for i in idx])
    so = np.arange(0, len(xs) + 1, FRAME, dtype=np.uint64)
    orc.stream_compress(xs[: FRAME * 8], so[:9], orc.AUTO, True, me, 0)  # warm-up
    t0 = time.perf_counter()
    bro, _, _ = orc.stream_compress(xs, so, orc.AUTO, True, me, 0)
    dt = time.perf_counter() - t0
    # The same sample over every host core the process may use: one fresh `python -c` worker per core
    # (they never see the GPU context), released together at an agreed wall-clock time; bounded by a
    # timeout, after which the workers started here are killed by PID and the figure is left out.
    # The reference itself is single-threaded, so this is extra information, not `value`.
    import subprocess
    import tempfile

    cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # a one-GPU box's CPU share is 16 cores
    nfr = len(idx)
    cuts = [nfr * k // cores for k in range(cores + 1)]
    all_cores = None
    procs = []
    try:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sample.npy")
            np.save(path, xs)
            start_at = time.time() + 6.0
            for k in range(cores):
                if cuts[k + 1] > cuts[k]:
                    procs.append(subprocess.Popen(
                        [sys.executable, "-c", _CPU_WORKER, ROOT, path, str(cuts[k]), str(cuts[k + 1]), repr(me),
                         repr(start_at)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            ends = []
            for pr in procs:
                out, _ = pr.communicate(timeout=180)
                if pr.returncode != 0:
                    raise RuntimeError("worker exit %d" % pr.returncode)
                late, end = out.split()
                if float(late) > 0:
                    raise RuntimeError("a worker was not ready at the start time")
                ends.append(float(end))
            all_cores = {"value": len(xs) / (max(ends) - start_at) / 1e6, "unit": "Msamples/s", "cores": len(procs)}
    except Exception as e:  # pragma: no cover - the single-thread figure stands on its own
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        sys.stderr.write("all-core CPU baseline skipped: %r\n" % (e,))
    return {
        "all_cores": all_cores,
        "value": len(xs) / dt / 1e6,
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d frames x %d (first %d frames of each of the first %d class blocks), %.1f s, "
                  "C restatement of the reference algorithm (oracle/), 1 thread, ratio %.2f" % (
                      len(idx), FRAME, frames_per_block, blocks, dt, 8.0 * len(xs) / len(bro)),
    }


def end_to_end(ctx, atsc_amd, x, off, me, reps=5):
    """Host buffer -> host BRO bytes and back through the host-pointer entry points (PCIe inclusive;
    SURVEY.md 8(d) asks for it next to the device-resident figure).  Never `value`."""
    import ctypes as C

    lib = atsc_amd.capi.lib()
    n = len(x)
    out = {}

    def timed(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return r, float(np.median(ts))

    # atsc_compress_frames, F256 framing (what INTEGRATION.md's Rust binding calls)
    nf = len(off) - 1
    cap = int(nf * (32 + 14 * FRAME + 16))
    body = np.empty(cap, dtype=np.uint8)
    blen = C.c_uint64()
    offc = np.ascontiguousarray(off, dtype=np.uint64)

    def cf():
        rc = lib.atsc_compress_frames(ctx._h, x.ctypes.data_as(C.POINTER(C.c_double)),
                                      offc.ctypes.data_as(C.POINTER(C.c_uint64)), nf, atsc_amd.AUTO, 1,
                                      C.c_float(np.float32(me)), 0, body.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
                                      C.byref(blen), None, None, None)
        atsc_amd.capi.check(rc, ctx._h)
        return blen.value

    nbytes, dt = timed(cf)
    out["compress_frames_f256_pageable"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3, "bytes": nbytes}
    rec = bytes(body[:nbytes])
    recarr = np.frombuffer(rec, dtype=np.uint8).copy()
    dec = np.empty(n, dtype=np.float64)
    on = C.c_uint64()

    def df():
        rc = lib.atsc_decompress_frames(ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(rec), 0,
                                        dec.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(on))
        atsc_amd.capi.check(rc, ctx._h)
        return on.value

    got, dt = timed(df)
    assert got == n
    out["decompress_frames_f256_pageable"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3}
    # the same two calls on buffers the caller page-locked once (atsc_host_register: what a long-lived host process
    # does with its ingest and result buffers): the copies become DMA transfers the host does not wait behind
    regs = [(x, x.nbytes), (body, body.nbytes), (recarr, recarr.nbytes), (dec, dec.nbytes)]
    done = []
    try:
        for a, nb in regs:
            atsc_amd.capi.check(lib.atsc_host_register(C.c_void_p(a.ctypes.data), nb), ctx._h)
            done.append(a)
        nb2, dt = timed(cf)
        assert nb2 == nbytes and bytes(body[:nbytes]) == rec
        out["compress_frames_f256"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3, "bytes": nbytes}
        got, dt = timed(df)
        assert got == n
        out["decompress_frames_f256"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3}
    finally:
        for a in done:
            lib.atsc_host_unregister(C.c_void_p(a.ctypes.data))
    # atsc_compress_data / atsc_decompress_data: the atsc CLI's calls, reference chunker framing (80 x 131072)
    bro = C.POINTER(C.c_uint8)()
    ln = C.c_uint64()

    def cd():
        rc = lib.atsc_compress_data(ctx._h, x.ctypes.data_as(C.POINTER(C.c_double)), n, atsc_amd.AUTO, ERROR_PCT, 0,
                                    C.byref(bro), C.byref(ln))
        atsc_amd.capi.check(rc, ctx._h)
        b = C.string_at(bro, ln.value)
        lib.atsc_free(bro)
        return b

    bro_bytes, dt = timed(cd)
    out["compress_data_chunker"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3, "bytes": len(bro_bytes)}
    broarr = np.frombuffer(bro_bytes, dtype=np.uint8)
    outp = C.POINTER(C.c_double)()
    cnt = C.c_uint64()

    def dd():
        rc = lib.atsc_decompress_data(ctx._h, broarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(bro_bytes),
                                      C.byref(outp), C.byref(cnt))
        atsc_amd.capi.check(rc, ctx._h)
        lib.atsc_free(outp)
        return cnt.value

    got, dt = timed(dd)
    assert got == n
    out["decompress_data_chunker"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3}
    out["note"] = ("host buffer -> host buffer, PCIe inclusive, median of %d calls; *_f256: caller buffers registered "
                   "once with atsc_host_register; *_pageable and *_chunker: pageable caller memory" % reps)
    return out


def pmc_traffic(key):
    """HBM bytes per launch / per call from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE doubled per
    the gfx950 rule + WRITE_SIZE, collected by separate rocprofv3 --pmc runs as MI355X_MICROARCH.md prescribes -- the
    counters cannot be read inside this run).  -> (bytes or None, source)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = d[key]
        return float(e["bytes"]), "profiles/%s (%s)" % (e["file"], d.get("note", "separate rocprofv3 --pmc passes"))
    except Exception:
        return None, "no committed PMC pass for this path"


def path_roofline(algo_bytes, gpu_ms, what, traffic):
    """A path's launches against the HBM roofline: algorithmic bytes of one call / GPU time of one call (HIP events on
    the stream the launches run on, calls back to back)."""
    ach = algo_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else 0.0
    return {"bound": "hbm", "kernels": what, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic[0], "traffic_source": traffic[1],
            "gpu_ms_per_call": gpu_ms, "algorithmic_bytes_per_call": algo_bytes}


def chunker_framing(ctx, atsc_amd, torch, dev, d_xs, me, stream):
    """The same batches in the reference CLI's own framing (OptimizerPlan::get_chunks_sizes, optimizer/mod.rs:78-98:
    80 frames of 131072 samples), device resident: plain calls, and batch after batch through the pipelined entry
    point.  Extra information next to `value` (which is the 256-sample framing BASELINE.json names)."""
    n = d_xs[0].numel()
    sizes = atsc_amd.chunk_sizes(n)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    plan = ctx.plan(off)
    NO = 8  # one output set per batch in flight (up to four chains x two scratch sets)
    outs = [plan.alloc_outputs(torch, dev) for _ in range(NO)]
    R = len(d_xs)
    for i in range(2):
        plan.compress(d_xs[i % R], outs[0], atsc_amd.AUTO, True, me, 0, stream)
    torch.cuda.synchronize()
    reps = 40  # (the one synchronisation at the end is ~20 us: spread over enough calls not to show)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(reps):
        plan.compress(d_xs[i % R], outs[0], atsc_amd.AUTO, True, me, 0, stream)
    ev1.record()
    torch.cuda.synchronize()
    dt_plain = (time.perf_counter() - t0) / reps
    gpu_ms_plain = ev0.elapsed_time(ev1) / reps  # plain calls enqueue everything on the caller's stream
    total = int(outs[0]["rec_off"][-1].item())
    body = outs[0]["body"][:total].cpu().numpy().tobytes()
    for i in range(8):
        plan.compress(d_xs[i % R], outs[i % NO], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.synchronize()
    reps = 80
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % R], outs[i % NO], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.synchronize()
    dt_pipe = (time.perf_counter() - t0) / reps
    dp = atsc_amd.DPlan(ctx, body)
    d_body = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        dp.decompress(d_body, d_out, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(20):
        dp.decompress(d_body, d_out, stream)
    ev1.record()
    torch.cuda.synchronize()
    dt_dec = (time.perf_counter() - t0) / 20
    gpu_ms_dec = ev0.elapsed_time(ev1) / 20
    dp.close()
    plan.close()
    algo = 8.0 * n + total  # samples read + record bytes written (compress); the reverse (decompress)
    return {"frames": len(sizes), "frame_len": int(sizes[0]), "ratio": 8.0 * n / (total + 12),
            "compress": {"value": n / dt_plain / 1e6, "unit": "Msamples/s", "ms": dt_plain * 1e3,
                         "roofline": path_roofline(algo, gpu_ms_plain, "the large tier's launches of one call (k_large_cols243, "
                                                   "k_large_rows9p, k_large_decide1, k_large_trip243, k_large_decide2, packing), "
                                                   "first launch to last", pmc_traffic("chunker_compress"))},
            "compress_pipelined": {"value": n / dt_pipe / 1e6, "unit": "Msamples/s", "ms": dt_pipe * 1e3},
            "decompress": {"value": n / dt_dec / 1e6, "unit": "Msamples/s", "ms": dt_dec * 1e3,
                           "roofline": path_roofline(algo, gpu_ms_dec, "k_large_dparse + k_large_trip243<true> (+ k_decompress_large "
                                                     "for what they leave)", pmc_traffic("chunker_decompress"))},
            "note": "device resident; the reference chunker's framing of the same batches (the atsc CLI's framing)"}


ROTATE = 5          # resident batches the timed loop cycles through (N = 1 workload); coprime with the chains
C3_SERIES = 4096    # configs[3]: 4096 series x 262144 samples = 2^30
C3_PER = 262144
C3_ERROR_PCT = 1
C3_ROTATE = 3       # resident batches per rank for configs[3] (different series values, same classes); coprime with the 4 scratch sets
SPINUP_S = 0.25     # untimed spin-up in front of the timed loop (GPU clocks ramp with load)
XGMI_LINK_GBS = 70.0  # per-direction rate of one xGMI link assumed by --root-weight auto


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--workload", choices=["auto", "config2", "config3"], default="auto",
                    help="auto: configs[2] at N = 1 (the metric's config), configs[3]'s per-rank share at N > 1")
    ap.add_argument("--series", type=int, default=C3_SERIES, help="config3 only: total series over all ranks")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="plain single-stream calls (atsc_compress_plan_dev) instead of the chained entry point "
                         "(atsc_compress_plan_dev_pipelined)")
    ap.add_argument("--chains", type=int, default=0, help="chains of the pipelined entry point (1..4; 0: the library's default: 4 with GPU_MAX_HW_QUEUES >= 8, which this script sets, else 2)")
    ap.add_argument("--no-decompress", action="store_true", help="skip the decompression measurement")
    ap.add_argument("--no-adaptive-order", action="store_true", help="configs[3] workload: frames in index order (see --adaptive-order)")
    ap.add_argument("--adaptive-order", action="store_true",
                    help="start the frames costliest-first instead of in index order (cost = shader clocks of the same "
                         "frame slot in an earlier batch of the same chain); off by default: with two chains in flight the "
                         "drain of one launch is covered by the other chain's kernel and `value` leans on no hint")
    ap.add_argument("--root-weight", default="1",
                    help="N > 1: rank 0's share relative to a peer's (atsc_shard_range_weighted); 'auto' = (B / R) / T1 from "
                         "an untimed single-rank measurement and %.0f GB/s per link" % XGMI_LINK_GBS)
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: every rank keeps its records (the reference writes one .bro per input file, "
                         "main.rs:121-124): no collective at all in the timed loop")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X; there is no CPU fallback for the compressor")
    # Rehearsal knobs (never set by the driver): ATSC_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # moves the record gather to gloo, so the N > 1 control flow can be exercised on a one-GPU box
    # (RCCL refuses two ranks on one device).
    share = os.environ.get("ATSC_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as G

    G.build()
    import atsc_amd
    from atsc_amd import parallel
    from tests import helpers as H

    workload = args.workload
    if workload == "auto":
        workload = "config2" if world == 1 else "config3"
    ctx = atsc_amd.Context(dev_index)
    if args.chains:
        ctx.set_chains(args.chains)
    # configs[3]: a series keeps its class from batch to batch (slot i = the same series one window later), which is
    # the recurring layout the cost hint is for, and C3_ROTATE is coprime with the scratch sets in rotation, so the
    # hint a step uses always comes from ANOTHER resident batch (other values of the same series): on unless asked not to
    if workload == "config3" and not args.no_adaptive_order:
        args.adaptive_order = True
    ctx.set_adaptive_order(bool(args.adaptive_order))
    root_weight = 1.0
    if workload == "config2":
        # BASELINE.json configs[2]: one 10,485,760-sample series per batch, classes cycled per 65536-sample block.
        # ROTATE different series are resident, each with ANOTHER block -> class mapping (class = (block + 2 b) % 5 for
        # batch b), and the timed loop walks through them: no step sees a batch it has seen before in the same slot of
        # a chain, and an optional cost hint (--adaptive-order) comes from a batch with a different layout.
        error_pct = ERROR_PCT
        n_local = N_SAMPLES
        xs = [H.synth_series(rank * ROTATE + b, N_SAMPLES, class_shift=2 * b) for b in range(ROTATE)]
        d_xs = [torch.from_numpy(v).to(dev) for v in xs]
        units_total = world * N_SAMPLES
        desc = ("BASELINE.json configs[2] per GPU: 10,485,760 f64 samples, 40960 frames x 256, --compressor auto, "
                "e=5%% (max_error=(float)5/100), classes C0-C4 cycled per 65536-sample block, inputs resident in "
                "HBM; the timed loop rotates over %d different resident batches (series ids %d..%d, batch b with the "
                "block -> class mapping shifted by 2 b)" % (ROTATE, rank * ROTATE, rank * ROTATE + ROTATE - 1))
    else:
        # BASELINE.json configs[3]: 4096 series x 262,144 samples (2^30), class = series % 5, auto e=1%,
        # series s on rank s / (4096 / N): total work fixed, per-rank share 4096 / N series
        # (rank 0 more than a peer with --root-weight).
        error_pct = C3_ERROR_PCT
        if args.root_weight == "auto":
            root_weight = None  # decided below, from an untimed measurement
        else:
            root_weight = float(args.root_weight)
        xs = None
        units_total = args.series * C3_PER

    me = float(np.float32(error_pct) / np.float32(100))
    stream = torch.cuda.current_stream().cuda_stream

    def c3_batch(sb, se, variant):
        """Series sb..se-1 of configs[3]; variant v: series id s + v * C3_SERIES (same class, other values)."""
        d_x = torch.empty((se - sb) * C3_PER, dtype=torch.float64, device=dev)
        for s in range(sb, se):
            d_x[(s - sb) * C3_PER:(s - sb + 1) * C3_PER] = H.synth_series_torch(torch, dev, s + variant * C3_SERIES, C3_PER, s % 5)
        return d_x

    if workload == "config3":
        if root_weight is None:
            # (B / R) / T1: a 64-series sample on this rank gives the codec rate and the record bytes per sample
            root_weight = 1.0
            if world > 1:
                ns = 64
                d_s = c3_batch(0, ns, 0)
                offs = np.arange(0, ns * C3_PER + 1, FRAME, dtype=np.uint64)
                pl = ctx.plan(offs)
                o = pl.alloc_outputs(torch, dev)
                for _ in range(2):
                    pl.compress(d_s, o, atsc_amd.AUTO, True, me, 0, stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    pl.compress(d_s, o, atsc_amd.AUTO, True, me, 0, stream)
                torch.cuda.synchronize()
                t1 = (time.perf_counter() - t0) / 4 * (args.series / ns)  # single-GPU codec time of the whole job
                bts = float(o["rec_off"][-1].item()) * (args.series / ns)
                pl.close()
                del d_s, o
                w = parallel.root_weight_for(t1, bts, XGMI_LINK_GBS * 1e9)
                t = torch.tensor([w], dtype=torch.float64, device="cpu" if share else dev)
                dist.broadcast(t, src=0)
                root_weight = float(t.item())
        sb, se = (parallel.shard_range_weighted(args.series, rank, world, root_weight) if root_weight != 1.0
                  else parallel.shard_range(args.series, rank, world))
        n_local = (se - sb) * C3_PER
        d_xs = [c3_batch(sb, se, v) for v in range(C3_ROTATE)]
        desc = ("BASELINE.json configs[3]: %d series x 262,144 f64 samples (%d samples over all ranks), "
                "class = series %% 5, --compressor auto, e=1%% (max_error=(float)1/100), 256-sample frames, series "
                "sharded contiguously by rank (%d on rank 0%s), inputs resident in HBM (%d resident batches per rank, "
                "rotating), %s" % (args.series, units_total, se - sb,
                                   ", root weight %.2f" % root_weight if root_weight != 1.0 else "", C3_ROTATE,
                                   "every rank keeps its records (no collective)" if args.no_gather else
                                   "encoded records gathered to rank 0 every step"))
    off = np.arange(0, n_local + 1, FRAME, dtype=np.uint64)
    plan = ctx.plan(off)

    # Steady state of a compression service: batch after batch through atsc_compress_plan_dev_pipelined -- consecutive
    # calls rotate over the plan's chains (streams of the context's own, two scratch sets each), so the kernels of
    # batch i + 1 fill the launch gap and the drain of batch i; for N > 1 the gather of step i is issued from a side
    # stream that waits for that step's records (atsc_plan_join).
    pipelined = not args.no_pipeline
    NOUT = 8  # output sets: one per batch in flight (up to four chains x two sets)
    outs2 = [plan.alloc_outputs(torch, dev) for _ in range(NOUT if pipelined else 1)]
    pg = None
    gather_on = world > 1 and not args.no_gather
    gstream = torch.cuda.Stream(device=dev) if world > 1 else None
    R = len(d_xs)

    def step(i, pipe=pipelined, gather=True):
        o = outs2[i % len(outs2)]
        slot = i % 2
        if pg is not None and gather:
            pg.before_produce(slot)
        plan.compress(d_xs[i % R], o, atsc_amd.AUTO, True, me, 0, stream, pipelined=pipe)
        if gather_on and gather:
            # the path's only exchange: the encoded records go to rank 0
            if pg is not None:
                if pipe:
                    plan.join(gstream.cuda_stream)
                    with torch.cuda.stream(gstream):
                        pg.submit(slot, o["body"], o["rec_off"][-1:])
                else:
                    pg.submit(slot, o["body"], o["rec_off"][-1:])
                return
            if pipe:
                plan.join(stream)
            if share:  # rehearsal on one GPU: gloo moves host tensors
                nb = int(o["rec_off"][-1].item())
                parallel.gather_records(dist, torch, o["body"][:nb].cpu(), nb, rank, world)
            else:
                parallel.gather_records(dist, torch, o["body"], o["rec_off"][-1:], rank, world)

    def timed_loop(steps, pipe, gather=True):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, pipe, gather)
        if pipe:
            plan.join(stream)
        if pg is not None and gather:
            pg.drain()
        torch.cuda.synchronize()  # device-wide: codec, pack and communicator streams
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # encoded bytes of every resident batch (untimed): the ratio and the algorithmic bytes are averages over them
    body_bytes_b = []
    for b in range(R):
        plan.compress(d_xs[b], outs2[0], atsc_amd.AUTO, True, me, 0, stream, pipelined=False)
        torch.cuda.synchronize()
        body_bytes_b.append(int(outs2[0]["rec_off"][-1].item()))
    chosen = outs2[0]["chosen"].cpu().numpy()
    # (at least one batch through every set of every chain: the sets are built by the first call)
    warm_steps = max(args.warmup, NOUT if pipelined else 0, 1 if world > 1 else 0)
    for i in range(warm_steps):
        step(i, gather=False)
    if pipelined:
        plan.join(stream)
    torch.cuda.synchronize()
    # Untimed spin-up.  The inputs were generated on the host for seconds while the GPU idled, and its clocks follow the
    # load: the same 20-step loop takes 103 us per step right behind another loop and 125 us after half a second of idle
    # (tools/fixed_cost_probe.py, profiles/r04_fixed_cost.txt).  --warmup W stays the minimum; the loop below keeps
    # stepping (same calls, results discarded) until SPINUP_S seconds of GPU work have passed, so that a short timed
    # region measures the steady state a longer one does.  Reported as `warmup_effective`.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < SPINUP_S:
        for i in range(8):
            step(warm_steps + i, gather=False)
        warm_steps += 8
        if pipelined:
            plan.join(stream)
        torch.cuda.synchronize()
    gather_mode = None
    if gather_on:
        gather_mode = "size all-gather + point-to-point sends (gather_records)"
    if gather_on and not share:
        # segment capacity agreed once from the warm-up result; no host sync inside the timed loop.
        # One untimed trial step validates the asynchronous gather on this backend; any exception
        # falls back to the simple size-exchange + send/recv gather (every rank takes the same branch:
        # the flag is agreed by an all-reduce).
        ok = 1
        try:
            pg = parallel.PipelinedGather(dist, torch, rank, world, dev, max(body_bytes_b), slack=1.05)
            step(0)
            pg.drain()
            torch.cuda.synchronize()
            if pg.overflowed():
                raise RuntimeError("trial step outgrew the agreed segment capacity")
        except Exception as e:  # pragma: no cover - depends on the communication backend
            sys.stderr.write("pipelined gather unavailable (%r); using the simple gather\n" % (e,))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            pg = None
        else:
            gather_mode = ("one asynchronous fixed-capacity gather per step (PipelinedGather: capacity = the largest "
                           "peer's records, the root's stay in place), overlapped with the next step's codecs")
    dt = timed_loop(args.steps, pipelined)
    # the same loop over ten times the steps (at least 200): what the fill and drain of the chains cost a short timed
    # region shows as the difference (`steady_state`); never `value`
    steady = None
    if world == 1 and pipelined and args.steps < 200:
        k2 = max(200, 10 * args.steps)
        dt2 = timed_loop(k2, pipelined)
        steady = {"steps": k2, "value": units_total * k2 / dt2 / 1e6, "ms_per_step": dt2 / k2 * 1e3,
                  "note": "the timed loop again over more steps: the chains' fill and drain (about one step's worth of "
                          "time per timed region) spread over more of them"}
    gathered_sizes = None
    if pg is not None:
        if pg.overflowed():
            raise SystemExit("a timed step's records outgrew the gather capacity: the run is void")
        if rank == 0:
            segs, gathered_sizes = pg.result((args.steps - 1) % 2)
            assert len(segs) == world and gathered_sizes[0] == body_bytes_b[(args.steps - 1) % R]
    # N > 1: the same loop without the exchange -- what the codecs alone sustain (the curve's fabric-free reading)
    value_codec_only = None
    gather_ms = None
    if world > 1:
        dt_codec = timed_loop(args.steps, pipelined, gather=False)
        value_codec_only = units_total * args.steps / dt_codec / 1e6
        gather_ms = max(0.0, (dt - dt_codec) / args.steps * 1e3)
    # The same steps through plain single-stream calls (atsc_compress_plan_dev: codecs, then the packing, on the
    # caller's stream, frames in index order), same rotation of batches.  The launches of this loop do not overlap
    # each other: the dominant kernel's HIP events (start / stop of the dispatch itself) are taken here.
    value_no_hint = None
    kern_ms, launches = 0.0, 0
    if world == 1:
        for i in range(2):
            step(i, False)
        torch.cuda.synchronize()
        ctx.set_profiling(True)
        dt_plain = timed_loop(args.steps, False)
        kern_ms, launches = ctx.profile_read()
        ctx.set_profiling(False)
        value_no_hint = units_total * args.steps / dt_plain / 1e6
        # For continuity with rounds 1-2 (whose roofline launches ran costliest-first on a hint from a batch of the same
        # layout): the same kernel on ONE chain with the cost order on and one resident batch -- the order is exact,
        # i.e. what the kernel does when its drain is not the issue.  Reported beside the roofline, never as `value`.
        ordered = None
        if workload == "config2" and pipelined:
            ctx.set_chains(1)
            ctx.set_adaptive_order(True)
            for i in range(6):
                plan.compress(d_xs[0], outs2[i % 2], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
            plan.join(stream)
            torch.cuda.synchronize()
            ctx.set_profiling(True)
            t0 = time.perf_counter()
            for i in range(args.steps):
                plan.compress(d_xs[0], outs2[i % 2], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
            plan.join(stream)
            torch.cuda.synchronize()
            dt_o = time.perf_counter() - t0
            km, kl = ctx.profile_read()
            ctx.set_profiling(False)
            ctx.set_adaptive_order(bool(args.adaptive_order))
            ctx.set_chains(args.chains if args.chains else (4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8 else 2))
            ordered = {"kernel_ms_avg": km / max(kl, 1), "launches": kl, "ms_per_step": dt_o / args.steps * 1e3,
                       "note": "one chain, cost order from the SAME resident batch (exact hint), as rounds 1-2 measured it"}

    # BASELINE.json configs[4] beside it: every rank decodes the records it encoded, device resident (the frame
    # table of the records is parsed once, untimed: atsc_dplan_create); no exchange of any kind on this path.
    decomp = None
    if not args.no_decompress:
        o = outs2[0]
        plan.compress(d_xs[0], o, atsc_amd.AUTO, True, me, 0, stream, pipelined=False)
        torch.cuda.synchronize()
        nb0 = int(o["rec_off"][-1].item())
        dp = atsc_amd.DPlan(ctx, o["body"][:nb0].cpu().numpy())
        d_out = torch.empty(n_local, dtype=torch.float64, device=dev)
        for _ in range(2):
            dp.decompress(o["body"], d_out, stream)
        torch.cuda.synchronize()
        # every frame's mean absolute percentage error is within the bound, so the batch's is
        xin = d_xs[0]
        nzm = xin != 0
        mape = float(((d_out[nzm] - xin[nzm]).abs() / xin[nzm].abs()).sum().item()) / float(n_local)
        if not (mape <= me * 1.0001 + 1e-9):
            raise SystemExit("decoded batch misses the error bound: %r > %r" % (mape, me))
        del nzm
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            dp.decompress(o["body"], d_out, stream)
        ev1.record()
        torch.cuda.synchronize()
        dec_gpu_ms = ev0.elapsed_time(ev1) / args.steps  # the decoder's launches run on this stream, back to back
        if world > 1:
            dist.barrier()
        dt_dec = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_dec], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_dec = float(t.item())
        decomp = {"metric": "Msamples/sec decompressed", "value": units_total * args.steps / dt_dec / 1e6,
                  "unit": "Msamples/s", "ms_per_step": dt_dec * 1e3 / args.steps, "steps": args.steps,
                  "batch_mape": mape,
                  # decompress = record bytes read + 8 B written per output sample
                  "roofline": path_roofline(nb0 + 8.0 * n_local, dec_gpu_ms, "k_decompress<1,5> (one launch per call)",
                                            pmc_traffic("decompress_f256")),
                  "config": "BASELINE.json configs[4]: every rank decodes the records of its own shard (device "
                            "resident, frame table parsed once); no collective"}
        del dp, d_out

    body_bytes = float(np.mean(body_bytes_b))
    if world > 1:
        t = torch.tensor([body_bytes_b[0], n_local], dtype=torch.int64, device="cpu" if share else dev)
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        per_rank_bytes = [int(v[0].item()) for v in lst]
        per_rank_samples = [int(v[1].item()) for v in lst]
        total_body = float(sum(per_rank_bytes))
    else:
        per_rank_bytes = [int(body_bytes_b[0])]
        per_rank_samples = [n_local]
        total_body = body_bytes

    if rank == 0:
        value = units_total * args.steps / dt / 1e6
        codecs = {atsc_amd.capi.COMPRESSOR_NAMES[int(c)]: int(np.sum(chosen == c)) for c in np.unique(chosen)}
        # roofline of the dominant kernel (k_compress<1,5,false,256>: every 256-sample frame of the batch).
        # Algorithmic bytes per launch: 8 B read per input sample + encoded record bytes.
        algo_bytes = 8.0 * n_local + body_bytes
        k_avg_ms = kern_ms / max(launches, 1)
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        out = {
            "metric": "Msamples/sec compressed (auto, e=%d%%)" % error_pct,
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if workload == "config2" else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "ratio": 8.0 * units_total / (total_body + 12 * world) if workload == "config3"
                     else 8.0 * n_local / (body_bytes + 12),
            "config": {
                "workload": desc,
                "frames_per_gpu": plan.n_frames,
                "frame_len": FRAME,
                "codecs_rank0": codecs,
                "encoded_bytes_rank0": int(body_bytes),
                "encoded_bytes_per_rank": per_rank_bytes,
                "samples_per_rank": per_rank_samples,
                "parallelism": ("frames sharded by rank over %d processes, backend %s (%d ranks in the process group), %s"
                                % (world, backend, dist.get_world_size(),
                                   "no collective: every rank keeps its records" if args.no_gather else
                                   "records gathered to rank 0: %s" % gather_mode))
                               if world > 1 else "single GPU",
                "gathered_bytes_last_step": gathered_sizes,
                "pipeline": ("atsc_compress_plan_dev_pipelined: consecutive batches rotate over %s chains (streams of the "
                             "context's own, two scratch sets each); frames start in %s"
                             % (args.chains or "the default %d" % (4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8 else 2),
                                "index order (no cost hint)" if not args.adaptive_order else
                                ("cost order (clocks of the same slot in an earlier batch of the chain -- another resident "
                                 "batch: other values of the same series, same class)" if workload == "config3" else
                                 "cost order (clocks of the same slot in an earlier batch of the chain -- a batch with "
                                 "another class layout)")))
                            if pipelined else "single stream, plain calls",
                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
            },
        }
        if world > 1:
            out["value_codec_only"] = value_codec_only
            out["gather_ms_per_step"] = gather_ms
            out["root_weight"] = root_weight
            out["note_scaling"] = ("value = all ranks' samples / max-over-ranks time with the gather inside the timed region; "
                                   "value_codec_only = the same loop without the exchange")
        if world == 1:
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "k_compress<1,5,false,256>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_compress_256")[0],
                "traffic_source": pmc_traffic("k_compress_256")[1],
                "kernel_ms_avg": k_avg_ms,
                "kernel_launches": launches,
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "launch durations from the plain single-stream loop of this run (value_no_hint: frames in index order, "
                        "no overlap between launches); the launches of the chained loop behind `value` overlap each other, so "
                        "their individual durations say nothing about the kernel -- per step that loop moves the same bytes in "
                        "ms_per_step, see `effective`",
                "effective": {"achieved": algo_bytes / (dt / args.steps) / 1e9, "frac": algo_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                              "note": "algorithmic bytes per step / ms_per_step of the chained loop"},
            }
            if ordered is not None:
                out["roofline"]["cost_ordered"] = dict(ordered, achieved=algo_bytes / (ordered["kernel_ms_avg"] * 1e-3) / 1e9,
                                                       frac=algo_bytes / (ordered["kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
        out["warmup_effective"] = warm_steps
        if steady is not None:
            out["steady_state"] = steady
        if decomp is not None:
            out["decompress"] = decomp
        if value_no_hint is not None:
            out["value_no_hint"] = value_no_hint
        vmod = os.path.join(ROOT, "profiles", "valu_model.json")
        if os.path.exists(vmod) and workload == "config2" and world == 1:
            try:
                vm = json.load(open(vmod))
                floor_ms = float(vm["issue_floor_us_per_launch"]) * 1e-3
                out["roofline_valu"] = {
                    "bound": "valu issue", "kernel": "k_compress<1,5,false,256>",
                    "floor_ms": floor_ms, "kernel_ms_avg": k_avg_ms, "frac": floor_ms / k_avg_ms if k_avg_ms > 0 else None,
                    "source": "profiles/valu_model.json: " + vm.get("source", ""),
                }
            except Exception:
                pass
        if world == 1 and workload == "config2" and not args.no_end_to_end:
            out["chunker_framing"] = chunker_framing(ctx, atsc_amd, torch, dev, d_xs, me, stream)
            out["end_to_end"] = end_to_end(ctx, atsc_amd, xs[0], off, me)
        if world == 1 and workload == "config2" and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(xs[0], off, me)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
