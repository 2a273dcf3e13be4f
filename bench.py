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
    out["compress_frames_f256"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3, "bytes": nbytes}
    rec = bytes(body[:nbytes])
    recarr = np.frombuffer(rec, dtype=np.uint8)
    dec = np.empty(n, dtype=np.float64)
    on = C.c_uint64()

    def df():
        rc = lib.atsc_decompress_frames(ctx._h, recarr.ctypes.data_as(C.POINTER(C.c_uint8)), len(rec), 0,
                                        dec.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(on))
        atsc_amd.capi.check(rc, ctx._h)
        return on.value

    got, dt = timed(df)
    assert got == n
    out["decompress_frames_f256"] = {"value": n / dt / 1e6, "unit": "Msamples/s", "ms": dt * 1e3}
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
    out["note"] = "host buffer -> host buffer, PCIe inclusive, median of %d calls; pageable caller memory" % reps
    return out


def chunker_framing(ctx, atsc_amd, torch, dev, d_xs, me, stream):
    """The same batches in the reference CLI's own framing (OptimizerPlan::get_chunks_sizes, optimizer/mod.rs:78-98:
    80 frames of 131072 samples), device resident: plain calls, and batch after batch through the pipelined entry
    point.  Extra information next to `value` (which is the 256-sample framing BASELINE.json names)."""
    n = d_xs[0].numel()
    sizes = atsc_amd.chunk_sizes(n)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    plan = ctx.plan(off)
    outs = [plan.alloc_outputs(torch, dev) for _ in range(4)]
    R = len(d_xs)
    for i in range(2):
        plan.compress(d_xs[i % R], outs[0], atsc_amd.AUTO, True, me, 0, stream)
    torch.cuda.synchronize()
    reps = 8
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % R], outs[0], atsc_amd.AUTO, True, me, 0, stream)
    torch.cuda.synchronize()
    dt_plain = (time.perf_counter() - t0) / reps
    total = int(outs[0]["rec_off"][-1].item())
    body = outs[0]["body"][:total].cpu().numpy().tobytes()
    for i in range(8):
        plan.compress(d_xs[i % R], outs[i % 4], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
    plan.join(stream)
    torch.cuda.synchronize()
    reps = 16
    t0 = time.perf_counter()
    for i in range(reps):
        plan.compress(d_xs[i % R], outs[i % 4], atsc_amd.AUTO, True, me, 0, stream, pipelined=True)
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
    for _ in range(5):
        dp.decompress(d_body, d_out, stream)
    torch.cuda.synchronize()
    dt_dec = (time.perf_counter() - t0) / 5
    dp.close()
    plan.close()
    return {"frames": len(sizes), "frame_len": int(sizes[0]), "ratio": 8.0 * n / (total + 12),
            "compress": {"value": n / dt_plain / 1e6, "unit": "Msamples/s", "ms": dt_plain * 1e3},
            "compress_pipelined": {"value": n / dt_pipe / 1e6, "unit": "Msamples/s", "ms": dt_pipe * 1e3},
            "decompress": {"value": n / dt_dec / 1e6, "unit": "Msamples/s", "ms": dt_dec * 1e3},
            "note": "device resident; the reference chunker's framing of the same batches (the atsc CLI's framing)"}


ROTATE = 4          # resident batches the timed loop cycles through (N = 1 workload)
C3_SERIES = 4096    # configs[3]: 4096 series x 262144 samples = 2^30
C3_PER = 262144
C3_ERROR_PCT = 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--workload", choices=["auto", "config2", "config3"], default="auto",
                    help="auto: configs[2] at N = 1 (the metric's config), configs[3]'s per-rank share at N > 1")
    ap.add_argument("--series", type=int, default=C3_SERIES, help="config3 only: total series over all ranks")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="pack the records on the codec stream (atsc_compress_plan_dev) instead of "
                         "overlapping them with the next step's codecs (atsc_compress_plan_dev_pipelined)")
    ap.add_argument("--chains", action="store_true",
                    help="N = 1: also time the batches dealt over four independent chains (`value_chains`); the "
                         "large-tier and host-pointer measurements are skipped in such a run (streams stay mapped "
                         "to the few hardware queues once they exist: the measurements would disturb one another)")
    ap.add_argument("--no-decompress", action="store_true", help="skip the decompression measurement")
    ap.add_argument("--adaptive-order", action="store_true",
                    help="start the frames costliest-first instead of in index order (cost = shader clocks of the same "
                         "frame slot in an earlier batch of the same chain); off by default: with two chains in flight the "
                         "drain of one launch is covered by the other chain's kernel and `value` leans on no hint")
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
    if workload == "config2":
        # BASELINE.json configs[2]: one 10,485,760-sample series per batch, classes cycled per 65536-sample
        # block.  ROTATE different series are resident and the timed loop walks through them, so a step never
        # sees the batch whose costs ordered its launch: the hint is a prediction from another batch of the
        # same layout, as in a service that gets the next window of the same series.
        error_pct = ERROR_PCT
        n_local = N_SAMPLES
        xs = [H.synth_series(rank * ROTATE + b, N_SAMPLES) for b in range(ROTATE)]
        d_xs = [torch.from_numpy(v).to(dev) for v in xs]
        units_total = world * N_SAMPLES
        desc = ("BASELINE.json configs[2] per GPU: 10,485,760 f64 samples, 40960 frames x 256, --compressor auto, "
                "e=5%% (max_error=(float)5/100), classes C0-C4 cycled per 65536-sample block, inputs resident in "
                "HBM; the timed loop rotates over %d different resident batches (series ids %d..%d)"
                % (ROTATE, rank * ROTATE, rank * ROTATE + ROTATE - 1))
    else:
        # BASELINE.json configs[3]: 4096 series x 262,144 samples (2^30), class = series % 5, auto e=1%,
        # series s on rank s / (4096 / N): total work fixed, per-rank share 4096 / N series.
        error_pct = C3_ERROR_PCT
        sb, se = parallel.shard_range(args.series, rank, world)
        n_local = (se - sb) * C3_PER
        d_x = torch.empty(n_local, dtype=torch.float64, device=dev)
        for s in range(sb, se):
            d_x[(s - sb) * C3_PER:(s - sb + 1) * C3_PER] = H.synth_series_torch(torch, dev, s, C3_PER, s % 5)
        d_xs = [d_x]
        xs = None
        units_total = args.series * C3_PER
        desc = ("BASELINE.json configs[3]: %d series x 262,144 f64 samples (%d samples over all ranks), "
                "class = series %% 5, --compressor auto, e=1%% (max_error=(float)1/100), 256-sample frames, series "
                "sharded contiguously by rank (%d per rank), inputs resident in HBM, encoded records gathered to "
                "rank 0 every step" % (args.series, units_total, se - sb))
    me = float(np.float32(error_pct) / np.float32(100))
    off = np.arange(0, n_local + 1, FRAME, dtype=np.uint64)
    plan = ctx.plan(off)
    stream = torch.cuda.current_stream().cuda_stream

    # Steady state of a compression service: batch after batch.  Two output sets; the record packing
    # of step i runs on the context's pack stream and overlaps the frame codecs of step i+1 (and, for
    # N > 1, so does the gather of step i, issued from a side stream that waits for that packing).
    pipelined = not args.no_pipeline
    ctx.set_adaptive_order(bool(args.adaptive_order))
    NOUT = 8  # output sets: one per batch in flight (the pipelined calls rotate over up to four chains)
    outs2 = [plan.alloc_outputs(torch, dev) for _ in range(NOUT)]
    pg = None
    gstream = torch.cuda.Stream(device=dev) if world > 1 else None
    R = len(d_xs)

    def step(i, pipe=pipelined):
        o = outs2[i % NOUT]
        if pg is not None:
            pg.before_produce(i % 2)
        plan.compress(d_xs[i % R], o, atsc_amd.AUTO, True, me, 0, stream, pipelined=pipe)
        if world > 1:
            # the path's only exchange: the encoded records go to rank 0
            if pg is not None:
                if pipe:
                    plan.join(gstream.cuda_stream)
                    with torch.cuda.stream(gstream):
                        pg.submit(i % 2, o["body"], o["rec_off"][-1:])
                else:
                    pg.submit(i % 2, o["body"], o["rec_off"][-1:])
                return
            if pipe:
                plan.join(stream)
            if share:  # rehearsal on one GPU: gloo moves host tensors
                nb = int(o["rec_off"][-1].item())
                parallel.gather_records(dist, torch, o["body"][:nb].cpu(), nb, rank, world)
            else:
                parallel.gather_records(dist, torch, o["body"], o["rec_off"][-1:], rank, world)

    def timed_loop(steps, pipe):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, pipe)
        if pipe:
            plan.join(stream)
        if pg is not None:
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
    # (at least one batch through every set of every chain: a set's first batch runs in index order, without the
    # cost hint the steady state has)
    for i in range(max(args.warmup, 2 * NOUT if pipelined else 0, 1 if world > 1 else 0)):
        step(i)
    if pipelined:
        plan.join(stream)
    torch.cuda.synchronize()
    gather_mode = None
    if world > 1:
        gather_mode = "size all-gather + point-to-point sends (gather_records)"
    if world > 1 and not share:
        # segment capacity agreed once from the warm-up result; no host sync inside the timed loop.
        # One untimed trial step validates the asynchronous gather on this backend; any exception
        # falls back to the simple size-exchange + send/recv gather (every rank takes the same branch:
        # the flag is agreed by an all-reduce).
        ok = 1
        try:
            # (one resident batch per rank: every step encodes to the same bytes, so the agreed capacity needs no
            # slack -- at N > 1 the step is bound by the bytes each link carries)
            pg = parallel.PipelinedGather(dist, torch, rank, world, dev, max(body_bytes_b),
                                          slack=1.0 if R == 1 else 1.05)
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
            gather_mode = "one asynchronous fixed-capacity gather per step (PipelinedGather), overlapped with the next step's codecs"
    ctx.set_profiling(True)
    dt = timed_loop(args.steps, pipelined)
    gathered_sizes = None
    if pg is not None:
        if pg.overflowed():
            raise SystemExit("a timed step's records outgrew the gather capacity: the run is void")
        if rank == 0:
            segs, gathered_sizes = pg.result((args.steps - 1) % 2)
            assert len(segs) == world and gathered_sizes[0] == body_bytes_b[(args.steps - 1) % R]
    kern_ms, launches = ctx.profile_read()
    ctx.set_profiling(False)
    # The same steps through plain single-stream calls (atsc_compress_plan_dev: codecs, then the packing, on
    # one stream, frames in index order, no cost hint), same rotation of batches.
    value_no_hint = None
    if world == 1:
        for i in range(2):
            step(i, False)
        torch.cuda.synchronize()
        dt_plain = timed_loop(args.steps, False)
        value_no_hint = units_total * args.steps / dt_plain / 1e6

    def measure_chains():
        # The same batches dealt round-robin over CHAINS independent chains -- a context, a plan and a stream each,
        # nothing waits across them -- as a service with several request queues would drive one GPU.  A frame kernel's
        # workgroups are short; a single hardware queue leaves a freed slot empty for a microsecond or two before the
        # next one starts, and several queues keep the slots fuller (tools/queue_pipe_probe.py).  Reported beside
        # `value`, which stays the single-chain figure the `roofline` launches belong to (--chains).
        chains = None
        if world == 1 and args.chains:
            CH = 4
            cctx = [ctx] + [atsc_amd.Context(0) for _ in range(CH - 1)]
            cplan = [plan] + [c.plan(off) for c in cctx[1:]]
            couts = [outs2] + [[p.alloc_outputs(torch, dev), p.alloc_outputs(torch, dev)] for p in cplan[1:]]
            cstream = [torch.cuda.Stream(device=dev) for _ in range(CH)]

            def cstep(i):
                c = i % CH
                cplan[c].compress(d_xs[i % R], couts[c][(i // CH) % 2], atsc_amd.AUTO, True, me, 0,
                                  cstream[c].cuda_stream, pipelined=True)
            for i in range(8 * CH):
                cstep(i)
            torch.cuda.synchronize()
            ksteps = max(args.steps, 16 * CH)  # its own count: long enough for the chains' ramp and drain not to weigh
            t0 = time.perf_counter()
            for i in range(ksteps):
                cstep(i)
            torch.cuda.synchronize()
            dt_ch = time.perf_counter() - t0
            chains = {"chains": CH, "value": units_total * ksteps / dt_ch / 1e6, "unit": "Msamples/s", "steps": ksteps,
                      "ms_per_step": dt_ch * 1e3 / ksteps,
                      "note": "batches round-robin over %d independent (context, plan, stream) chains on one GPU; each "
                              "chain is the pipelined path `value` times on one" % CH}
            del cplan, couts, cctx
        return chains

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
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dp.decompress(o["body"], d_out, stream)
        torch.cuda.synchronize()
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
                  "config": "BASELINE.json configs[4]: every rank decodes the records of its own shard (device "
                            "resident, frame table parsed once); no collective"}
        del dp, d_out

    body_bytes = float(np.mean(body_bytes_b))
    if world > 1:
        t = torch.tensor([body_bytes_b[0]], dtype=torch.int64, device="cpu" if share else dev)
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        per_rank_bytes = [int(v.item()) for v in lst]
        total_body = float(sum(per_rank_bytes))
    else:
        per_rank_bytes = [int(body_bytes_b[0])]
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
                "parallelism": ("frames sharded by rank over %d processes, backend %s (%d ranks in the group), records "
                                "gathered to rank 0: %s" % (world, backend, dist.get_world_size(), gather_mode))
                               if world > 1 else "single GPU",
                "gathered_bytes_last_step": gathered_sizes,
                "pipeline": ("record packing of step i on the pack stream overlaps the codecs of step i+1 "
                             "(two scratch + output sets)" +
                             ("" if not args.adaptive_order else "; within a launch the frames start costliest "
                              "first, cost = shader clocks the same frame slot took two steps earlier (another batch)"))
                            if pipelined else "single stream",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_compress<1,5,false,256>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel_ms_avg": k_avg_ms,
                "kernel_launches": launches,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if decomp is not None:
            out["decompress"] = decomp
        if value_no_hint is not None:
            out["value_no_hint"] = value_no_hint
        vmod = os.path.join(ROOT, "profiles", "valu_model.json")
        if os.path.exists(vmod) and workload == "config2":
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
        if world == 1 and args.chains:
            out["value_chains"] = measure_chains()
        if world == 1 and workload == "config2" and not args.no_end_to_end and not args.chains:
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
