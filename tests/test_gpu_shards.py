"""The multi-GPU claim on real records: frames are independent (atsc/src/main.rs:146-163,
atsc/src/data.rs:104-109), so a batch sharded over N processes by contiguous frame ranges and gathered in
rank order must be the single-process stream, byte for byte, and decode to the same samples.

Two fresh child processes (spawn) share cuda:0 -- RCCL refuses two ranks on one device, so the records
travel over gloo as in bench.py's ATSC_BENCH_SHARE_GPU rehearsal; sharding, per-rank compression through
the C ABI, the size exchange and the rank-order concatenation are the code the 8-GPU run uses.
Workload: BASELINE.json configs[3]'s shape (series of 262,144 samples, class = series % 5, auto e = 1 %),
64 series, in both framings (256-sample frames; the reference chunker's 131072-sample frames)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ME1 = float(np.float32(1) / np.float32(100))
SERIES, PER = 64, 262144


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _series_block(s0, s1):
    return np.concatenate([H.synth_series(s, PER, klass=s % 5) for s in range(s0, s1)])


def _compress_dev(torch, A, ctx, x, frame, me):
    """x (host) -> (record bytes, chosen) through the device-resident entry point."""
    dev = torch.device("cuda:0")
    off = np.arange(0, len(x) + 1, frame, dtype=np.uint64)
    plan = ctx.plan(off)
    outs = plan.alloc_outputs(torch, dev)
    d_x = torch.from_numpy(x).to(dev)
    plan.compress(d_x, outs, A.AUTO, True, me, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nb = int(outs["rec_off"][-1].item())
    rec = outs["body"][:nb].cpu()
    chosen = outs["chosen"].cpu().numpy().copy()
    plan.close()
    return rec, nb, chosen


def _rank_main(rank, world, port, frame, path, root_weight=1.0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import atsc_amd as A
    from atsc_amd import parallel as P

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = A.Context(0)
    # whole series per rank: SURVEY 8(d) config 4; root_weight != 1: the root-heavy split (rank 0 takes more series)
    s0, s1 = P.shard_range_weighted(SERIES, rank, world, root_weight) if root_weight != 1.0 else P.shard_range(SERIES, rank, world)
    x = _series_block(s0, s1)
    rec, nb, _ = _compress_dev(torch, A, ctx, x, frame, ME1)
    out, sizes = P.gather_records(dist, torch, rec, nb, rank, world)
    if rank == 0:
        with open(path, "wb") as f:
            f.write(bytes(out.numpy().tobytes()))
        with open(path + ".sizes", "w") as f:
            f.write(" ".join(str(s) for s in sizes))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("frame,root_weight", [(256, 1.0), (131072, 1.0), (256, 1.51), (131072, 2.2)])
def test_sharded_stream_is_the_single_process_stream(frame, root_weight, tmp_path):
    """(root_weight != 1: uneven shards -- the root-heavy split of DESIGN.md section 5 -- give the same bytes)"""
    import torch
    import torch.multiprocessing as mp

    world = 2
    path = str(tmp_path / ("gathered_%d.bin" % frame))
    mpc = mp.get_context("spawn")
    port = _free_port()
    procs = [mpc.Process(target=_rank_main, args=(r, world, port, frame, path, root_weight)) for r in range(world)]
    for p in procs:
        p.start()
    # meanwhile, the same batch in this process on one context
    import __graft_entry__ as G

    G.build()
    import atsc_amd as A

    ctx = A.Context(0)
    x = _series_block(0, SERIES)
    rec, nb, chosen = _compress_dev(torch, A, ctx, x, frame, ME1)
    single = bytes(rec.numpy().tobytes())
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    gathered = open(path, "rb").read()
    sizes = [int(v) for v in open(path + ".sizes").read().split()]
    assert len(sizes) == world and all(s > 0 for s in sizes) and sum(sizes) == len(gathered)
    if root_weight != 1.0:
        from atsc_amd import parallel as P

        b0, e0 = P.shard_range_weighted(SERIES, 0, world, root_weight)
        assert e0 - b0 > SERIES - (e0 - b0), "the root carries the larger shard"
    assert gathered == single, "sharded + gathered records differ from the single-process stream"
    # the gathered bytes are a well-formed stream of the right frames and decode to the same samples
    nf = len(x) // frame
    frames = H.parse_bro_body(gathered, with_count=False)
    assert len(frames) == nf and all(f[1] == frame for f in frames)
    a = ctx.decompress_host(gathered)
    b = ctx.decompress_host(single)
    assert len(a) == len(x) and np.array_equal(a, b)
    lossless = np.isin(chosen, (A.CONSTANT, A.RLE))
    xa = x.reshape(nf, frame)
    aa = a.reshape(nf, frame)
    assert np.array_equal(aa[lossless], xa[lossless])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity.log"), "a") as f:
        f.write("sharded stream (2 processes, frame %d, root weight %.2f): %d frames, %d bytes (%s per rank), identical to the "
                "single-process stream; codecs %s\n" % (frame, root_weight, nf, len(gathered), sizes,
                                                        {int(c): int(np.sum(chosen == c)) for c in np.unique(chosen)}))
    ctx.close()


def test_compress_frames_sharded_matches_one_context():
    """atsc_compress_frames_sharded (one process, one context per device, a host thread per shard) with two
    and three contexts -- both on device 0 here -- returns what a single context returns."""
    import ctypes as C

    import __graft_entry__ as G

    G.build()
    import atsc_amd as A

    lib = A.capi.lib()
    x = np.concatenate([H.synth_series(900 + s, 8192, klass=s % 5) for s in range(25)])
    for frame in (256, 2048, 8192):
        off = np.arange(0, len(x) + 1, frame, dtype=np.uint64)
        nf = len(off) - 1
        one = A.Context(0)
        want, want_off, want_ch, want_err = one.compress_host(x, off, A.AUTO, True, float(np.float32(0.05)), 0)
        one.close()
        for n_ctx in (2, 3):
            ctxs = [A.Context(0) for _ in range(n_ctx)]
            arr = (C.c_void_p * n_ctx)(*[c._h for c in ctxs])
            cap = int(nf * (48 + 17 * frame))
            body = np.empty(cap, dtype=np.uint8)
            blen = C.c_uint64()
            rec = np.zeros(nf + 1, dtype=np.uint64)
            ch = np.zeros(nf, dtype=np.uint8)
            err = np.zeros(nf, dtype=np.float64)
            rc = lib.atsc_compress_frames_sharded(
                arr, n_ctx, x.ctypes.data_as(C.POINTER(C.c_double)), off.ctypes.data_as(C.POINTER(C.c_uint64)), nf,
                A.AUTO, 1, C.c_float(np.float32(0.05)), 0, body.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
                C.byref(blen), rec.ctypes.data_as(C.POINTER(C.c_uint64)), ch.ctypes.data_as(C.POINTER(C.c_uint8)),
                err.ctypes.data_as(C.POINTER(C.c_double)))
            assert rc == 0
            assert bytes(body[:blen.value]) == want, (frame, n_ctx)
            assert np.array_equal(rec, want_off) and np.array_equal(ch, want_ch)
            assert np.array_equal(err, want_err, equal_nan=True)
            for c in ctxs:
                c.close()


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[2]
import torch, torch.distributed as dist
from atsc_amd import parallel
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
body = torch.arange(0, 5000, dtype=torch.int64, device=dev).to(torch.uint8)
nb = torch.tensor([4321], dtype=torch.int64, device=dev)
out, sizes = parallel.gather_records(dist, torch, body, nb, 0, 1)
assert sizes == [4321] and torch.equal(out, body[:4321])
pg = parallel.PipelinedGather(dist, torch, 0, 1, dev, 4321)
side = torch.cuda.Stream(device=dev)
for step in range(4):
    slot = step % 2
    pg.before_produce(slot)
    with torch.cuda.stream(side):
        pg.submit(slot, body, nb)
pg.drain()
torch.cuda.synchronize()
assert not pg.overflowed()
segs, sizes = pg.result(1)
assert sizes == [4321] and torch.equal(segs[0], body[:4321])
dist.barrier()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


def test_record_gather_runs_on_the_rccl_backend_with_one_rank():
    """The exchange `bench.py --gpus N` uses -- gather_records and PipelinedGather (an all-reduce, an 8-byte all-gather,
    one asynchronous fixed-capacity gather per step issued from a side stream) -- on the backend the 8-GPU run uses
    ("nccl" = RCCL), with the one rank a one-GPU box allows: the calls, dtypes and stream use are RCCL's to accept or
    refuse even when nothing crosses a link.  (Two ranks on one device are refused by RCCL: the two-process tests
    above carry their bytes over gloo.)"""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT, str(_free_port())], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-300:], r.stderr[-1500:])
