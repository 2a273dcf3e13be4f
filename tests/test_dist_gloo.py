"""world_size-2 (and 3) gloo tests of the N>1 path on CPU: frame sharding and the gather of
encoded records to rank 0.  The compressor itself needs a GPU, so the per-rank "records" here
are deterministic byte strings; what is checked is ownership, order and byte-exact
concatenation -- the only logic the multi-GPU path adds."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(frame_ids):
    # one variable-length "record" per frame: 1 + (id % 7) bytes, content derived from the id
    out = bytearray()
    for f in frame_ids:
        out += bytes([(f * 31 + k) & 0xFF for k in range(1 + f % 7)])
    return bytes(out)


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atsc_amd import parallel as P

    b, e = P.shard_range(n_frames, rank, world)
    rec = _fake_records(range(b, e))
    body = torch.zeros(len(rec) + 64, dtype=torch.uint8)
    body[: len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
    out, sizes = None, None
    for _ in range(2):  # twice: buffers are reused between steps in bench.py
        out, sizes = P.gather_records(dist, torch, body, len(rec), rank, world)
    if rank == 0:
        q.put((bytes(out.numpy().tobytes()), sizes))
    dist.barrier()
    dist.destroy_process_group()


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atsc_amd import parallel as P

    dev = torch.device("cpu")
    steps = 5
    # per-step payloads differ in content and (slightly) in size, as real steps could
    payload = lambda i: _fake_records(range(rank * 100 + i, rank * 100 + i + 40 + rank))
    pg = P.PipelinedGather(dist, torch, rank, world, dev, len(payload(0)) + 8)
    bufs = [torch.zeros(pg.cap + 64, dtype=torch.uint8) for _ in range(2)]
    for i in range(steps):
        pg.before_produce(i % 2)
        rec = payload(i)
        bufs[i % 2][: len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
        pg.submit(i % 2, bufs[i % 2], torch.tensor([len(rec)], dtype=torch.int64))
    pg.drain()
    if rank == 0:
        segs, sizes = pg.result((steps - 1) % 2)
        q.put((b"".join(bytes(s.numpy().tobytes()) for s in segs), sizes))
    dist.barrier()
    dist.destroy_process_group()


def _overflow_worker(rank, world, port, q):
    """A later step produces more bytes than the agreed capacity: nothing may be truncated silently."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atsc_amd import parallel as P

    dev = torch.device("cpu")
    small = _fake_records(range(rank * 10, rank * 10 + 30))
    pg = P.PipelinedGather(dist, torch, rank, world, dev, len(small))
    # step 1 fits; in step 2 rank 1 alone outgrows the capacity
    big = _fake_records(range(5000)) if rank == 1 else small
    assert len(_fake_records(range(5000))) > pg.room
    bufs = []
    for i, rec in enumerate((small, big)):
        b = torch.zeros(max(len(rec), pg.cap) + 64, dtype=torch.uint8)
        b[: len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
        bufs.append(b)
        pg.before_produce(i % 2)
        pg.submit(i % 2, b, torch.tensor([len(rec)], dtype=torch.int64))
    pg.drain()
    over = pg.overflowed()
    assert over, "the overflow went unnoticed on rank %d" % rank
    assert not pg.overflowed()  # the flag is consumed
    assert bytes(bufs[1][: len(big)].numpy().tobytes()) == big  # the record buffer itself was never written to
    if rank == 0:
        ok = False
        try:
            pg.result(1)
        except RuntimeError:
            ok = True
        assert ok
        segs, sizes = pg.result(0)  # the step that did fit is intact
        q.put(("fit", b"".join(bytes(s.numpy().tobytes()) for s in segs)))
    # the caller's fallback for the step that did not fit: the gather without a capacity
    out, sizes = P.gather_records(dist, torch, bufs[1], len(big), rank, world)
    if rank == 0:
        q.put(("fallback", bytes(out.numpy().tobytes())))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_overflow_is_detected():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got["fit"] == b"".join(_fake_records(range(r * 10, r * 10 + 30)) for r in range(world))
    assert got["fallback"] == _fake_records(range(0, 30)) + _fake_records(range(5000))


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, sizes = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = b"".join(_fake_records(range(r * 100 + 4, r * 100 + 4 + 40 + r)) for r in range(world))
    assert got == want and sum(sizes) == len(want)


@pytest.mark.parametrize("world,n_frames", [(2, 11), (2, 40960), (3, 10)])
def test_gather_records_gloo(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, sizes = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == _fake_records(range(n_frames))
    assert len(sizes) == world and sum(sizes) == len(got)


def _uneven_worker(rank, world, port, q):
    """Root-heavy shards (shard_range_weighted) through the steady-state gather: the capacity is the largest
    PEER's size, the root's own (larger) records never travel, and the stream is the frames in order."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atsc_amd import parallel as P

    n_frames = 1000
    b, e = P.shard_range_weighted(n_frames, rank, world, 1.6)
    rec = _fake_records(range(b, e))
    pg = P.PipelinedGather(dist, torch, rank, world, torch.device("cpu"), len(rec), slack=1.0)
    sizes_all = [None] * world
    dist.all_gather_object(sizes_all, len(rec))
    assert pg.room == ((max(sizes_all[1:]) + 4096 + 8 + 15) & ~15) - 8, (pg.room, sizes_all)  # the peers' sizes alone
    bufs = [torch.zeros(len(rec) + 64, dtype=torch.uint8) for _ in range(2)]
    for i in range(3):
        pg.before_produce(i % 2)
        bufs[i % 2][: len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
        pg.submit(i % 2, bufs[i % 2], torch.tensor([len(rec)], dtype=torch.int64))
    pg.drain()
    assert not pg.overflowed()
    if rank == 0:
        segs, sizes = pg.result(0)
        q.put((b"".join(bytes(s.numpy().tobytes()) for s in segs), sizes, (b, e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gather_root_heavy_shards(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, sizes, root = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == _fake_records(range(1000)) and sum(sizes) == len(got)
    assert root[1] - root[0] > 1000 // world  # the root's shard is the largest


def test_shard_range_weighted_partitions():
    sys.path.insert(0, ROOT)
    from atsc_amd import parallel as P

    import ctypes as C

    import __graft_entry__ as G

    G.build()
    from atsc_amd import capi

    for n in (1, 7, 64, 4096, 40960, 4194304):
        for world in (1, 2, 3, 8):
            for w in (1.0, 1.51, 2.2, 0.5):
                parts = [P.shard_range_weighted(n, r, world, w) for r in range(world)]
                for r in range(world):  # the C ABI cuts at the same frames
                    b, e = C.c_uint64(), C.c_uint64()
                    capi.lib().atsc_shard_range_weighted(n, r, world, int(round(w * 1000)), C.byref(b), C.byref(e))
                    assert (b.value, e.value) == parts[r], (n, world, w, r)
                assert parts[0][0] == 0 and parts[-1][1] == n
                for a, b in zip(parts, parts[1:]):
                    assert a[1] == b[0]
                if world > 1 and n >= 1000:
                    sz = [e - b for b, e in parts]
                    peers = sz[1:]
                    assert max(peers) - min(peers) <= 1
                    assert abs(sz[0] / max(1.0, sum(peers) / len(peers)) - w) < 0.02 * w + 0.01
    # the model of DESIGN.md section 5: configs[3] on 8 GPUs
    w = P.root_weight_for(17.9e-3, 1.9e9, 70e9)
    assert 1.45 < w < 1.58  # (B / R) / T1 = 27.1 ms / 17.9 ms
    b0, e0 = P.shard_range_weighted(4096, 0, 8, w)
    b1, e1 = P.shard_range_weighted(4096, 1, 8, w)
    assert (e0 - b0) > (e1 - b1)


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from atsc_amd import parallel as P

    import ctypes as C

    import __graft_entry__ as G

    G.build()
    from atsc_amd import capi

    for n in (1, 7, 8, 4096, 40960):
        for world in (1, 2, 3, 4, 8):
            parts = [P.shard_range(n, r, world) for r in range(world)]
            for r in range(world):  # the C ABI's split (what a Rust / C host uses) is the same function
                b, e = C.c_uint64(), C.c_uint64()
                capi.lib().atsc_shard_range(n, r, world, C.byref(b), C.byref(e))
                assert (b.value, e.value) == parts[r]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for a, b in zip(parts, parts[1:]):
                assert a[1] == b[0]
            sz = [e - b for b, e in parts]
            assert max(sz) - min(sz) <= 1
