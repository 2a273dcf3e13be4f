"""Multi-GPU plumbing: frames shard across ranks (contiguous ranges, frame order = rank order),
no data-path collective; the only exchange is the gather of the encoded records to rank 0
(SURVEY.md 8(e)).  Works over RCCL ("nccl") on GPUs and over gloo on CPU tensors (tests)."""


def shard_range(n_units, rank, world):
    """Contiguous [begin, end) of `n_units` frames/series owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_records(dist, torch, body, nbytes, rank, world, sizes_buf=None, out=None):
    """Concatenates each rank's first `nbytes` bytes of `body` (uint8 tensor) on rank 0 in rank
    order.  Returns (tensor on rank 0 | None, sizes list).  One 8-byte all-gather of the sizes
    plus point-to-point sends: every peer has its own xGMI link to the root, so the sends
    arrive concurrently."""
    dev = body.device
    if sizes_buf is None:
        sizes_buf = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = nbytes if torch.is_tensor(nbytes) else torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_buf, mine.reshape(1).contiguous())
    sizes = [int(v) for v in sizes_buf.tolist()]
    if rank != 0:
        if sizes[rank]:
            dist.send(body[: sizes[rank]], dst=0)
        return None, sizes
    need = sum(sizes)
    if out is None or out.numel() < need:
        out = torch.empty(max(int(need * 1.25), 16), dtype=torch.uint8, device=dev)
    pos = sizes[0]
    out[:pos].copy_(body[:pos])
    reqs = []
    for r in range(1, world):
        if sizes[r]:
            reqs.append(dist.irecv(out[pos:pos + sizes[r]], src=r))
        pos += sizes[r]
    for q in reqs:
        q.wait()
    return out[:need], sizes
