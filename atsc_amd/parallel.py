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


class PipelinedGather:
    """Steady-state gather for a loop that produces one record buffer per step on every rank.

    Segment capacity is agreed once (max over ranks of the observed size, plus slack); after that
    every step issues ONE asynchronous `gather` of `cap` bytes per rank, with no host synchronisation:
    the byte count of the step rides in the last 8 bytes of the segment (a device-side copy into the
    record buffer, which is far larger than `cap`), so the collective count per step is one and the
    communicator's stream keeps up with the codecs.  The producer moves on to the other buffer set
    (depth 2).  Rank 0 ends up with `world` segments at stride `cap`: the encoded stream is the
    concatenation of segment r's first size_r bytes, in rank order = frame order."""

    TAIL = 8  # bytes at the end of every segment that carry the step's byte count (int64)

    def __init__(self, dist, torch, rank, world, device, observed_bytes, slack=1.05, depth=2):
        self.dist, self.torch, self.rank, self.world, self.depth = dist, torch, rank, world, depth
        t = torch.tensor([int(observed_bytes)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.cap = (int(int(t.item()) * slack) + 4096 + self.TAIL + 15) & ~15
        self.seg = [None] * depth
        self.work = [None] * depth
        if rank == 0:
            for d in range(depth):
                self.seg[d] = [torch.empty(self.cap, dtype=torch.uint8, device=device) for _ in range(world)]

    def before_produce(self, slot):
        """Call before overwriting buffer set `slot`: waits for the gather that last read it."""
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def submit(self, slot, body, nbytes_tensor):
        """body: this rank's uint8 buffer (>= cap bytes, 8-byte aligned); nbytes_tensor: int64[1] on
        the same device.  The records must end before cap - TAIL (checked by `result`)."""
        tail = body[self.cap - self.TAIL: self.cap].view(self.torch.int64)
        tail.copy_(nbytes_tensor.reshape(1))
        self.work[slot] = self.dist.gather(body[: self.cap], gather_list=self.seg[slot] if self.rank == 0 else None,
                                           dst=0, async_op=True)

    def drain(self):
        for slot in range(self.depth):
            self.before_produce(slot)

    def result(self, slot):
        """(rank 0, after drain) -> list of byte views in rank order, and their sizes."""
        sizes = [int(self.seg[slot][r][self.cap - self.TAIL:].view(self.torch.int64).item()) for r in range(self.world)]
        assert all(0 <= s <= self.cap - self.TAIL for s in sizes), "a segment outgrew the agreed capacity"
        return [self.seg[slot][r][: sizes[r]] for r in range(self.world)], sizes
