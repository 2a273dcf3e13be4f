"""Multi-GPU plumbing: frames shard across ranks (contiguous ranges, frame order = rank order),
no data-path collective; the only exchange is the gather of the encoded records to rank 0
(SURVEY.md 8(e)).  Works over RCCL ("nccl") on GPUs and over gloo on CPU tensors (tests)."""


def shard_range(n_units, rank, world):
    """Contiguous [begin, end) of `n_units` frames/series owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_range_weighted(n_units, rank, world, root_weight=1.0):
    """Contiguous [begin, end) of `n_units` for `rank` when rank 0 carries `root_weight` times a peer's share.
    The records of every peer cross ONE xGMI link to the root while the root's own records stay where they are,
    so when the gather bounds a step the root can take more frames than a peer (DESIGN.md section 5:
    root_weight = (B / R) / T1 for record bytes B per step, link rate R and single-GPU codec time T1, when that
    exceeds 1).
    root_weight = 1 is shard_range; the split is the C ABI's atsc_shard_range_weighted (integer arithmetic on
    thousandths, so every host language cuts at the same frame)."""
    w0 = max(1, int(round(float(root_weight) * 1000.0)))
    total = w0 + 1000 * (world - 1)

    def cut(r):  # frames in front of rank r
        if r <= 0:
            return 0
        if r >= world:
            return n_units
        return (n_units * (w0 + 1000 * (r - 1))) // total

    return cut(rank), cut(rank + 1)


def root_weight_for(codec_s_all, record_bytes, link_bytes_per_s):
    """The root's weight that equalises the root's codec time (share_0 * T1) with a peer's time on its link
    (share_p * B / R), from the single-GPU codec time T1 of the whole job, the record bytes B of one step and the
    per-link rate R: share_0 / share_p = (B / R) / T1.  Never below 1: when the codecs bound the step the even
    split is the best one."""
    if codec_s_all <= 0:
        return 1.0
    return max(1.0, (record_bytes / link_bytes_per_s) / codec_s_all)


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
    every step issues ONE asynchronous `gather` of `cap` bytes per rank, with no host synchronisation.
    Each rank stages its records in a segment of its own whose last 8 bytes carry the step's byte
    count (the record buffer itself is never written to), so the collective count per step is one and
    the communicator's stream keeps up with the codecs.  A step whose records do not fit the agreed
    capacity is not truncated silently: the sending rank raises a sticky device-side flag,
    `overflowed()` (a collective) reports it on every rank, and the caller repeats that exchange with
    `gather_records`, which has no capacity.  Rank 0 ends up with `world` segments at stride `cap`: the
    encoded stream is the concatenation of segment r's first size_r bytes, in rank order = frame order."""

    TAIL = 8  # bytes at the end of every segment that carry the step's byte count (int64)

    def __init__(self, dist, torch, rank, world, device, observed_bytes, slack=1.05, depth=2):
        self.dist, self.torch, self.rank, self.world, self.depth = dist, torch, rank, world, depth
        # The capacity is the largest PEER's size: the root's own records never travel (they are read where the
        # codecs left them), so a root that carries more frames than a peer (shard_range_weighted) does not make
        # every link carry the root's size.
        mine = torch.tensor([int(observed_bytes) if rank != 0 else 0], dtype=torch.int64, device=device)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        self.cap = (int(int(mine.item()) * slack) + 4096 + self.TAIL + 15) & ~15
        self.room = self.cap - self.TAIL  # record bytes a segment can carry
        self.seg = [None] * depth
        self.work = [None] * depth
        self.own = [None] * depth  # root: (record buffer, byte count tensor) of the step in each slot
        self.stage = [torch.empty(self.cap, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.over = torch.zeros(1, dtype=torch.int64, device=device)  # sticky: some step did not fit
        self.submitted_bytes = 0  # bytes this rank put on its link (peers) since construction
        if rank == 0:
            for d in range(depth):
                self.seg[d] = [torch.empty(self.cap, dtype=torch.uint8, device=device) for _ in range(world)]

    def before_produce(self, slot):
        """Call before reusing slot `slot`: waits for the gather that last read its staging segment."""
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def submit(self, slot, body, nbytes_tensor):
        """body: this rank's uint8 record buffer; nbytes_tensor: int64[1] on the same device (the number
        of record bytes in it).  Everything is enqueued on the current stream; nothing blocks the host.
        The root keeps a reference to its own buffer instead of staging it: do not overwrite that buffer
        before `result(slot)` has been read."""
        torch = self.torch
        st = self.stage[slot]
        nb = nbytes_tensor.reshape(1).to(torch.int64)
        if self.rank == 0:
            self.own[slot] = (body, nb.clone())
        else:
            k = min(self.room, body.numel())
            st[:k].copy_(body[:k])  # bytes beyond the step's size are never read by anybody
            self.over.copy_(torch.maximum(self.over, (nb > self.room).to(torch.int64)))
            self.submitted_bytes += self.cap
        st[self.room:].view(torch.int64).copy_(nb)
        self.work[slot] = self.dist.gather(st, gather_list=self.seg[slot] if self.rank == 0 else None,
                                           dst=0, async_op=True)

    def drain(self):
        for slot in range(self.depth):
            self.before_produce(slot)

    def overflowed(self):
        """Collective: True on every rank if any rank's records outgrew the agreed capacity in any step
        since the last call (those steps' segments are truncated and must not be used)."""
        t = self.over.clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.over.zero_()
        return bool(int(t.item()))

    def result(self, slot):
        """(rank 0, after drain) -> list of byte views in rank order, and their sizes."""
        sizes = [int(self.seg[slot][r][self.room:].view(self.torch.int64).item()) for r in range(self.world)]
        if not all(0 <= s <= self.room for s in sizes[1:]):
            raise RuntimeError("a segment outgrew the agreed capacity (%r > %d): repeat the exchange with "
                               "gather_records" % (sizes, self.room))
        body0, nb0 = self.own[slot]
        sizes[0] = int(nb0.item())
        return [body0[: sizes[0]]] + [self.seg[slot][r][: sizes[r]] for r in range(1, self.world)], sizes
