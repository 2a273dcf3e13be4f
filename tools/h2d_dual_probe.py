"""Host-to-device rate from pinned memory: one stream against two and four streams copying disjoint pieces at once
(is one DMA engine the limit, or the link?).  GPU box only."""
import time
import torch

N = 84 * 1024 * 1024
h = torch.empty(N, dtype=torch.uint8).pin_memory()
d = torch.empty(N, dtype=torch.uint8, device="cuda:0")
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    piece = 8 * 1024 * 1024
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for o in range(0, N, piece):
            with torch.cuda.stream(streams[k % ns]):
                d[o:o + piece].copy_(h[o:o + piece], non_blocking=True)
            k += 1
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("h2d %d stream(s): %.3f ms  %.1f GB/s" % (ns, best * 1e3, N / best / 1e9))
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for o in range(0, N, piece):
            with torch.cuda.stream(streams[k % ns]):
                h[o:o + piece].copy_(d[o:o + piece], non_blocking=True)
            k += 1
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("d2h %d stream(s): %.3f ms  %.1f GB/s" % (ns, best * 1e3, N / best / 1e9))
