"""Host -> device copy rates for the host-pointer entry points: pageable hipMemcpy, pinned hipMemcpyAsync,
host memcpy into pinned staging (1..8 threads), and the chunked staging pipeline."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
n = 10485760
x = np.random.rand(n)
dev = torch.device("cuda:0")
d = torch.empty(n, dtype=torch.float64, device=dev)
xt = torch.from_numpy(x)
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts)
dt = t(lambda: d.copy_(xt))
print("pageable H2D 84 MB: %.2f ms  %.1f GB/s" % (dt * 1e3, 8 * n / dt / 1e9))
xp = torch.empty(n, dtype=torch.float64).pin_memory()
xp.copy_(xt)
dt = t(lambda: d.copy_(xp, non_blocking=True))
print("pinned H2D 84 MB: %.2f ms  %.1f GB/s" % (dt * 1e3, 8 * n / dt / 1e9))
for th in (1, 2, 4, 8):
    def work():
        cuts = [n * k // th for k in range(th + 1)]
        ths = [threading.Thread(target=lambda a, b: xp[a:b].copy_(xt[a:b]), args=(cuts[k], cuts[k + 1])) for k in range(th)]
        [q.start() for q in ths]; [q.join() for q in ths]
    dt = t(work)
    print("host memcpy pageable -> pinned, %d threads: %.2f ms  %.1f GB/s" % (th, dt * 1e3, 8 * n / dt / 1e9))
h = torch.empty(n, dtype=torch.float64)
dt = t(lambda: h.copy_(d))
print("pageable D2H 84 MB: %.2f ms  %.1f GB/s" % (dt * 1e3, 8 * n / dt / 1e9))
hp = torch.empty(n, dtype=torch.float64).pin_memory()
dt = t(lambda: hp.copy_(d, non_blocking=True))
print("pinned D2H 84 MB: %.2f ms  %.1f GB/s" % (dt * 1e3, 8 * n / dt / 1e9))
h2 = np.empty(n)
dt = t(lambda: np.copyto(h2, hp.numpy()))
print("host memcpy pinned -> pageable (1 thread, warm pages): %.2f ms" % (dt * 1e3))
