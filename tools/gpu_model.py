"""Numpy model of the device algorithm (development aid, not shipped, not an oracle).

Mirrors the index math of atsc_amd/csrc kernels so it can be checked on CPU against
oracle/ before a GPU is involved: Stockham mixed-radix FFT, ascending-only bitonic
network for arbitrary n, incremental sparse inverse DFT ladder, closed-form Catmull-Rom
segment lookup, RLE size formula.
"""
import math
import numpy as np


def next_size(n):
    def dec(v):
        while v % 2 == 0: v //= 2
        while v % 3 == 0: v //= 3
        return v == 1
    n += 1
    while not dec(n): n += 1
    return n


def radices(L):
    r = []
    while L % 4 == 0: r.append(4); L //= 4
    while L % 2 == 0: r.append(2); L //= 2
    while L % 3 == 0: r.append(3); L //= 3
    assert L == 1
    return r


def twiddle_table(L):
    t = np.arange(L, dtype=np.float64)
    a = 2.0 * math.pi * t / L
    return np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)


def stockham_forward(x, L):
    """x: complex64 length L. returns unnormalised forward DFT (complex64 arithmetic)."""
    c, s = twiddle_table(L)
    X = x.astype(np.complex64).copy()
    Y = np.empty_like(X)
    ncur, st = L, 1
    for r in radices(L):
        m = ncur // r
        nb = L // r
        for t in range(nb):
            q = t % st
            p = t // st
            a = [X[q + st * (p + m * j)] for j in range(r)]
            if r == 2:
                b = [a[0] + a[1], a[0] - a[1]]
            elif r == 4:
                t0, t1, t2 = a[0] + a[2], a[0] - a[2], a[1] + a[3]
                d = a[1] - a[3]
                t3 = np.complex64(complex(d.imag, -d.real))
                b = [t0 + t2, t1 + t3, t0 - t2, t1 - t3]
            else:
                t1 = a[1] + a[2]
                t2 = a[0] - np.complex64(0.5) * t1
                d = a[1] - a[2]
                h = np.float32(0.8660254037844386)
                t3 = np.complex64(complex(h * d.imag, -h * d.real))
                b = [a[0] + t1, t2 + t3, t2 - t3]
            for k in range(r):
                idx = p * k * st
                w = np.complex64(complex(c[idx], -s[idx]))
                Y[q + st * (r * p + k)] = b[k] * w if k else b[k]
        X, Y = Y, X
        ncur = m
        st *= r
    return X


def stockham_generic(X, M, L, rads):
    """length-M Stockham (complex64) using the length-L table (tw index scaled by L/M)."""
    c, s = twiddle_table(L)
    sc = L // M
    X = X.astype(np.complex64).copy(); Y = np.empty_like(X)
    ncur, st = M, 1
    h = np.float32(0.8660254037844386)
    for r in rads:
        m = ncur // r
        for t in range(M // r):
            q = t % st; p = t // st
            a = [X[q + st * (p + m * j)] for j in range(r)]
            if r == 2:
                b = [a[0] + a[1], a[0] - a[1]]
            elif r == 4:
                t0, t1, t2 = a[0] + a[2], a[0] - a[2], a[1] + a[3]
                d = a[1] - a[3]; t3 = np.complex64(complex(d.imag, -d.real))
                b = [t0 + t2, t1 + t3, t0 - t2, t1 - t3]
            else:
                t1 = a[1] + a[2]; t2 = a[0] - np.complex64(0.5) * t1; d = a[1] - a[2]
                t3 = np.complex64(complex(h * d.imag, -h * d.real))
                b = [a[0] + t1, t2 + t3, t2 - t3]
            for k in range(r):
                idx = p * k * st * sc
                w = np.complex64(complex(c[idx], -s[idx]))
                Y[q + st * (r * p + k)] = b[k] * w if k else b[k]
        X, Y = Y, X; ncur = m; st *= r
    return X


def real_fft_forward(g32, L):
    """bins 0..L/2 of the DFT of the real f32 signal, via one complex FFT of length L/2."""
    M = L // 2
    z = (g32[0::2] + 1j * g32[1::2]).astype(np.complex64)
    Z = stockham_generic(z, M, L, radices(M))
    c, s = twiddle_table(L)
    out = np.empty(M + 1, dtype=np.complex64)
    half = np.float32(0.5)
    for k in range(M + 1):
        Zk = Z[k % M]; Zc = np.conj(Z[(M - k) % M])
        a = np.complex64(Zk + Zc); b = np.complex64(Zk - Zc)
        t = np.complex64(complex(b.imag, -b.real))
        w = np.complex64(complex(c[k], -s[k]))
        out[k] = np.complex64(half * a + half * np.complex64(w * t))
    return out


def flip_bitonic_sort(keys):
    """ascending-only bitonic network valid for any length (virtual +inf padding)."""
    a = list(keys)
    n = len(a)
    P = 1
    while P < n: P *= 2
    k = 2
    while k <= P:
        # flip step
        for i in range(P):
            l = i ^ (k - 1)
            if l > i and l < n and a[i] > a[l]:
                a[i], a[l] = a[l], a[i]
        j = k // 4
        while j >= 1:
            for i in range(P):
                l = i ^ j
                if l > i and l < n and a[i] > a[l]:
                    a[i], a[l] = a[l], a[i]
            j //= 2
        k *= 2
    return a


def sat_i32(v):
    if v != v: return 0
    if v >= 2147483647.0: return 2147483647
    if v <= -2147483648.0: return -2147483648
    return int(v)


def fft_ladder(x, max_err):
    """returns (K_stored, err, trips, order positions)"""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    mx32, mn32 = np.float32(x.max()), np.float32(x.min())
    if mx32 == mn32:
        return 0, 0.0, 0, []
    mf = max(3, n // 100)
    if n >= 128:
        L = next_size(n); pre = (L - n) // 2
        g = np.concatenate([np.full(pre, x[0]), x, np.full(L - n - pre, x[-1])])
    else:
        L = n; pre = 0; g = x.copy()
    if L >= 128 and L % 2 == 0:
        X = np.concatenate([real_fft_forward(g.astype(np.float32), L), np.zeros(L - L // 2 - 1, dtype=np.complex64)])
    elif L >= 128:
        X = stockham_forward(g.astype(np.float32).astype(np.complex64), L)
    else:
        X = np.fft.fft(g.astype(np.float32).astype(np.complex128)).astype(np.complex64)
    bins = L // 2 + 1
    re = X.real[:bins].astype(np.float32); im = X.imag[:bins].astype(np.float32)
    norm = np.sqrt(re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2).astype(np.float32)
    keys = [((~(int(norm[i].view(np.uint32)) << 32 | (0xFFFFFFFF - i))) & ((1 << 64) - 1), i) for i in range(bins)]
    order = [i for _, i in flip_bitonic_sort(keys)]
    Z = bins
    for r_, i in enumerate(order):
        if re[i] == 0 and im[i] == 0:
            Z = r_; break
    c, s = twiddle_table(L)
    acc = np.zeros(L, dtype=np.float32)
    dc = np.float32(0)
    j = np.arange(L)
    used = 0
    jump = 0
    err = max_err + 1.0
    trips = 0
    inv = 1.0 / g
    mxd, mnd = float(mx32), float(mn32)
    while sat_i32(max_err * 1000.0) < sat_i32(err * 1000.0):
        trips += 1
        K = min(mf + jump, Z, bins)
        while used < K:
            k = order[used]
            cf = 1.0 if (k == 0 or 2 * k == L) else 2.0
            a = np.float32(cf * float(re[k]) / L); b = np.float32(cf * float(im[k]) / L)
            if k == 0:
                dc = a
            else:
                idx = (j * k) % L
                acc = (acc + a * c[idx]).astype(np.float32)
                acc = (acc - b * s[idx]).astype(np.float32)
            used += 1
        v = (acc + dc).astype(np.float32).astype(np.float64)
        out = np.floor(np.abs(v * 1e5) + 0.5) * np.sign(v) / 1e5
        out = np.where(out > mxd, mxd, np.where(out < mnd, mnd, out))
        err = float(np.sum(np.abs(out - g) * np.abs(inv)) / L)
        if trips <= 17: jump += max(mf // 2, 1)
        elif trips <= 22: jump += max(mf // 10, 1)
        else: break
    return used, err, trips, order[:used]


def poly_ladder(x, max_err):
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    mn, mx = x.min(), x.max()
    if mn == mx: return 1, 0, 0.0, 0
    base = max(3, n // 100)
    r3 = lambda v: math.floor(abs(v * 1000) + 0.5) * (1 if v >= 0 else -1) / 1000
    r4 = lambda v: math.floor(abs(v * 10000) + 0.5) * (1 if v >= 0 else -1) / 10000
    target = r3(max_err)
    err = max_err + 1.0
    jump = 0; it = 0; K = 0; step = 1
    while target < r4(err):
        it += 1
        pts = base + jump
        step = max(n // pts, 1)
        cnt = (n + step - 1) // step
        push = ((cnt - 1) * step != n - 1)
        K = cnt + (1 if push else 0)
        T = lambda k: (n - 1) if (push and k == K - 1) else k * step
        V = lambda k: x[T(k)]
        out = np.empty(n)
        for i in range(n):
            if i == n - 1:
                sv = V(K - 1)
            else:
                seg = min(i // step, K - 2)
                t0, t1 = T(seg), T(seg + 1)
                nt = (i - t0) / (t1 - t0)
                if seg > 0 and K - seg > 2:
                    xm, xp = T(seg - 1), T(seg + 2)
                    t2_ = nt * nt; t3_ = t2_ * nt
                    m0 = (V(seg + 1) - V(seg - 1)) / (t1 - xm) * (t1 - t0)
                    m1 = (V(seg + 2) - V(seg)) / (xp - t0) * (t1 - t0)
                    sv = (V(seg) * (t3_ * 2 - t2_ * 3 + 1) + m0 * (t3_ - t2_ * 2 + nt)
                          + V(seg + 1) * (t2_ * 3 - t3_ * 2) + m1 * (t3_ - t2_))
                else:
                    sv = V(seg) * (1 - nt) + V(seg + 1) * nt
            o = math.floor(abs(sv * 1e5) + 0.5) * (1 if sv >= 0 else -1) / 1e5
            out[i] = mn if o < mn else (mx if o > mx else o)
        err = float(np.sum(np.abs((out - x) / x)) / n)
        if it <= 17: jump += max(n // 10, 1)
        elif it <= 22: jump += max(n // 100, 1)
        elif target > r4(err): break
        else:
            step = 1; K = n; err = 0.0; break
        if K == n:
            err = 0.0; break
    return step, K, err, it
