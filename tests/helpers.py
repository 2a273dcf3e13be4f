"""Test-side helpers: independent WBRO/CSV/BRO parsers (python) and the shared
synthetic generator.  Independent of both the oracle
and the product so they can cross-check either."""
import math
import os
import struct

import numpy as np

MASK64 = (1 << 64) - 1


def read_wbro(path):
    """WBRO = "WBRO0000WBRO" + rkyv 0.7.44 archive (wavbrro/src/wavbrro.rs:37-46)."""
    raw = open(path, "rb").read()
    assert raw[0:4] == b"WBRO" and raw[8:12] == b"WBRO"
    body = raw[12:]
    root = len(body) - 16
    rel, n_chunks, sample_count, bitdepth = struct.unpack_from("<iIIB", body, root)
    assert bitdepth == 5
    arr = root + rel
    out = []
    for c in range(n_chunks):
        ent = arr + 8 * c
        crel, clen = struct.unpack_from("<iI", body, ent)
        start = ent + crel
        out.append(np.frombuffer(body, dtype="<f8", count=clen, offset=start))
    data = np.concatenate(out) if out else np.zeros(0)
    assert len(data) == sample_count
    return data.astype(np.float64)


def read_csv_values(path, header=True, value_field="value"):
    lines = open(path).read().splitlines()
    col = 0
    if header:
        names = lines[0].split(",")
        col = names.index(value_field)
        lines = lines[1:]
    return np.array([float(l.split(",")[col]) for l in lines if l], dtype=np.float64)


def varint_decode(b, pos):
    t = b[pos]
    if t < 251:
        return t, pos + 1
    if t == 251:
        return struct.unpack_from("<H", b, pos + 1)[0], pos + 3
    if t == 252:
        return struct.unpack_from("<I", b, pos + 1)[0], pos + 5
    if t == 253:
        return struct.unpack_from("<Q", b, pos + 1)[0], pos + 9
    raise ValueError("bad varint tag %d" % t)


def parse_bro(bro):
    """Returns (frame_count_u8, [(frame_size, samples, tag, payload bytes)])  (SURVEY App. A)."""
    assert bro[0:4] == b"BRRO"
    assert struct.unpack_from("<I", bro, 4)[0] == 1
    fc = bro[8]
    return fc, parse_bro_body(bro[9:])


def parse_bro_body(body, with_count=True):
    pos = 0
    frames = []
    if with_count:
        n, pos = varint_decode(body, pos)
    else:
        n = None
    while (n is None and pos < len(body)) or (n is not None and len(frames) < n):
        fs, pos = varint_decode(body, pos)
        sc, pos = varint_decode(body, pos)
        tag, pos = varint_decode(body, pos)
        ln, pos = varint_decode(body, pos)
        frames.append((fs, sc, tag, bytes(body[pos:pos + ln])))
        pos += ln
    assert pos == len(body)
    return frames


def parse_fft_payload(p):
    """-> (freqs [(pos, re_f32, im_f32)], max_f32, min_f32)   fft.rs:119-130"""
    assert p[0] == 15
    k, pos = varint_decode(p, 1)
    freqs = []
    for _ in range(k):
        fp, pos = varint_decode(p, pos)
        re, im = struct.unpack_from("<ff", p, pos)
        pos += 8
        freqs.append((fp, re, im))
    mx, mn = struct.unpack_from("<ff", p, pos)
    assert pos + 8 == len(p)
    return freqs, mx, mn


def parse_poly_payload(p):
    """-> dict(id, bitdepth, npoints, step)   polynomial.rs:54-87"""
    pid, pos = varint_decode(p, 0)
    bd, pos = varint_decode(p, pos)
    cnt, pos = varint_decode(p, pos)
    return {"id": pid, "bitdepth": bd, "npoints": cnt, "step": p[-1],
            "min": struct.unpack_from("<d", p, len(p) - 17)[0],
            "max": struct.unpack_from("<d", p, len(p) - 9)[0]}


# --------------------------------------------------------------------------
# synthetic generator -- SURVEY.md section 8(d) (splitmix64 + five classes)
# --------------------------------------------------------------------------

def _splitmix_stream(seed, count):
    """vectorised splitmix64: returns `count` u64 outputs starting from state seed."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed, count):
    return (_splitmix_stream(seed, count) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def synth_series(series_id, n, klass=None, block=65536, class_shift=0):
    """Series `series_id`, n samples.  klass None => class = (i // block + class_shift) % 5 (single-series
    configs; class_shift moves the block -> class mapping, bench.py uses it to give every resident batch
    another layout); else fixed class.  One uniform draw is consumed per sample index for every class so
    the stream position is a function of i alone."""
    seed = (0xA75C000000000000 + series_id) & MASK64
    i = np.arange(n, dtype=np.float64)
    u = _uniform(seed, n)
    out = np.empty(n, dtype=np.float64)
    if klass is None:
        cls = (np.arange(n) // block + class_shift) % 5
    else:
        cls = np.full(n, klass)
    two_pi = 2.0 * math.pi
    m = cls == 0
    if m.any():
        out[m] = (1000.0 + 200.0 * np.sin(two_pi * i[m] / 97.0)
                  + 50.0 * np.sin(two_pi * i[m] / 1013.0 + 0.3) + 5.0 * (u[m] - 0.5))
    m = cls == 1
    if m.any():
        out[m] = (1000.0 + 300.0 * np.sin(two_pi * i[m] / 41.0) + 200.0 * np.sin(two_pi * i[m] / 11.7)
                  + 100.0 * np.sin(two_pi * i[m] / 5.3) + 60.0 * np.sin(two_pi * i[m] / 2.9)
                  + 20.0 * (u[m] - 0.5))
    m = cls == 2
    if m.any():
        ii = np.arange(n)[m]
        out[m] = np.floor(500.0 + (ii // 64) % 400 + ((ii % 64) * 3) / 4.0)
    m = cls == 3
    if m.any():
        # gauge with runs: value 100+floor(u*41) held for 16+floor(u'*240) samples
        idxs = np.nonzero(m)[0]
        vals = np.empty(len(idxs))
        k = 0
        pos = 0
        while pos < len(idxs):
            uu = u[idxs[pos]]
            ul = u[idxs[min(pos + 1, len(idxs) - 1)]]
            v = 100.0 + math.floor(uu * 41.0)
            ln = 16 + int(math.floor(ul * 240.0))
            vals[pos:pos + ln] = v
            pos += ln
            k += 1
        out[m] = vals
    m = cls == 4
    if m.any():
        out[m] = 42.0 + (series_id % 7)
    return out


def synth_series_torch(torch, device, series_id, n, klass):
    """The same generator on the GPU for the billion-sample configs (bench.py, N > 1): classes 0-2 are
    evaluated with torch on `device` (identical splitmix64 stream; sin() may differ from numpy's in the
    last place, which no test depends on), classes 3 and 4 come from synth_series."""
    if klass in (3, 4):
        return torch.from_numpy(synth_series(series_id, n, klass=klass)).to(device)

    def s64(v):  # two's-complement image of a u64 constant
        v &= MASK64
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    seed = (0xA75C000000000000 + series_id) & MASK64
    idx = torch.arange(1, n + 1, dtype=torch.int64, device=device)
    z = idx * s64(0x9E3779B97F4A7C15) + s64(seed)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    u = lsr(z, 11).to(torch.float64) * (2.0 ** -53)
    i = torch.arange(n, dtype=torch.float64, device=device)
    two_pi = 2.0 * math.pi
    if klass == 0:
        return (1000.0 + 200.0 * torch.sin(two_pi * i / 97.0) + 50.0 * torch.sin(two_pi * i / 1013.0 + 0.3)
                + 5.0 * (u - 0.5))
    if klass == 1:
        return (1000.0 + 300.0 * torch.sin(two_pi * i / 41.0) + 200.0 * torch.sin(two_pi * i / 11.7)
                + 100.0 * torch.sin(two_pi * i / 5.3) + 60.0 * torch.sin(two_pi * i / 2.9) + 20.0 * (u - 0.5))
    ii = torch.arange(n, dtype=torch.int64, device=device)
    return torch.floor(500.0 + ((ii // 64) % 400).to(torch.float64) + ((ii % 64) * 3).to(torch.float64) / 4.0)


def frame_offsets(n, frame):
    offs = list(range(0, n, frame)) + [n]
    return np.array(offs, dtype=np.uint64)


def mape(orig, gen):
    orig = np.asarray(orig, dtype=np.float64)
    gen = np.asarray(gen, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.sum(np.abs((gen - orig) / orig)) / len(orig))


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
