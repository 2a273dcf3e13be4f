"""How often does a large frame's admitted set hold two bins of bit-equal f32 norm?  (The large tier orders such bins by
position where the reference's BinaryHeap pops them in heap order, DESIGN.md section 4: this is the number behind that
deviation.)  Forced FFT at e = 5 % and 1 % on 131072-sample frames of classes C0-C3 and on the reference's .wbro fixtures
cut by the reference chunker; the stored bins of every GPU payload are checked.  (GPU box only.)"""
import os, sys, glob
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atsc_amd
from tests import helpers as H

ctx = atsc_amd.Context(0)
F = 131072


def tied(payload):
    fr, _, _ = H.parse_fft_payload(payload)
    if not fr:
        return 0, 0
    re = np.array([f[1] for f in fr], dtype=np.float32).astype(np.float64)
    im = np.array([f[2] for f in fr], dtype=np.float32).astype(np.float64)
    nrm = np.sqrt(re * re + im * im).astype(np.float32)  # (float)sqrt((double)re^2 + im^2), as the kernels and glibc hypotf
    u, c = np.unique(nrm.view(np.uint32), return_counts=True)
    return int(np.sum(c[c > 1])), len(fr)


def run(name, x, off, e):
    me = float(np.float32(e) / np.float32(100))
    rec, _, chosen, _ = ctx.compress_host(x, off, atsc_amd.FFT, True, me, 0)
    frames = H.parse_bro_body(rec, with_count=False)
    nt = nb = nbins = 0
    for fs, sc, tag, payload in frames:
        if sc <= 4096:
            continue
        t, k = tied(payload)
        nt += 1 if t else 0
        nb += t
        nbins += k
    print("%-34s e=%d%%: %3d large frames, %3d with a bit-equal pair among their stored bins (%d of %d bins in such pairs)" % (
        name, e, len([f for f in frames if f[1] > 4096]), nt, nb, nbins), flush=True)


for e in (5, 1):
    for klass in (0, 1, 2, 3):
        x = H.synth_series(11 + klass, 16 * F, klass=klass)
        run("synthetic class C%d, 16 x 131072" % klass, x, H.frame_offsets(len(x), F), e)
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "wbros", "*.wbro"))):
        x = H.read_wbro(path)
        sizes = atsc_amd.chunk_sizes(len(x))
        off = np.cumsum([0] + sizes).astype(np.uint64)
        if max(sizes) > 4096:
            run(os.path.basename(path) + " (chunker framing)", x, off, e)
