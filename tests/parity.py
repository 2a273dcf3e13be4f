"""Frame-by-frame comparison of the HIP path (through the C ABI) with the CPU oracle.

Bars (north_star / SURVEY 8):
  * Constant / Noop / RLE / Polynomial payloads: byte-identical (integer, byte and f64 index work).
  * FFT payloads: complex-f32 arithmetic that is not bit-reproducible even between two CPUs
    running the reference (rustfft picks its SIMD path at run time).  Same ladder trip count,
    same number of stored bins, same bin positions in the same order, coefficients within
    FFT_COEF_RTOL of the frame's largest bin, reported error within FFT_ERR_ATOL + FFT_ERR_RTOL * error.
  * A frame whose oracle error sits within BOUNDARY_EPS of a ladder/selector threshold may
    legitimately stop one trip apart (f32 noise crossing the threshold); such frames are
    counted separately and must be rare (< BOUNDARY_FRAC of the batch).
"""
import numpy as np

from tests import helpers as H

FFT_COEF_RTOL = 4e-6     # relative to the largest |bin| of the frame (~32 f32 ulp)
FFT_ERR_ATOL = 5e-6      # absolute, on MAPE: f32 transform noise
FFT_ERR_RTOL = 2e-6      # relative, for the large MAPE values of frames with near-zero samples (noise / |g|)
# Plus the decode tolerance carried into the MAPE: two correct f32 implementations may differ by
# nu = (4 + log2 n) f32 ulp of the frame's magnitude + one step of the 5-decimal grid on a
# reconstructed sample (the bar of the decode tests), i.e. by nu / |g| on that sample's term and by at
# most nu * mean(1 / |g|) on the MAPE.  Short frames with a small sample show it: one grid flip on
# g = 0.05 moves the MAPE of a 29-sample frame by 7e-6 (fuzz seeds 616, 662, 670).
POLY_ERR_RTOL = 1e-11    # summation order only
BOUNDARY_EPS = 5e-6
BOUNDARY_FRAC = 0.002
# Frames in which two bins of (near-)equal norm were admitted in another order, or another one of them
# was admitted at the cut: counted on their own and capped like the boundary frames.  Within one spectrum
# bit-equal norms are ordered as the reference's BinaryHeap pops them (tests/test_gpu_parity.py::
# test_fft_tie_order_*); what remains here are norms that tie in one implementation's f32 spectrum and
# differ by an ulp in the other's.
TIE_FRAC = 0.002
INV_G_CLAMP = 1.0e3   # fft_err_noise: a sample within 1e-3 of zero does not widen the tolerance any further


def _near_threshold(err, max_error):
    """True when err is within BOUNDARY_EPS of a value where the reference's decisions flip:
    the FFT loop exit trunc(err*1000) (fft.rs:334), the poly exit round(err,4) vs round(max,3)
    (polynomial.rs:230-231) or the selector's exact filter err <= max_error (frame/mod.rs:128)."""
    if not np.isfinite(err):
        return False
    me = float(np.float32(max_error))
    cands = [me, np.floor(me * 1000.0 + 1) / 1000.0, round(me, 3) + 0.00005]
    return any(abs(err - c) < BOUNDARY_EPS for c in cands)


def compare_frame(oracle, x, max_error, tag, payload, chosen_o, payload_o, report, idx,
                  oracle_errs=None):
    """Returns 'exact' | 'tol' | 'tie' | 'boundary' | 'FAIL:<why>'."""
    if tag != chosen_o:
        errs = oracle_errs or {}
        if any(_near_threshold(e, max_error) for e in errs.values()):
            return "boundary"
        return "FAIL:codec gpu=%d oracle=%d" % (tag, chosen_o)
    if payload == payload_o:
        return "exact"
    if tag != oracle.FFT:
        return "FAIL:bytes differ for codec %d (len %d vs %d)" % (tag, len(payload), len(payload_o))
    fg, mxg, mng = H.parse_fft_payload(payload)
    fo, mxo, mno = H.parse_fft_payload(payload_o)
    if (mxg, mng) != (mxo, mno):
        return "FAIL:fft min/max"
    if len(fg) != len(fo):
        errs = oracle_errs or {}
        if any(_near_threshold(e, max_error) for e in errs.values()):
            return "boundary"
        # Tied norms (documented deviation): the two ladders admitted different bins
        # of equal norm on the way, so their errors -- and the trip at which they stop -- differ.  Seen as:
        # over the common length the norms agree in order while the position sets do not.
        m = min(len(fg), len(fo))
        if m and set(f[0] for f in fg[:m]) != set(f[0] for f in fo[:m]):
            sc = max(np.hypot(r, i) for _, r, i in (fo if len(fo) >= len(fg) else fg))
            ng = [np.hypot(r, i) for _, r, i in fg[:m]]
            no = [np.hypot(r, i) for _, r, i in fo[:m]]
            if max(abs(a - b) for a, b in zip(ng, no)) <= FFT_COEF_RTOL * sc:
                return "tie"
        return "FAIL:fft K gpu=%d oracle=%d" % (len(fg), len(fo))
    scale = max(np.hypot(r, i) for _, r, i in fo) if fo else 1.0
    pos_g = [f[0] for f in fg]
    pos_o = [f[0] for f in fo]
    if pos_g != pos_o:
        # near-equal norms may swap order; accept only if the multisets agree and the swapped
        # bins' norms are within tolerance of each other
        if sorted(pos_g) != sorted(pos_o):
            # last admitted bin may differ when two norms tie at the cut
            ng = {p: np.hypot(r, i) for p, r, i in fg}
            no = {p: np.hypot(r, i) for p, r, i in fo}
            diff = set(pos_g) ^ set(pos_o)
            norms = [ng.get(p, no.get(p)) for p in diff]
            if max(norms) - min(norms) > FFT_COEF_RTOL * scale:
                return "FAIL:fft bin set differs %s" % sorted(diff)
            return "tie"  # different (tied) bins admitted: the reconstructions, hence the errors, differ"
        for (a, ra, ia), (b, rb, ib) in zip(fg, fo):
            if a != b:
                na, nb = np.hypot(ra, ia), np.hypot(rb, ib)
                if abs(na - nb) > FFT_COEF_RTOL * scale:
                    return "FAIL:fft order"

    # A stored position can occur twice: `pos as u16` (fft.rs:242) folds bin p + 65536 of a 131072-sample frame onto p,
    # and a deep ladder admits both.  Entries are therefore matched by (position, occurrence), not by position alone.
    def keyed(fr):
        seen, out = {}, {}
        for p, r, i in fr:
            k = seen.get(p, 0)
            seen[p] = k + 1
            out[(p, k)] = (r, i)
        return out

    dg, do = keyed(fg), keyed(fo)
    for key, (r, i) in do.items():
        if key in dg:
            rg, ig = dg[key]
            if abs(rg - r) > FFT_COEF_RTOL * scale or abs(ig - i) > FFT_COEF_RTOL * scale:
                # the two occurrences of a folded position may have swapped (near-equal norms): try the other one
                alt = dg.get((key[0], 1 - key[1])) if key[1] < 2 else None
                if alt is not None and abs(alt[0] - r) <= FFT_COEF_RTOL * scale and abs(alt[1] - i) <= FFT_COEF_RTOL * scale:
                    continue
                return "FAIL:fft coef pos=%d gpu=(%r,%r) oracle=(%r,%r) scale=%r" % (
                    key[0], rg, ig, r, i, scale)
    return "tol"


def fft_err_noise(fx):
    g = np.abs(np.asarray(fx, dtype=np.float64))
    g = g[np.isfinite(g) & (g > 0)]
    if not g.size:
        return 0.0
    nu = (4 + np.log2(max(len(fx), 2))) * float(g.max()) * 2.0 ** -23 + 1.00001e-5
    return nu * float(np.mean(np.minimum(1.0 / g, INV_G_CLAMP))) * g.size / len(fx)


def compare_batch(oracle, ctx, x, off, compressor, bounded, max_error, level=0, want_diag=False):
    """Runs the batch through the C ABI and the oracle; returns a summary dict."""
    import atsc_amd

    me = float(np.float32(max_error))
    # (the per-frame diagnostics record routes uniform batches to the table-driven kernels: off unless asked for,
    # so that the fixed-length instantiations -- the production path -- are what the comparisons exercise)
    ctx.enable_diag(bool(want_diag))
    rec, rec_off, chosen, err = ctx.compress_host(x, off, compressor, bounded, me, level)
    frames = H.parse_bro_body(rec, with_count=False)
    nf = len(off) - 1
    assert len(frames) == nf
    assert int(rec_off[-1]) == len(rec)
    summary = {"exact": 0, "tol": 0, "tie": 0, "boundary": 0, "tie_frames": [], "boundary_frames": [], "fail": [], "codecs": {},
               "bytes": len(rec),
               "oracle_bytes": 0, "records": rec, "chosen": chosen, "err": err}
    for i, (fs, sc, tag, payload) in enumerate(frames):
        fx = x[int(off[i]):int(off[i + 1])]
        assert fs == 41 and sc == len(fx)
        assert tag == chosen[i]
        errs = {}
        if compressor == atsc_amd.AUTO:
            po, cho, eo = oracle.compress_best(fx, me, level)
            if True:
                _, errs["fft"], _ = oracle.fft_allowed_error(fx, me)
                _, errs["poly"], _ = oracle.polynomial_allowed_error(fx, me)
        else:
            po, eo = oracle.compress(compressor, fx, bounded, me)
            cho = compressor
            errs["e"] = eo
        summary["oracle_bytes"] += len(po) + 1 + len(H_varint(len(fx))) + 1 + len(H_varint(len(po)))
        verdict = compare_frame(oracle, fx, me, tag, payload, cho, po, summary, i, errs)
        summary["codecs"][tag] = summary["codecs"].get(tag, 0) + 1
        if verdict.startswith("FAIL"):
            summary["fail"].append((i, verdict))
        elif verdict == "tie":
            summary["tie"] += 1
            summary["tie_frames"].append(i)
            # A tied frame admitted another bin of (near-)equal norm somewhere: its reconstruction differs by that
            # bin pair's contribution, not by more.  When both ladders stopped at the same K the reported errors are
            # still held together, with the tolerance widened by the decode bar's full noise term (TIE_ERR_FACTOR x).
            if tag == oracle.FFT and chosen_same_k(payload, po):
                tol = TIE_ERR_FACTOR * (FFT_ERR_ATOL + FFT_ERR_RTOL * abs(eo) + fft_err_noise(fx)) + tie_swap_bound(payload, po, fx)
                if not (abs(err[i] - eo) <= tol or (np.isnan(err[i]) and np.isnan(eo))):
                    summary["fail"].append((i, "FAIL:err of a tie frame gpu=%r oracle=%r tol=%r" % (err[i], eo, tol)))
        else:
            summary[verdict] += 1
            if verdict == "boundary":
                summary["boundary_frames"].append(i)
            if verdict != "boundary":
                # reported error: exact codecs report 0.0; lossy within tolerance
                tol = (FFT_ERR_ATOL + FFT_ERR_RTOL * abs(eo) + fft_err_noise(fx)) if tag == oracle.FFT \
                    else max(POLY_ERR_RTOL * abs(eo), 1e-300)
                if not (err[i] == eo or abs(err[i] - eo) <= tol or (np.isnan(err[i]) and np.isnan(eo))):
                    summary["fail"].append((i, "FAIL:err gpu=%r oracle=%r" % (err[i], eo)))
    return summary


TIE_ERR_FACTOR = 8.0


def tie_swap_bound(payload, payload_o, fx):
    """Two ladders that stopped at the same K but admitted different bins of (near-)equal norm reconstruct signals
    that differ, per sample, by at most the swapped bins' amplitudes (2 |c| / L each, L >= n), i.e. their errors by at
    most that times mean(1 / |g|) -- which is large for a frame with samples next to zero (fuzz seed 517: a 64-sample
    frame with MAPE 3.0 against 3.7).  Zero when the two admitted the same set."""
    try:
        fg, fo = H.parse_fft_payload(payload)[0], H.parse_fft_payload(payload_o)[0]
    except Exception:
        return 0.0
    ng = {p: float(np.hypot(r, i)) for p, r, i in fg}
    no = {p: float(np.hypot(r, i)) for p, r, i in fo}
    diff = set(ng) ^ set(no)
    if not diff:
        return 0.0
    amp = sum(ng.get(p, no.get(p)) for p in diff) * 2.0 / max(len(fx), 1)
    g = np.abs(np.asarray(fx, dtype=np.float64))
    g = g[np.isfinite(g) & (g > 0)]
    if not g.size:
        return 0.0
    return amp * float(np.mean(np.minimum(1.0 / g, INV_G_CLAMP)))


def chosen_same_k(payload, payload_o):
    try:
        return len(H.parse_fft_payload(payload)[0]) == len(H.parse_fft_payload(payload_o)[0])
    except Exception:
        return False


def H_varint(v):
    if v < 251:
        return bytes([v])
    if v < 65536:
        return bytes([251]) + int(v).to_bytes(2, "little")
    if v < 2 ** 32:
        return bytes([252]) + int(v).to_bytes(4, "little")
    return bytes([253]) + int(v).to_bytes(8, "little")


def assert_summary(summary, nf, what="", allow_tie=(), allow_boundary=()):
    """Every frame compared, none failed; `tie` and `boundary` verdicts (which skip part of the comparison) are capped:
    batches of 500 frames and more at TIE_FRAC / BOUNDARY_FRAC of the batch, smaller ones at ONE tie and NO boundary
    frame -- unless the test names the frames it expects there (allow_tie / allow_boundary: frame indices).  A
    one-frame batch therefore passes only with verdict exact or tol."""
    msg = "%s: %d frames exact=%d tol=%d tie=%d %s boundary=%d %s fail=%d %s" % (
        what, nf, summary["exact"], summary["tol"], summary.get("tie", 0), summary.get("tie_frames", [])[:8],
        summary["boundary"], summary.get("boundary_frames", [])[:8], len(summary["fail"]), summary["fail"][:8])
    assert not summary["fail"], msg
    ties = [i for i in summary.get("tie_frames", []) if i not in allow_tie]
    bounds = [i for i in summary.get("boundary_frames", []) if i not in allow_boundary]
    if nf >= 500:
        assert len(bounds) <= int(BOUNDARY_FRAC * nf), msg
        assert len(ties) <= int(TIE_FRAC * nf), msg
    else:
        assert len(bounds) == 0, msg
        assert len(ties) <= (1 if nf > 1 else 0), msg
    return msg
