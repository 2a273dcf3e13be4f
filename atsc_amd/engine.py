"""Thin Python plumbing over the C ABI: device memory and streams come from PyTorch-ROCm,
everything else happens inside libatsc_hip.so.  No compression logic lives here."""
import ctypes as C
import weakref

import numpy as np

from . import capi


def _u64(arr):
    a = np.ascontiguousarray(np.asarray(arr, dtype=np.uint64))
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


class Context:
    """One atsc_ctx (one per host thread / per GPU rank)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        capi.check(capi.lib().atsc_ctx_create(C.byref(self._h), int(device)))
        self.device = int(device)
        self._children = weakref.WeakSet()  # plans own device blocks of this context's pool

    def close(self):
        if self._h:
            for ch in list(self._children):  # plans go first: atsc_ctx_destroy releases the pool
                ch.close()
            capi.lib().atsc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- diagnostics -----------------------------------------------------------------------
    def enable_diag(self, on=True):
        capi.check(capi.lib().atsc_ctx_enable_diag(self._h, int(on)), self._h)

    def last_diag(self, n_frames):
        arr = (capi.FrameDiag * n_frames)()
        capi.check(capi.lib().atsc_ctx_last_diag(self._h, arr, n_frames), self._h)
        return arr

    def set_adaptive_order(self, on=True):
        """Pipelined calls start a class's costliest frames (previous batch's clocks) first."""
        capi.check(capi.lib().atsc_ctx_set_adaptive_order(self._h, int(on)), self._h)

    def set_chains(self, n):
        """Chains (context streams + scratch sets) the pipelined calls rotate over: 1..4."""
        capi.check(capi.lib().atsc_ctx_set_chains(self._h, int(n)), self._h)

    def set_profiling(self, on=True):
        capi.check(capi.lib().atsc_ctx_set_profiling(self._h, int(on)), self._h)

    def profile_read(self):
        """-> (summed ms of the dominant k_compress launches, number of launches)"""
        ms = C.c_double()
        cnt = C.c_uint64()
        capi.check(capi.lib().atsc_ctx_profile_read(self._h, C.byref(ms), C.byref(cnt)), self._h)
        return ms.value, cnt.value

    # ---- host-pointer convenience ----------------------------------------------------------
    def compress_host(self, samples, frame_off, compressor=capi.AUTO, bounded=True, max_error=0.03,
                      level=0):
        """-> (records bytes, rec_off uint64[n+1], chosen uint8[n], err float64[n])"""
        x = np.ascontiguousarray(np.asarray(samples, dtype=np.float64))
        off, poff = _u64(frame_off)
        nf = len(off) - 1
        lens = np.diff(off.astype(np.int64))
        cap = int(np.sum(48 + np.where(lens > 65535, 17, 14) * lens))  # sum of atsc_payload_bound_bytes + 16
        body = np.empty(max(cap, 16), dtype=np.uint8)
        blen = C.c_uint64()
        rec = np.zeros(nf + 1, dtype=np.uint64)
        chosen = np.zeros(nf, dtype=np.uint8)
        err = np.zeros(nf, dtype=np.float64)
        rc = capi.lib().atsc_compress_frames(
            self._h, x.ctypes.data_as(C.POINTER(C.c_double)), poff, nf, int(compressor),
            int(bool(bounded)), C.c_float(np.float32(max_error)), int(level),
            body.ctypes.data_as(C.POINTER(C.c_uint8)), body.size, C.byref(blen),
            rec.ctypes.data_as(C.POINTER(C.c_uint64)), chosen.ctypes.data_as(C.POINTER(C.c_uint8)),
            err.ctypes.data_as(C.POINTER(C.c_double)))
        capi.check(rc, self._h)
        return bytes(body[: blen.value]), rec, chosen, err

    def decompress_host(self, records, has_count=False):
        b = np.frombuffer(bytes(records), dtype=np.uint8)
        dp = DPlan(self, records, has_count)
        n = dp.n_samples
        dp.close()
        out = np.empty(max(n, 1), dtype=np.float64)
        on = C.c_uint64()
        rc = capi.lib().atsc_decompress_frames(
            self._h, b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), int(has_count),
            out.ctypes.data_as(C.POINTER(C.c_double)), out.size, C.byref(on))
        capi.check(rc, self._h)
        return out[: on.value]

    # ---- device-resident path --------------------------------------------------------------
    def plan(self, frame_off):
        return Plan(self, frame_off)


class Plan:
    """Frame layout of one batch, uploaded once (atsc_plan)."""

    def __init__(self, ctx, frame_off):
        self.ctx = ctx
        off, poff = _u64(frame_off)
        self._h = C.c_void_p()
        capi.check(capi.lib().atsc_plan_create(ctx._h, poff, len(off) - 1, C.byref(self._h)), ctx._h)
        self.n_frames = int(capi.lib().atsc_plan_n_frames(self._h))
        self.n_samples = int(capi.lib().atsc_plan_n_samples(self._h))
        self.body_bound = int(capi.lib().atsc_plan_body_bound(self._h))
        ctx._children.add(self)

    def close(self):
        if self._h:
            capi.lib().atsc_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc_outputs(self, torch, device):
        """Device buffers for one compress call (torch tensors own the memory)."""
        return {
            "body": torch.empty(max(self.body_bound, 16), dtype=torch.uint8, device=device),
            "rec_off": torch.empty(self.n_frames + 1, dtype=torch.int64, device=device),
            "chosen": torch.empty(self.n_frames, dtype=torch.uint8, device=device),
            "err": torch.empty(self.n_frames, dtype=torch.float64, device=device),
        }

    def compress(self, d_samples, outs, compressor=capi.AUTO, bounded=True, max_error=0.03, level=0,
                 stream=0, pipelined=False):
        """Enqueues the compression of every frame on `stream` (a raw hipStream_t or 0).
        pipelined=True: atsc_compress_plan_dev_pipelined -- consecutive calls rotate over the plan's chains
        (streams of the context's own); `join(stream)` orders a stream after their records,
        `input_release(stream)` after their last read of d_samples."""
        assert d_samples.dtype.is_floating_point and d_samples.element_size() == 8
        assert d_samples.is_contiguous() and d_samples.numel() >= self.n_samples
        fn = capi.lib().atsc_compress_plan_dev_pipelined if pipelined else capi.lib().atsc_compress_plan_dev
        rc = fn(
            self.ctx._h, self._h, C.c_void_p(d_samples.data_ptr()), int(compressor),
            int(bool(bounded)), C.c_float(np.float32(max_error)), int(level),
            C.c_void_p(outs["body"].data_ptr()), outs["body"].numel(),
            C.c_void_p(outs["rec_off"].data_ptr()), C.c_void_p(outs["chosen"].data_ptr()),
            C.c_void_p(outs["err"].data_ptr()), C.c_void_p(stream))
        capi.check(rc, self.ctx._h)

    def join(self, stream=0):
        """`stream` waits on the device for every pipelined batch enqueued so far (records packed)."""
        capi.check(capi.lib().atsc_plan_join(self.ctx._h, self._h, C.c_void_p(stream)), self.ctx._h)

    def input_release(self, stream=0):
        """`stream` waits on the device until the pipelined calls enqueued so far have read their inputs."""
        capi.check(capi.lib().atsc_plan_input_release(self.ctx._h, self._h, C.c_void_p(stream)), self.ctx._h)


class DPlan:
    """Parsed frame table of encoded records (atsc_dplan)."""

    def __init__(self, ctx, records, has_count=False):
        self.ctx = ctx
        self._bytes = np.frombuffer(bytes(records), dtype=np.uint8)
        self._h = C.c_void_p()
        capi.check(capi.lib().atsc_dplan_create(
            ctx._h, self._bytes.ctypes.data_as(C.POINTER(C.c_uint8)), len(self._bytes),
            int(has_count), C.byref(self._h)), ctx._h)
        self.n_frames = int(capi.lib().atsc_dplan_n_frames(self._h))
        self.n_samples = int(capi.lib().atsc_dplan_n_samples(self._h))
        ctx._children.add(self)

    def close(self):
        if self._h:
            capi.lib().atsc_dplan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decompress(self, d_body, d_out, stream=0):
        rc = capi.lib().atsc_decompress_plan_dev(
            self.ctx._h, self._h, C.c_void_p(d_body.data_ptr()), C.c_void_p(d_out.data_ptr()),
            C.c_void_p(stream))
        capi.check(rc, self.ctx._h)


# ---- host-only helpers (no GPU) ------------------------------------------------------------
def chunk_sizes(n):
    cnt = capi.lib().atsc_chunk_sizes(n, None, 0)
    out = np.zeros(max(cnt, 1), dtype=np.uint64)
    capi.lib().atsc_chunk_sizes(n, out.ctypes.data_as(C.POINTER(C.c_uint64)), cnt)
    return [int(v) for v in out[:cnt]]


def clean_data(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    out = np.empty(max(len(a), 1), dtype=np.float64)
    k = capi.lib().atsc_clean_data(a.ctypes.data_as(C.POINTER(C.c_double)), len(a),
                                   out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:k].copy()


def bro_prefix(n_frames):
    buf = (C.c_uint8 * 32)()
    k = capi.lib().atsc_bro_prefix(n_frames, buf)
    return bytes(buf[:k])


def bro_open(bro):
    b = np.frombuffer(bytes(bro), dtype=np.uint8)
    off = C.c_uint64()
    nf = C.c_uint64()
    rc = capi.lib().atsc_bro_open(b.ctypes.data_as(C.POINTER(C.c_uint8)), len(b), C.byref(off),
                                  C.byref(nf))
    capi.check(rc)
    return off.value, nf.value
