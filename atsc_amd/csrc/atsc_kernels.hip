// atsc_kernels.hip -- hand-written gfx950 kernels for the ATSC per-frame compressor path.
//
// One workgroup of W wavefronts (64 lanes each) owns one frame.  The frame's f64 samples,
// the twiddle table of its transform length and two complex-f32 work buffers live in LDS;
// each lane keeps its share of the (Gibbs padded) samples in registers for the error ladders.
//
// What replaces what (reference paths relative to the instaclustr/atsc repository root):
//   stats_phase        DataStats::new / bitdepth / split_n      atsc/src/optimizer/utils.rs:39-160
//   fft_forward        rustfft forward plan + process            atsc/src/compressor/fft.rs:315-323
//   sort + ladder      fft_trim + mirrored iFFT + round + MAPE   atsc/src/compressor/fft.rs:231-257,334-353
//   poly ladder        Polynomial::compress_bounded              atsc/src/compressor/polynomial.rs:209-305,342-373
//   rle phase          IndexRLE::new + Encode                    atsc/src/compressor/rle.rs:40-67,142-189
//   selector           CompressorFrame::compress_best            atsc/src/frame/mod.rs:71-149
//   emitters           bincode Encode impls                      fft.rs:119-130, polynomial.rs:54-87,
//                                                                constant.rs:37-64, rle.rs:40-67, noop.rs:23-27
//
// The FFT ladder does not run 23 inverse FFTs.  The inverse transform of a K-sparse Hermitian
// spectrum is linear, so each ladder trip only adds the newly admitted bins to a per-sample
// running sum (2 FMAs per bin and sample, twiddles from the LDS table); the result is the
// same quantity the reference's inverse FFT produces, up to complex-f32 rounding.
//
// No MFMA: there is no dense contraction on this path.  Built with -ffp-contract=off so the
// f64 spline arithmetic is evaluated exactly as written (Rust never fuses a*b+c).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <string.h>

#include "atsc_device.h"

namespace atsc {

// --------------------------------------------------------------------------------------------
// forward transform, complex f32, unnormalised, e^{-i...}  (rustfft plan_fft_forward)
// Stockham autosort of length M = P.M, radices 4/2/3; tw[t] = (cos, sin)(2 pi t / L) is the
// length-L table, so the stage twiddle w_M^e is tw[e * P.sc].  Returns the result buffer.
// --------------------------------------------------------------------------------------------
// (cmul_conj_tw and fft_forward<W>: atsc_device.h -- the decoder's inverse transform runs through them too)
// Real input, even L = 2M: the packed signal z[j] = g[2j] + i g[2j+1] went through one complex FFT
// of length M (Z).  Bins 0..M of the length-L transform of g:
//   X[k] = (Z[k] + conj Z[M-k]) / 2  +  w_L^k * (-i/2) (Z[k] - conj Z[M-k])
template <int W>
DEVI void fft_untangle(const DevPlan &P, const float2 *Z, float2 *out, const float2 *tw)
{
    constexpr int T = 64 * W;
    const uint32_t M = P.M;
    for (uint32_t k = tid_now<W>(); k <= M; k += T) {
        const float2 zk = Z[k == M ? 0 : k];
        const float2 zm = Z[k == 0 ? 0 : M - k];
        const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);   // Z[k] + conj Z[M-k]
        const float2 b = make_float2(zk.x - zm.x, zk.y + zm.y);   // Z[k] - conj Z[M-k]
        const float2 t = cmul_conj_tw(make_float2(b.y, -b.x), tw[k]);  // w^k * (-i b)
        out[k] = make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
    }
    __syncthreads();
}

// The same transform with the length and the radix sequence known at compile time (frame lengths
// that get their own instantiation of k_compress): identical arithmetic, operation for operation, but
// strides, trip counts and the t / stride split are constants.
// pairs of radix-3 stages fused (fft_stage33_fixed): multi-wavefront frames, whose stages end in a real barrier
#ifndef ATSC_FUSE33_MIN_W
#define ATSC_FUSE33_MIN_W 2
#endif
template <int W>
constexpr bool FUSE33 = W >= ATSC_FUSE33_MIN_W;
template <int W, int M, int SC, int R, int ST>
DEVI void fft_stage_fixed(const float2 *X, float2 *Y, const float2 *tw)
{
    constexpr int T = 64 * W;
    constexpr uint32_t nb = M / R, m = (M / ST) / R, sm = ST * m;
    for (uint32_t t = tid_now<W>(); t < nb; t += T) {
        const uint32_t p = t / ST, q = t - p * ST;
        const uint32_t ib = t, ob = q + ST * (R * p), tb = p * ST * SC;
        if (R == 4) {
            const float2 a0 = X[ib], a1 = X[ib + sm], a2 = X[ib + 2 * sm], a3 = X[ib + 3 * sm];
            const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y);
            const float2 t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
            const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
            const float2 d = make_float2(a1.x - a3.x, a1.y - a3.y);
            const float2 t3 = make_float2(d.y, -d.x);
            Y[ob] = make_float2(t0.x + t2.x, t0.y + t2.y);
            Y[ob + ST] = cmul_conj_tw(make_float2(t1.x + t3.x, t1.y + t3.y), tw[tb]);
            Y[ob + 2 * ST] = cmul_conj_tw(make_float2(t0.x - t2.x, t0.y - t2.y), tw[2 * tb]);
            Y[ob + 3 * ST] = cmul_conj_tw(make_float2(t1.x - t3.x, t1.y - t3.y), tw[3 * tb]);
        } else if (R == 2) {
            const float2 a0 = X[ib], a1 = X[ib + sm];
            Y[ob] = make_float2(a0.x + a1.x, a0.y + a1.y);
            Y[ob + ST] = cmul_conj_tw(make_float2(a0.x - a1.x, a0.y - a1.y), tw[tb]);
        } else {
            const float2 a0 = X[ib], a1 = X[ib + sm], a2 = X[ib + 2 * sm];
            const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
            const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
            const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
            const float h = 0.8660254037844386f;
            const float2 t3 = make_float2(h * d.y, -h * d.x);
            Y[ob] = make_float2(a0.x + t1.x, a0.y + t1.y);
            Y[ob + ST] = cmul_conj_tw(make_float2(t2.x + t3.x, t2.y + t3.y), tw[tb]);
            Y[ob + 2 * ST] = cmul_conj_tw(make_float2(t2.x - t3.x, t2.y - t3.y), tw[2 * tb]);
        }
    }
    __syncthreads();
}
// Two radix-3 stages in one pass: the three butterflies p = p' + m' j' (j' = 0 .. 2) of the stage with stride ST write
// exactly the nine points that the three butterflies (p', q + ST k) of the next stage (stride 3 ST) read, so a thread
// that runs the former can run the latter on their results in registers -- the same operations in the same order, the
// results bit for bit those of the two passes -- and the nine points make one trip through LDS and one barrier instead
// of two.  A 4096-sample frame's transform (M = 3^7) is seven radix-3 stages otherwise: seven barriers of the one
// workgroup its CU holds.
template <int W, int M, int SC, int ST>
DEVI void fft_stage33_fixed(const float2 *X, float2 *Y, const float2 *tw)
{
    constexpr int T = 64 * W;
    constexpr uint32_t nb = M / 9, m1 = (M / ST) / 3, m2 = m1 / 3, sm = ST * m1, ST2 = 3 * ST;
    const float h = 0.8660254037844386f;
    auto bfly = [&](const float2 a0, const float2 a1, const float2 a2, const uint32_t tb, float2 &y0, float2 &y1, float2 &y2) {
        const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
        const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
        const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
        const float2 t3 = make_float2(h * d.y, -h * d.x);
        y0 = make_float2(a0.x + t1.x, a0.y + t1.y);
        y1 = cmul_conj_tw(make_float2(t2.x + t3.x, t2.y + t3.y), tw[tb]);
        y2 = cmul_conj_tw(make_float2(t2.x - t3.x, t2.y - t3.y), tw[2 * tb]);
    };
    for (uint32_t t = tid_now<W>(); t < nb; t += T) {
        const uint32_t p2 = t / ST, q = t - p2 * ST;  // p' < m2
        float2 y[3][3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const uint32_t p = p2 + m2 * jj;
            const uint32_t ib = q + ST * p;
            bfly(X[ib], X[ib + sm], X[ib + 2 * sm], p * ST * SC, y[jj][0], y[jj][1], y[jj][2]);
        }
        const uint32_t tb2 = p2 * ST2 * SC;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t ob = (q + ST * k) + ST2 * (3 * p2);
            float2 z0, z1, z2;
            bfly(y[0][k], y[1][k], y[2][k], tb2, z0, z1, z2);
            Y[ob] = z0;
            Y[ob + ST2] = z1;
            Y[ob + 2 * ST2] = z2;
        }
    }
    __syncthreads();
}
// the stage list of the host (build_plan_entry: 4s, then 2s, then 3s) unrolled at compile time,
// e.g. n = 256: L = 288, M = 144 = 4 * 4 * 3 * 3; pairs of radix-3 stages run fused (fft_stage33_fixed)
template <int W, int M, int SC, int REM = M, int ST = 1>
DEVI float2 *fft_forward_fixed(float2 *X, float2 *Y, const float2 *tw)
{
    if constexpr (REM == 1) {
        return X;
    } else if constexpr (FUSE33<W> && REM % 2 != 0 && REM % 9 == 0) {
        fft_stage33_fixed<W, M, SC, ST>(X, Y, tw);
        return fft_forward_fixed<W, M, SC, REM / 9, ST * 9>(Y, X, tw);
    } else {
        constexpr int R = (REM % 4 == 0) ? 4 : (REM % 2 == 0) ? 2 : 3;
        fft_stage_fixed<W, M, SC, R, ST>(X, Y, tw);
        return fft_forward_fixed<W, M, SC, REM / R, ST * R>(Y, X, tw);
    }
}
template <int W, int M>
DEVI void fft_untangle_fixed(const float2 *Z, float2 *out, const float2 *tw)
{
    constexpr int T = 64 * W;
#pragma unroll
    for (uint32_t k0 = 0; k0 <= M; k0 += T) {
        const uint32_t k = k0 + tid_now<W>();
        if (k <= M) {
            const float2 zk = Z[k == M ? 0 : k];
            const float2 zm = Z[k == 0 ? 0 : M - k];
            const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);
            const float2 b = make_float2(zk.x - zm.x, zk.y + zm.y);
            const float2 t = cmul_conj_tw(make_float2(b.y, -b.x), tw[k]);
            out[k] = make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
        }
    }
    __syncthreads();
}

constexpr uint32_t cx_next_size(uint32_t n)  // utils/mod.rs:32-49 at compile time
{
    for (uint32_t v = n + 1;; ++v) {
        uint32_t r = v;
        while (r % 2 == 0) r /= 2;
        while (r % 3 == 0) r /= 3;
        if (r == 1) return v;
    }
}

// O(n^2) transform for n < 128 (any n, primes included).  f64 accumulation, f32 result.
template <int W>
DEVI void dft_direct(const DevPlan &P, const double *xs, float2 *out, const float2 *tw)
{
    constexpr int T = 64 * W;
    const uint32_t n = P.L;
    for (uint32_t k = tid_now<W>(); k < P.bins; k += T) {
        double re = 0.0, im = 0.0;
        uint32_t idx = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const double g = (double)(float)xs[j];
            const float2 w = tw[idx];
            re += g * (double)w.x;
            im -= g * (double)w.y;
            idx += k;
            if (idx >= n) idx -= n;
        }
        out[k] = make_float2((float)re, (float)im);
    }
    __syncthreads();
}

// Sort of RLE run records rec = (start << 16 | end) by (bits of the run value xs[end], start):
// the BTreeMap<u64, Vec<usize>> order of rle.rs:146-182.  Ascending-only bitonic network.
template <int W>
DEVI void block_sort_runs(uint32_t *rec, const double *xs, uint32_t count, uint32_t P2)
{
    constexpr int T = 64 * W;
    const uint32_t tid = tid_now<W>();
    const uint32_t npairs = P2 >> 1;
    auto ce = [&](uint32_t i, uint32_t l) {
        if (l < count) {
            const uint32_t ra = rec[i], rb = rec[l];
            const uint64_t ka = (uint64_t)__double_as_longlong(xs[ra & 0xffffu]);
            const uint64_t kb = (uint64_t)__double_as_longlong(xs[rb & 0xffffu]);
            if (ka > kb || (ka == kb && ra > rb)) { rec[i] = rb; rec[l] = ra; }
        }
    };
    uint32_t lk = 1;
    for (uint32_t k = 2; k <= P2; k <<= 1, ++lk) {
        const uint32_t half = k >> 1;
        for (uint32_t t = tid; t < npairs; t += T) {
            const uint32_t i = ((t >> (lk - 1)) << lk) | (t & (half - 1));
            ce(i, i ^ (k - 1));
        }
        __syncthreads();
        for (uint32_t j = half >> 1; j >= 1; j >>= 1) {
            for (uint32_t t = tid; t < npairs; t += T) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                ce(i, i | j);
            }
            __syncthreads();
        }
    }
}

// Sorting a few hundred DISTINCT keys by counting: a key's place is the number of smaller keys.  The bitonic
// networks above are log2(P2) (log2(P2) + 1) / 2 rounds, each ending in the whole workgroup's barrier -- 36 rounds for
// the 176 keys of a 4096-sample frame's first admission order, with 88 of 1024 threads at work -- where this is four
// barriers and count^2 / T compares per thread.  p threads share a key (a power of two, at most 16): each counts over
// its stride of the keys and adds its share to the key's rank in LDS.  tmp: count u64 + count u32 of scratch.
template <int W>
DEVI void block_rank_sort(uint64_t *keys, uint64_t *tmp, uint32_t count)
{
    constexpr uint32_t T = 64 * W;
    const uint32_t tid = tid_now<W>();
    uint32_t *rank = (uint32_t *)(tmp + count);
    uint32_t p = 1;
    while (p < 16 && 2 * p * count <= T) p <<= 1;
    for (uint32_t i = tid; i < count; i += T) rank[i] = 0;
    __syncthreads();
    for (uint32_t w = tid; w < p * count; w += T) {
        const uint32_t i = w / p, q = w - i * p;
        const uint64_t mine = keys[i];
        uint32_t r = 0;
        for (uint32_t j = q; j < count; j += p) r += keys[j] < mine ? 1u : 0u;
        if (p == 1) rank[i] = r;
        else if (r) atomicAdd(&rank[i], r);
    }
    __syncthreads();
    for (uint32_t i = tid; i < count; i += T) tmp[rank[i]] = keys[i];
    __syncthreads();
    for (uint32_t i = tid; i < count; i += T) keys[i] = tmp[i];
    __syncthreads();
}
// The same for RLE run records (block_sort_runs' order: value bits, then the record, i.e. the start).  tmp: 2 count u32.
template <int W>
DEVI void block_rank_sort_runs(uint32_t *rec, const double *xs, uint32_t count, uint32_t *tmp)
{
    constexpr uint32_t T = 64 * W;
    const uint32_t tid = tid_now<W>();
    uint32_t *rank = tmp + count;
    uint32_t p = 1;
    while (p < 16 && 2 * p * count <= T) p <<= 1;
    for (uint32_t i = tid; i < count; i += T) rank[i] = 0;
    __syncthreads();
    for (uint32_t w = tid; w < p * count; w += T) {
        const uint32_t i = w / p, q = w - i * p;
        const uint32_t ra = rec[i];
        const uint64_t ka = (uint64_t)__double_as_longlong(xs[ra & 0xffffu]);
        uint32_t r = 0;
        for (uint32_t j = q; j < count; j += p) {
            const uint32_t rb = rec[j];
            const uint64_t kb = (uint64_t)__double_as_longlong(xs[rb & 0xffffu]);
            r += (kb < ka || (kb == ka && rb < ra)) ? 1u : 0u;
        }
        if (p == 1) rank[i] = r;
        else if (r) atomicAdd(&rank[i], r);
    }
    __syncthreads();
    for (uint32_t i = tid; i < count; i += T) tmp[rank[i]] = rec[i];
    __syncthreads();
    for (uint32_t i = tid; i < count; i += T) rec[i] = tmp[i];
    __syncthreads();
}

// --------------------------------------------------------------------------------------------
// the frame kernel
// --------------------------------------------------------------------------------------------
#ifdef ATSC_STAMPS
// Dev build only (-DATSC_STAMPS, tools/stamp_probe.py): shader-clock cycles per phase of the fixed-length
// one-wavefront kernel.  A frame sums its phases in LDS (128 static bytes: 22 instead of 23 frames per CU) and
// adds them to one of 64 global rows when it ends.
__device__ unsigned long long g_phase_cyc[64 * 16];
__device__ unsigned long long g_frame_span[4 * 65536 * 3];  // wall clock (100 MHz) at a frame's start and end, by launch slot; HW_ID | XCC_ID << 32
#define PH(i) do { if (FN != 0) { const long long now_ = clock64(); \
    if (tid == 0) atomicAdd(&ph_acc[i], (unsigned long long)(now_ - ph_t)); ph_t = clock64(); } } while (0)
#else
#define PH(i) do {} while (0)
#endif
// one-wavefront frames of the 256-sample class: ask for 6 wavefronts per SIMD (<= 80 VGPRs).
// FN != 0: every frame of the launch has FN samples (FN >= 128, even transform length) and the frame
// geometry is folded at compile time; FN == 0 reads it from the per-length table.
DEVI void frame_prio(uint32_t trips)  // (s_setprio takes an immediate)
{
#ifndef ATSC_FRAME_PRIO  // off: measured 1-2 % slower for the one-workgroup-per-frame launch (below)
    return;
#endif
    if (trips == 3) __builtin_amdgcn_s_setprio(1);
    else if (trips == 7) __builtin_amdgcn_s_setprio(2);
    else if (trips == 12) __builtin_amdgcn_s_setprio(3);
}

template <int W, int SPL, bool IDW, int FN, bool LEAN_ = (FN != 0)>
__device__ __forceinline__ void compress_frame(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames,
    const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, const KParams &prm, uint8_t *__restrict__ slots,
    DevResult *__restrict__ res, atsc_frame_diag *__restrict__ diag, const UniArgs &uni, const uint32_t bid)
{
    constexpr int T = 64 * W;
    constexpr bool FIX_ = FN != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = tid_now<W>();  // (opaque: see tid_now)
    // (-DATSC_FRAME_PRIO: issue priority follows the frame's age in ladder trips, frame_prio above.  It exists for
    // resident workgroups -- k_compress_resident: their wavefronts were all started together, the hardware's oldest-first
    // arbitration ranks them once and for all, and the last of a SIMD's five took 100 us and more over a frame that
    // takes 15 -- and costs the per-frame grid 1-2 %: 97.3 vs 99.2 Gsamples/s, configs[3] 58.3 vs 58.8.)
#ifdef ATSC_FRAME_PRIO
    if (W == 1) __builtin_amdgcn_s_setprio(0);
#endif
    uint32_t fid;
    DevFrame fr;
    const long long t_start = prm.cost ? clock64() : 0;
#ifdef ATSC_STAMPS
    __shared__ unsigned long long ph_acc[16];
    if (tid < 16) ph_acc[tid] = 0;
    __syncthreads();
    long long ph_t = clock64();
    const uint32_t span0 = 3u * 65536u * (((uint32_t)prm.debug_stop >> 24) & 3u);
    if (FN != 0 && tid == 0 && bid < 65536) {
        g_frame_span[span0 + 3 * bid] = wall_clock64();
        g_frame_span[span0 + 3 * bid + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                                           ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);
    }
#endif
    if (FIX_ || uni.enabled) {  // (the fixed-length kernels are only launched on uniform classes)
        // Index order starts the frames series block by series block: when the last block of a batch is a busy one the
        // launch drains on a few CUs.  Without a cost order the frames are dealt with a fixed stride instead, so that
        // every stretch of the launch is a mix of the batch's blocks (results are positional: nothing else changes).
        // (the same for every lane; said so, because a pointer that reached this function through memory makes
        // the load a per-lane one and everything derived from it per-lane arithmetic)
        const uint32_t r = uni.adaptive ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(ids[bid] - uni.fid0))
                           : uni.spread ? (uint32_t)(((uint64_t)bid * uni.spread) % uni.count) : bid;
        fid = uni.fid0 + r;
        fr.sample_off = uni.sample_off0 + (uint64_t)r * uni.n;
        fr.slot_off = uni.slot_off0 + (uint64_t)r * uni.slot_stride;
        fr.n = uni.n;
        fr.plan = uni.plan;
    } else {
        fid = ids[bid];
        fr = frames[fid];
    }
    // (through the constant address space: the plan table is read-only for the life of the kernel, and a pointer that
    // reached this function through memory -- k_compress_resident -- carries no such promise by itself; without it
    // every table read is a vector load that has to be repeated after each store)
    typedef const __attribute__((address_space(4))) DevPlan PlanConst;
    PlanConst &P = *(PlanConst *)(plans + fr.plan);
#define P_GENERIC (*(const DevPlan *)&P)  // for the helpers that take the plan by (generic) reference
    constexpr bool FIX = FN != 0;
    // The fixed-length instantiations are the production path: no diagnostics record, no phase stops, no
    // sample-level trial (launch_class routes those calls to the table-driven instantiation).  Their
    // bookkeeping would otherwise sit in scalar registers for the whole kernel.  They also only serve the auto
    // selector with an error bound >= 0 (`atsc -e`, BASELINE's configurations): mode, bounded and prune fold.
    constexpr bool LEAN = LEAN_;  // (every fixed-length instantiation is lean; the table-driven ones come in both forms)
    static_assert(LEAN || !FIX, "fixed-length instantiations are lean");
    constexpr uint32_t cL = FIX ? cx_next_size(FIX ? FN : 1) : 0, cmf = (3 >= FN / 100) ? 3 : FN / 100;
    static_assert(!FIX || (FN >= 128 && cL <= 64 * W * SPL), "fixed-length instantiation");
    constexpr bool chalf = cL % 2 == 0;              // even L: packed real transform of length L / 2
    constexpr uint32_t cM = chalf ? cL / 2 : cL;     // points through the Stockham stages
    const uint32_t n = FIX ? FN : P.n, L = FIX ? cL : P.L, pre = FIX ? (cL - FN) / 2 : P.pre;
    const uint32_t bins = FIX ? cL / 2 + 1 : P.bins;
    const uint32_t mf = FIX ? cmf : P.mf;
    constexpr uint32_t cdk1 = cmf / 2 > 1 ? cmf / 2 : 1, cdk2 = cmf / 10 > 1 ? cmf / 10 : 1;
    constexpr uint32_t ckcap = (cL / 2 + 1 < cmf + 17 * cdk1 + 5 * cdk2) ? cL / 2 + 1 : cmf + 17 * cdk1 + 5 * cdk2;
    const uint32_t dk1 = FIX ? cdk1 : P.dk1, dk2 = FIX ? cdk2 : P.dk2;
    // LDS carve-up: the host's layout function evaluated at compile time for a fixed length
    constexpr EncLds lay = enc_lds(FIX ? FN : 1, FIX ? cL : 1, FIX ? cM : 1, false, ckcap, W == 1);
    const uint32_t o_xs = FIX ? lay.o_xs : P.o_xs, o_tw = FIX ? lay.o_tw : P.o_tw, o_ab = FIX ? lay.o_ab : P.o_ab;
    const uint32_t ab_half = FIX ? lay.ab_half : P.ab_half, ab_bytes = FIX ? lay.ab_bytes : P.ab_bytes;
    const uint32_t o_sel = FIX ? lay.o_sel : P.o_sel, o_aux = FIX ? lay.o_aux : P.o_aux, o_red = FIX ? lay.o_red : P.o_red;
    const uint32_t o_hist = FIX ? lay.o_hist : P.o_hist;
    const uint32_t kcap = FIX ? ckcap : P.kcap;

    double *xs = (double *)(smem + o_xs);
    float2 *tw = (float2 *)(smem + o_tw);
    unsigned char *AB = smem + o_ab;
    float2 *A = (float2 *)AB;
    float2 *B = (float2 *)(AB + ab_half);
    Sel *sel = (Sel *)(smem + o_sel);
    uint32_t *aux = (uint32_t *)(smem + o_aux);
    double *red = (double *)(smem + o_red);
    uint32_t *wsum = (W == 1) ? (uint32_t *)red : (uint32_t *)(red + 32);  // W == 1: `red` is 16 bytes
    int parity = 0;

    uint8_t *out = slots + fr.slot_off;
    int mode = LEAN ? (int)ATSC_AUTO : prm.mode;
    const bool bounded = LEAN ? true : (prm.bounded != 0);

    constexpr bool REGSTAT = FIX && W == 1 && (FN == 256 || FN == 512);  // (128: the extra state spills)
    bool reg_stats = false;
    double rs_mn = 0.0, rs_mx = 0.0;
    uint32_t rs_frac = 0, rs_pk = 0;
    // ---- load samples (the twiddles follow when the FFT candidate starts: until then and after
    // its ladder their region hosts `aux` and the RLE group table) ----------------------------
    {
        const double *src = samples + fr.sample_off;
        if constexpr (REGSTAT) {
            if ((fr.sample_off & 1ull) == 0) {
                // One-wavefront frames of a fixed even length: the lane's pairs stay in registers for the statistics
                // and the RLE bound (run starts and their index bytes), which so need neither the LDS round trips of
                // reading the samples back (4 + 8 per lane at n = 256) nor the barrier in front of them.  A sample's
                // left neighbour is the pair's other half, or the lane before's second half (DPP wave_shr:1), or --
                // lane 0 -- the last lane's second half of the chunk before (v_readlane).
                constexpr int NP = FN / 128;  // pairs per lane
                const double2 *src2 = (const double2 *)src;
                double2 *xs2 = (double2 *)xs;
                double2 v[NP];
#pragma unroll
                for (int m = 0; m < NP; ++m) v[m] = src2[tid + 64 * m];
#pragma unroll
                for (int m = 0; m < NP; ++m) xs2[tid + 64 * m] = v[m];
                const double x0 = lane_f64(v[0].x, 0);
                rs_mn = x0;
                rs_mx = x0;
#pragma unroll
                for (int m = 0; m < NP; ++m) {
                    const double a = v[m].x, b = v[m].y;
                    rs_frac |= (frac_nonzero(a) || frac_nonzero(b)) ? 1u : 0u;
                    if (a > rs_mx) rs_mx = a;
                    if (a < rs_mn) rs_mn = a;
                    if (b > rs_mx) rs_mx = b;
                    if (b < rs_mn) rs_mn = b;
                    double left = wave_shr1_f64(b);                     // lane - 1's second half
                    if (m > 0) {
                        const double wrap = lane_f64(v[m - 1].y, 63);   // (uniform)
                        left = (tid == 0) ? wrap : left;
                    }
                    const uint32_t jx = 2u * (tid + 64u * (uint32_t)m);
                    if (jx == 0 || a != left) rs_pk += (vlen(jx) << 13) | 1u;
                    if (b != a) rs_pk += (vlen(jx + 1) << 13) | 1u;
                }
                reg_stats = true;
            }
        }
        if (reg_stats) {
        } else if (((fr.sample_off | n) & 1ull) == 0) {  // 16 B per lane when the frame is 16-B aligned
            const double2 *src2 = (const double2 *)src;
            double2 *xs2 = (double2 *)xs;
            for_strided<FN / 2, T, (SPL + 1) / 2>(tid, n >> 1, [&](uint32_t j) { xs2[j] = src2[j]; });
        } else {
            for (uint32_t j = tid; j < n; j += T) xs[j] = src[j];
        }
    }
    __syncthreads();

    PH(0);
    // ---- stats (optimizer/utils.rs:39-89): min / max are the FIRST occurrence of the extreme
    // value (strict compares), so +0.0 / -0.0 resolve as the sequential scan does ---------------
    double smin, smax;
    uint32_t bitdepth;
    uint32_t stat_pk = 0;       // multi-wavefront frames: (index varint bytes << 13 | run count), summed with the statistics
    bool stat_pk_done = false;
    {
        double mn, mx;
        uint32_t fr_any = 0;
        if (reg_stats) {
            mn = rs_mn; mx = rs_mx; fr_any = rs_frac;
        } else {
            const double x0 = xs[0];
            mn = x0; mx = x0;
            for_strided<FN, T, SPL>(tid, n, [&](uint32_t j) {
                const double v = xs[j];
                fr_any |= frac_nonzero(v) ? 1u : 0u;
                if (v > mx) mx = v;
                if (v < mn) mn = v;
                if constexpr (W > 1) {  // the RLE bound's run starts and index bytes ride on the same walk
                    const double b = xs[j ? j - 1 : 0];
                    if (j == 0 || v != b) stat_pk += (vlen(j) << 13) | 1u;
                }
            });
        }
        if constexpr (W > 1) {
            const BlockStats bs = block_stats4<W>(mn, mx, fr_any, stat_pk, red, wsum);
            mn = bs.mn; mx = bs.mx; fr_any = bs.frac; stat_pk = bs.pk;
            stat_pk_done = true;
        } else {
            mn = block_minmax_f64<W, true>(mn, red, parity);
            mx = block_minmax_f64<W, false>(mx, red, parity);
            fr_any = block_or_u32<W>(fr_any, red, parity);
        }
        // Samples that compare equal to the extreme value have the same bits, except for zeros
        // (+0.0 == -0.0): only then the scan's "first occurrence" has to be looked up.  A NaN in
        // data[0] poisons every compare and the scan keeps data[0] (mn, mx are x0 then).
        smin = mn;
        smax = mx;
        if (mn == 0.0 || mx == 0.0) {
            uint32_t mni = 0xFFFFFFFFu, mxi = 0xFFFFFFFFu;
            for_strided<FN, T, SPL>(tid, n, [&](uint32_t j) {
                const double v = xs[j];
                if (v == mn) mni = min(mni, j);
                if (v == mx) mxi = min(mxi, j);
            });
            mni = block_min_u32<W>(mni, red, parity);
            mxi = block_min_u32<W>(mxi, red, parity);
            if (mni < n) smin = xs[mni];
            if (mxi < n) smax = xs[mxi];
        }
        int64_t maxi, mini;
        bool fz;
        split_n(smax, maxi, fz);
        split_n(smin, mini, fz);
        bitdepth = fr_any ? 0u : bitdepth_of(maxi, mini);
    }

    PH(1);
    if (!LEAN && prm.debug_stop == 1) return;
    atsc_frame_diag dg;
    dg.fft_size = dg.poly_size = dg.rle_size = 0xFFFFFFFFu;
    dg.fft_trips = dg.fft_k = dg.poly_trips = dg.poly_step = 0;
    dg.poly_points = 0;
    dg.fft_err = dg.poly_err = 0.0;

    // ---- Constant: frame/mod.rs:82-88 (auto shortcut) or forced (constant.rs:135-139) ------
    if (mode == ATSC_CONSTANT || (mode == ATSC_AUTO && (LEAN || !prm.trial) && smin == smax)) {
        if (tid == 0) {
            out[0] = 30;
            out[1] = (uint8_t)bitdepth;
            const uint32_t vb = put_value(out + 2, bitdepth, smin);
            res[fid].err = 0.0;
            res[fid].len = 2 + vb;
            res[fid].chosen = ATSC_CONSTANT;
            if (!LEAN && diag) diag[fid] = dg;
            if (prm.cost) prm.cost[fid] = (uint32_t)min((unsigned long long)(clock64() - t_start) >> 6, 0xFFFFFFFFull);
        }
        return;
    }

    // ---- Noop: noop.rs:37-43,72-77 ---------------------------------------------------------
    if (mode == ATSC_NOOP) {
        for (uint32_t j = tid; j < n; j += T) aux[j] = vlen(zigzag(sat_i64(round(xs[j]))));
        __syncthreads();
        const uint32_t tot = block_excl_scan<W>(aux, n, wsum);
        const uint32_t hdr = 1 + vlen(n);
        for (uint32_t j = tid; j < n; j += T)
            put_varint(out + hdr + aux[j], zigzag(sat_i64(round(xs[j]))));
        if (tid == 0) {
            out[0] = 250;
            put_varint(out + 1, n);
            res[fid].err = 0.0;
            res[fid].len = hdr + tot;
            res[fid].chosen = ATSC_NOOP;
            if (!LEAN && diag) diag[fid] = dg;
        }
        return;
    }

    // frame/mod.rs:89-111: frames of at least COMPRESSION_SPEED[level] samples take the codec the
    // trial on their first COMPRESSION_SPEED[level] samples chose, whatever error it then reaches
    if (!LEAN && mode == ATSC_AUTO && prm.trial_res != nullptr && n >= prm.trial_min_n)
        mode = (int)prm.trial_res[fid].chosen;

    if (!LEAN && prm.debug_stop == 2) return;
    // per-lane share of the padded signal g (fft.rs:184-204): lane owns j = tid + m*T.  Filled below,
    // once it is known that a ladder will run at all.
    double g[SPL], inv[SPL];

    if (!LEAN && prm.debug_stop == 3) return;
    // =========================================================================================
    // Candidates.  frame/mod.rs:113-147 keeps the smallest payload among the candidates whose error
    // passes `err <= max_error`, the first of [FFT, Polynomial, RLE] on ties.  Every ladder's payload
    // only grows from trip to trip, so a ladder is abandoned as soon as its next payload could no
    // longer beat a candidate that already passes -- the winner and its bytes are unchanged.
    // =========================================================================================
    const bool run_fft = (mode == ATSC_AUTO || mode == ATSC_FFT);
    const bool run_poly = (mode == ATSC_AUTO || mode == ATSC_POLYNOMIAL || mode == ATSC_IDW);
    const bool run_rle = (mode == ATSC_AUTO || mode == ATSC_RLE);
    // polynomial.rs:29-34,202-207: Idw is the same codec with another interpolation; it is only
    // reachable as a forced codec and is compiled as its own instantiation (IDW)
    const bool idw = IDW && (mode == ATSC_IDW);
    const double me = prm.max_err;
    // RLE reports error 0.0: it passes, and bounds the others, exactly when 0.0 <= max_error
    const bool prune = LEAN ? true : (mode == ATSC_AUTO) && (0.0 <= me);
    uint32_t best_size = 0xFFFFFFFFu;
    int best_owner = 3;  // 0 FFT, 1 Polynomial, 2 RLE: the tie order of frame/mod.rs:77
    auto can_win = [&](uint32_t size_lb, int owner) {
        return size_lb < best_size || (size_lb == best_size && owner < best_owner);
    };
    auto offer = [&](uint32_t size, int owner) {
        if (can_win(size, owner)) { best_size = size; best_owner = owner; }
    };
    uint32_t fft_k = 0, fft_size = 0xFFFFFFFFu, fft_trips = 0;
    double fft_err = 0.0;
    bool fft_done = false;
    const float mxf = (float)smax, mnf = (float)smin;
    uint32_t poly_step = 1, poly_K = 0, poly_size = 0xFFFFFFFFu, poly_trips = 0;
    double poly_err = 0.0;
    bool poly_done = false;
    // exact Polynomial payload size for (step, K): polynomial.rs:54-87
    uint32_t psz_step = 0, psz_K = 0xFFFFFFFFu, psz = 0;  // last (step, K) sized: asked for up to three times
    auto poly_payload_size = [&](uint32_t step, uint32_t K) -> uint32_t {
        if (step == psz_step && K == psz_K) return psz;
        uint32_t vb = 0;
        if (bitdepth == 0 || bitdepth == 3) {
            vb = K * (bitdepth == 0 ? 8u : 1u);
        } else {
            for (uint32_t k = tid; k < K; k += T) {
                const uint32_t t = (k == K - 1) ? (n - 1) : k * step;
                vb += value_bytes(bitdepth, xs[t]);
            }
            vb = block_sum_u32<W>(vb, red, parity);
        }
        psz_step = step;
        psz_K = K;
        psz = 1 + 1 + vlen(K) + vb + 8 + 8 + 1;
        return psz;
    };

    // ---- RLE (rle.rs:142-189): cheap bound first; exact right away when there are few runs ----
    uint32_t rle_size = 0xFFFFFFFFu, rle_R = 0, rle_D = 0, rle_ib = 0, rle_lb = 0xFFFFFFFFu;
    bool rle_sorted = false, rle_pending = false;
    bool rle_early = false;  // the early run arrays in AB are still intact
    uint32_t *rrec_std = (uint32_t *)AB;        // 4n: run records (start << 16 | end), n <= 4096
    uint32_t *rhp_std = rrec_std + n;           // 4n+4: group head positions hp[0..D]
    uint32_t *rph_std = (uint32_t *)tw;         // 4n  group header bytes / their prefix (aux = tw + 4n)
    auto run_key = [&](uint32_t rec) { return (uint64_t)__double_as_longlong(xs[rec & 0xffffu]); };
    // Sorts the runs by (value bits, start) = BTreeMap order (rle.rs:146,158-169,180-182) and sizes
    // the groups.  Leaves: rrec sorted, aux[i] = heads before i, rhp[g] = index of the first run of
    // group g (rhp[D] = R), rph[g] = header bytes of group g.
    auto rle_sort_and_group = [&](uint32_t *rrec, uint32_t *rhp, uint32_t *rph) {
        __syncthreads();
        for (uint32_t j = tid; j < n; j += T)
            aux[j] = (j + 1 >= n || xs[j + 1] != xs[j]) ? 1u : 0u;  // run ends (rle.rs:154)
        __syncthreads();
        const uint32_t R = block_excl_scan<W>(aux, n, wsum);
        uint32_t *ends = rhp;  // scratch until the groups are built
        for (uint32_t j = tid; j < n; j += T)
            if (j + 1 >= n || xs[j + 1] != xs[j]) ends[aux[j]] = j;
        __syncthreads();
        for (uint32_t r = tid; r < R; r += T) {
            const uint32_t st = r ? ends[r - 1] + 1 : 0;
            rrec[r] = (st << 16) | ends[r];
        }
        __syncthreads();
        if (R <= 512 && 2 * R <= n + 2) {
            block_rank_sort_runs<W>(rrec, xs, R, aux);  // (aux is free between the scan above and the flags below)
        } else {
            uint32_t p2 = 1;
            while (p2 < R) p2 <<= 1;
            block_sort_runs<W>(rrec, xs, R, p2);
        }
        for (uint32_t i = tid; i < R; i += T)
            aux[i] = (i == 0 || run_key(rrec[i]) != run_key(rrec[i - 1])) ? 1u : 0u;
        __syncthreads();
        const uint32_t D = block_excl_scan<W>(aux, R, wsum);
        for (uint32_t i = tid; i < R; i += T)
            if (i == 0 || run_key(rrec[i]) != run_key(rrec[i - 1])) rhp[aux[i]] = i;
        if (tid == 0) rhp[D] = R;
        __syncthreads();
        uint32_t hb = 0;
        for (uint32_t gi = tid; gi < D; gi += T) {
            const uint32_t h0 = rhp[gi], h1 = rhp[gi + 1];
            const uint32_t b = value_bytes(bitdepth, xs[rrec[h0] & 0xffffu]) + vlen(h1 - h0);
            rph[gi] = b;
            hb += b;
        }
        __syncthreads();
        hb = block_sum_u32<W>(hb, red, parity);
        rle_R = R;
        rle_D = D;
        rle_size = 2 + vlen(D) + hb + rle_ib;
        rle_sorted = true;
    };
    // One wavefront and at most 16 runs: one run per lane, no LDS scans or sort network.
    //  build: each lane walks its contiguous share of the samples, a DPP scan places the run ends,
    //         lane r ranks run r among all runs by (value bits, start) with R broadcast compares and
    //         drops its record at that rank (AB[0..R) = sorted records, as the general path leaves);
    //  group: sorted run i sits in lane i; a group head differs from its left neighbour (row_shr:1,
    //         R <= 16 keeps everything in DPP row 0); the ballot of the heads gives the group count
    //         and every group's length; header / index bytes are prefix sums across lanes.
    bool rle_small = false;
    struct RleLane { uint32_t rec, cnt, hb, D, hb_total; bool head, live; };
    auto rle_small_build = [&]() {
        uint32_t *srt = (uint32_t *)AB, *ends = srt + 16;
        __syncthreads();
        const uint32_t C = (n + 63) >> 6, j0 = tid * C;
        uint32_t cnt = 0;
        for (uint32_t k = 0; k < C; ++k) {
            const uint32_t j = j0 + k;
            if (j < n && (j + 1 >= n || xs[j + 1] != xs[j])) ++cnt;
        }
        uint32_t pos = wave_incl_scan_u32(cnt) - cnt;
        for (uint32_t k = 0; k < C; ++k) {
            const uint32_t j = j0 + k;
            if (j < n && (j + 1 >= n || xs[j + 1] != xs[j])) ends[pos++] = j;
        }
        __syncthreads();
        const uint32_t R = rle_R;
        uint32_t rec = 0, klo = 0, khi = 0;
        if (tid < R) {
            const uint32_t e = ends[tid], st = tid ? ends[tid - 1] + 1 : 0;
            rec = (st << 16) | e;
            const uint64_t key = (uint64_t)__double_as_longlong(xs[e]);
            klo = (uint32_t)key;
            khi = (uint32_t)(key >> 32);
        }
        uint32_t rank = 0;
        for (uint32_t q = 0; q < R; ++q) {  // runs are in start order: q < tid  <=>  start_q < start_tid
            const uint32_t qlo = (uint32_t)__builtin_amdgcn_readlane((int)klo, (int)q);
            const uint32_t qhi = (uint32_t)__builtin_amdgcn_readlane((int)khi, (int)q);
            const bool less = qhi < khi || (qhi == khi && (qlo < klo || (qlo == klo && q < tid)));
            rank += less ? 1u : 0u;
        }
        if (tid < R) srt[rank] = rec;
        __syncthreads();
    };
    auto rle_small_group = [&]() -> RleLane {
        const uint32_t *srt = (const uint32_t *)AB;
        const uint32_t R = rle_R;
        RleLane g;
        g.live = tid < R;
        g.rec = g.live ? srt[tid] : 0u;
        const uint64_t key = g.live ? (uint64_t)__double_as_longlong(xs[g.rec & 0xffffu]) : 0ull;
        const uint32_t plo = dpp_u32<0x111, 0xf>((uint32_t)key), phi = dpp_u32<0x111, 0xf>((uint32_t)(key >> 32));
        g.head = g.live && (tid == 0 || plo != (uint32_t)key || phi != (uint32_t)(key >> 32));
        const uint64_t hm = __ballot(g.head);
        g.D = (uint32_t)__popcll(hm);
        const uint64_t after = (tid < 63) ? (hm >> (tid + 1)) : 0ull;
        const uint32_t next = after ? tid + 1 + (uint32_t)__builtin_ctzll(after) : R;
        g.cnt = next - tid;
        g.hb = g.head ? value_bytes(bitdepth, xs[g.rec & 0xffffu]) + vlen(g.cnt) : 0u;
        g.hb_total = wave_sum_u32(g.hb);
        return g;
    };
    if (run_rle) {
        // run starts: j == 0 or x[j] != x[j-1]; every start index costs a varint
        uint32_t pk = 0;  // (sum of index varint bytes) << 13 | run count   (n <= 4096)
        if (stat_pk_done) {
            pk = stat_pk;  // summed with the statistics (block_stats4)
        } else {
            if (reg_stats) {
                pk = rs_pk;  // counted from the registers the samples arrived in
            } else {
                for_strided<FN, T, SPL>(tid, n, [&](uint32_t j) {
                    const double a = xs[j], b = xs[j ? j - 1 : 0];  // (both loads unconditional: no branch around an LDS round trip)
                    if (j == 0 || a != b) pk += (vlen(j) << 13) | 1u;
                });
            }
            pk = block_sum_u32<W>(pk, red, parity);
        }
        rle_R = pk & 0x1fffu;
        rle_ib = pk >> 13;
        const uint32_t minval = (bitdepth == 0) ? 8u : 1u;
        rle_lb = 3 + rle_ib + (rle_R >= 2 ? 2u : 1u) * (minval + 1);
        if (W == 1 && rle_R <= 16 && ab_bytes >= 128) {
            rle_small_build();
            const RleLane g = rle_small_group();
            rle_D = g.D;
            rle_size = 2 + vlen(g.D) + g.hb_total + rle_ib;
            rle_small = true;
            rle_early = true;  // AB[0..R) stays valid until a ladder reuses AB
            if (prune) offer(rle_size, 2);
        } else if (mode == ATSC_RLE) {
            rle_sort_and_group(rrec_std, rhp_std, rph_std);
            rle_sorted = true;
        } else if (rle_R <= 64 && 12 * rle_R + 16 <= ab_bytes) {
            // few runs: size it now (AB is still free), so that it can stop the ladders early
            uint32_t *e_rec = (uint32_t *)AB;
            rle_sort_and_group(e_rec, e_rec + rle_R, e_rec + 2 * rle_R + 1);
            rle_sorted = false;  // the ladders below may reuse AB; the emitter sorts again unless rle_early survives
            rle_early = true;
            if (prune) offer(rle_size, 2);
        } else {
            rle_pending = true;
        }
    }

    PH(2);
    if (!LEAN && prm.debug_stop == 10) return;
    // A payload below 19 bytes (FFT with one bin; Polynomial needs at least 23) already beats both
    // ladders: skip loading their operands.
    if (!(prune && best_size < 19)) {
#pragma unroll
        for (int m = 0; m < SPL; ++m) {
            const uint32_t j = tid + m * T;
            if (j < L) {
                int32_t i = (int32_t)j - (int32_t)pre;
                i = i < 0 ? 0 : (i >= (int32_t)n ? (int32_t)n - 1 : i);
                g[m] = xs[i];
                inv[m] = recip_abs(g[m]);
            } else {
                g[m] = 1.0;
                inv[m] = 0.0;
            }
        }
    }

    PH(3);
    if (!LEAN && prm.debug_stop == 11) return;
    // ---- which ladder first: the one whose first payload is the smaller (FFT wins ties) ----
    bool poly_first = false;
    if (prune && run_fft && run_poly && smax != smin) {
        poly_first = poly_payload_size(P.pstep[0], P.pK[0]) < 1 + 1 + 9 * min(mf, bins) + 8;
    }
    // The ladder runs in two goes when it runs first: its first trip before the FFT candidate, the rest after
    // it.  A frame whose polynomial passes at once never transforms anything it does not have to; a frame whose
    // first trip fails would otherwise walk the whole ladder unpruned (nothing passes yet) although the FFT
    // candidate, a few trips later, stores less than the polynomial's next trips could: those are then pruned.
    // Pruning only skips candidates that cannot win, so the selection does not depend on the order.  Only where it
    // pays: 16-wavefront frames (n up to 4096: mf = 40 bins make the FFT's first payload 370 bytes, so the polynomial
    // goes first on most frames, and every trip is a handful of barriers across the one workgroup a CU holds:
    // 35 -> 40 Gsamples/s); smaller frames lose a little to the extra state (256: 110.5 -> 112 us).
    constexpr bool POLY2GO = W >= 16;
    double poly_cur = prm.max_err + 1.0;
    bool poly_paused = false;
    auto eval_poly = [&](const bool first_go) {
        // =========================================================================================
        // Polynomial (Catmull-Rom) candidate: polynomial.rs:209-277
        // =========================================================================================
        if (run_poly) {
            bool poly_pruned = false;
            if (smax == smin) {
                poly_K = 0;  // polynomial.rs:210-213
                poly_step = 1;
            } else if (!bounded) {
                // Compressor::compress -> polynomial() -> compress_hinted(baseline points), no error loop
                // (polynomial.rs:307-314,407-413)
                poly_step = P.pstep[0];
                poly_K = P.pK[0];
            } else {
                // the (step, K, ...) of every trip come from the per-length table (DevPlan::pstep ...)
                double2 *mm = (double2 *)AB;  // per-segment Hermite tangents (m0, m1); AB is free here
                const double inv_n = P.inv_n;
                const bool pfast = fabs(smin) < 1e150 && fabs(smax) < 1e150;  // no overflow/NaN out of the spline
                const bool nozero = smin > 0.0 || smax < 0.0;                  // every 1/|g| is finite
                double cur = POLY2GO ? poly_cur : prm.max_err + 1.0;
                while (round(cur * 10000.0) > prm.poly_q_hi) {  // polynomial.rs:231: target < round(err, 4)
                    const uint32_t ti = poly_trips;  // 0 .. 22
                    const uint32_t step = P.pstep[ti];
                    const uint32_t K = P.pK[ti];
                    if (prune && !can_win(poly_payload_size(step, K), 1)) { poly_pruned = true; break; }
                    ++poly_trips;
                    if (W == 1) frame_prio(poly_trips);
                    poly_step = step;
                    poly_K = K;
                    // A trip whose (step, K) are the trip's before -- n / points rounds to the same step for two point
                    // counts in a row: 256 samples, 103 and 128 points, step 2 -- evaluates the same spline on the same
                    // samples; the reference does it again (polynomial.rs:231-270), here the error (or the lower bound that
                    // already failed it) is simply kept: the loop only got here because that value failed.
                    const bool same_trip = ti > 0 && step == P.pstep[ti - 1] && K == P.pK[ti - 1];
                    if (same_trip) {
                    } else if (IDW && step > 1 && idw) {
                        // polynomial.rs:375-393 + inverse_distance_weight 0.1.1 (oracle: poly_idw_to_data):
                        // every sample sums over ALL K points in ascending order; an exact hit returns the
                        // point's value.  O(n K) per trip, forced `--compressor idw` only.
                        double s = 0.0;
#pragma unroll
                        for (int m = 0; m < SPL; ++m) {
                            const uint32_t j = tid + m * T;
                            if (j >= pre && j < pre + n) {
                                const double x = (double)(j - pre);
                                double num = 0.0, den = 0.0, hitv = 0.0;
                                bool hit = false;
                                for (uint32_t k = 0; k < K; ++k) {
                                    const uint32_t pk = (k == K - 1) ? (n - 1) : k * step;
                                    const double v = xs[pk];
                                    const double d = fabs((double)pk - x);
                                    if (d == 0.0) {
                                        if (!hit) { hit = true; hitv = v; }
                                    } else if (!hit) {
                                        const double w = 1.0 / (d * d);
                                        num += w * v;
                                        den += w;
                                    }
                                }
                                const double sv = hit ? hitv : num / den;
                                double o = div1e5(round(sv * 100000.0));
                                if (o < smin) o = smin;
                                else if (o > smax) o = smax;
                                s += fabs(o - g[m]) * inv[m];
                            }
                        }
                        s = block_sum_f64<W>(s, red, parity);
                        cur = s * inv_n;
                    } else if (step > 1) {
                        // keys: T(k) = k*step, T(K-1) = n-1.  Catmull-Rom on segments 1..K-3, linear on
                        // the first and the last one (polynomial.rs:349-353).  Everything that is the
                        // same for all samples of a segment (the tangents m0, m1) or for all samples at
                        // the same offset r inside a segment (the four Hermite basis values) is computed
                        // once, in the crate's operation order (oracle: cubic_hermite), so each sample's
                        // value keeps the oracle's bits.
                        rle_early = false;  // the tangent / basis tables live in AB
                        const uint32_t magic = P.pmagic[ti];
                        const uint32_t gapL = P.pgap[ti];  // length of the last segment
                        const double stepd = (double)step, gapLd = (double)gapL;
                        const double ry = P.pry[ti], ryL = P.pryL[ti];
                        const uint32_t mm_bytes = (16 * K + 15) & ~15u;
                        const bool use_tab = (K >= 6) && (mm_bytes + 32 * step <= ab_bytes);
                        double4 *hb = (double4 *)(AB + mm_bytes);  // basis (h00, h10, h01, h11) per offset r
                        __syncthreads();
                        for (uint32_t sg = tid + 1; sg + 2 < K; sg += T) {
                            const uint32_t t0i = sg * step;
                            const uint32_t t1i = (sg + 1 == K - 1) ? (n - 1) : (sg + 1) * step;
                            const uint32_t tmi = (sg - 1) * step;
                            const uint32_t tpi = (sg + 2 == K - 1) ? (n - 1) : (sg + 2) * step;
                            const double t0 = (double)t0i, t1 = (double)t1i;
                            const double v0 = xs[t0i], v1 = xs[t1i], vm = xs[tmi], vp = xs[tpi];
                            double2 t;
                            t.x = (v1 - vm) / (t1 - (double)tmi) * (t1 - t0);
                            t.y = (vp - v0) / ((double)tpi - t0) * (t1 - t0);
                            mm[sg] = t;
                        }
                        if (use_tab) {
                            for (uint32_t r = tid; r < step; r += T) {
                                const double nt = div_small((double)r, stepd, ry);
                                const double t2 = nt * nt;
                                const double t3 = t2 * nt;
                                const double two_t3 = t3 * 2.0;
                                const double two_t2 = t2 * 2.0;
                                const double three_t2 = t2 * 3.0;
                                double4 h;
                                h.x = two_t3 - three_t2 + 1.0;
                                h.y = t3 - two_t2 + nt;
                                h.z = three_t2 - two_t3;
                                h.w = t3 - t2;
                                hb[r] = h;
                            }
                        }
                        __syncthreads();
                        // one sample's share of the MAPE sum (slice m of the padded layout)
                        auto term = [&](const int m) -> double {
                            const uint32_t j = tid + m * T;
                            if (!(j >= pre && j < pre + n)) return 0.0;
                            const uint32_t i = j - pre;
                            double sv;
                            if (i == n - 1) {
                                sv = xs[n - 1];
                            } else {
                                uint32_t sg = __umulhi(i, magic);  // i / step
                                if (sg > K - 2) sg = K - 2;
                                const uint32_t t0i = sg * step;
                                const bool last = (sg == K - 2);
                                const uint32_t t1i = last ? (n - 1) : t0i + step;
                                const double v0 = xs[t0i], v1 = xs[t1i];
                                if (sg > 0 && !last) {  // Catmull-Rom: sg in 1..K-3
                                    const double2 t = mm[sg];
                                    if (use_tab) {
                                        const double4 h = hb[i - t0i];
                                        sv = v0 * h.x + t.x * h.y + v1 * h.z + t.y * h.w;
                                    } else {
                                        const double nt = div_small((double)(i - t0i), stepd, ry);
                                        const double t2 = nt * nt;
                                        const double t3 = t2 * nt;
                                        const double two_t3 = t3 * 2.0;
                                        const double two_t2 = t2 * 2.0;
                                        const double three_t2 = t2 * 3.0;
                                        sv = v0 * (two_t3 - three_t2 + 1.0) + t.x * (t3 - two_t2 + nt) +
                                             v1 * (three_t2 - two_t3) + t.y * (t3 - t2);
                                    }
                                } else {
                                    const double nt = div_small((double)(i - t0i), last ? gapLd : stepd,
                                                            last ? ryL : ry);
                                    sv = v0 * (1.0 - nt) + v1 * nt;
                                }
                            }
                            double o = div1e5(round(sv * 100000.0));  // utils/mod.rs:66-74
                            if (pfast) {
                                o = fmin(fmax(o, smin), smax);  // o is finite: same as the compares
                            } else {
                                if (o < smin) o = smin;
                                else if (o > smax) o = smax;
                            }
                            return fabs(o - g[m]) * inv[m];
                        };
                        // Every term is >= 0 and, with a finite range and no zero sample, finite: the sum
                        // over one slice of the samples is then a lower bound of the trip's error (floating
                        // point addition is monotone), and if that alone already fails the trip, nothing the
                        // other slices add can change the decision.  The first trips of a busy frame end here.
                        double t1 = 0.0;
                        bool fails = false;
                        if (SPL >= 2 && pfast && nozero) {
                            t1 = term(1);
                            const double lb = block_sum_f64<W>(t1, red, parity) * inv_n;
                            if (round(lb * 10000.0) > prm.poly_q_hi) { cur = lb; fails = true; }
                        }
                        if (!fails) {
                            double s = 0.0;
#pragma unroll
                            for (int m = 0; m < SPL; ++m) s += (m == 1 && SPL >= 2 && pfast && nozero) ? t1 : term(m);
                            s = block_sum_f64<W>(s, red, parity);
                            cur = s * inv_n;
                        }
                    }
                    if (poly_trips > 22) {  // polynomial.rs:255-263: the jumps are spent
                        if (round(cur * 10000.0) < prm.poly_q_lo) break;  // target > round(err, 4)
                        poly_step = 1; poly_K = n; cur = 0.0;
                        break;
                    }
                    if (K == n) { cur = 0.0; break; }  // polynomial.rs:264-269
                    if (POLY2GO && first_go && round(cur * 10000.0) > prm.poly_q_hi) {  // failed: the FFT candidate goes next
                        poly_cur = cur;
                        poly_paused = true;
                        return;
                    }
                }
                poly_err = cur;
            }
            poly_size = poly_payload_size(poly_step, poly_K);
            poly_done = !poly_pruned;
            if (prune && poly_done && poly_err <= me) offer(poly_size, 1);
            dg.poly_size = poly_size; dg.poly_trips = (uint16_t)poly_trips;
            dg.poly_step = (uint16_t)poly_step; dg.poly_points = poly_K; dg.poly_err = poly_err;
        }

    };

    if (poly_first) eval_poly(true);
    PH(4);
    if (!LEAN && prm.debug_stop == 13) return;
    // =========================================================================================
    // FFT candidate: fft.rs:288-362
    // =========================================================================================
    if (run_fft) {
        if (mxf == mnf) {
            fft_k = 0;  // fft.rs:289-292 ; error None -> 0.0 (fft.rs:523)
            fft_size = 1 + 1 + 8;
            fft_done = true;
        } else if (prune && !can_win(1 + 1 + 9 + 8, 0)) {
            // even a single stored bin is larger than a payload that already passes
        } else {
            rle_early = false;  // AB and the twiddle region (aux) are about to be overwritten
            {
                const float2 *twp = twpool + P.tw_off;
                for_strided<FIX ? (int)cL : 0, T, SPL>(tid, L, [&](uint32_t j) { tw[j] = twp[j]; });
            }
            float2 *spec;
            if (FIX && chalf) {
                float *Af = (float *)A;
#pragma unroll
                for (int m = 0; m < SPL; ++m) {
                    const uint32_t j = tid + m * T;
                    if (j < L) Af[j] = (float)g[m];
                }
                __syncthreads();
                float2 *Z = fft_forward_fixed<W, FIX ? cM : 2, 2>(A, B, tw);
                spec = (Z == A) ? B : A;
                fft_untangle_fixed<W, FIX ? cM : 2>(Z, spec, tw);
            } else if (FIX) {  // odd transform length: full complex transform of the real signal
#pragma unroll
                for (int m = 0; m < SPL; ++m) {
                    const uint32_t j = tid + m * T;
                    if (j < L) A[j] = make_float2((float)g[m], 0.0f);
                }
                __syncthreads();
                spec = fft_forward_fixed<W, FIX ? cM : 3, 1>(A, B, tw);
            } else if (P.direct) {
                __syncthreads();
                dft_direct<W>(P_GENERIC, xs, A, tw);
                spec = A;
            } else if (P.half) {
                float *Af = (float *)A;  // z[j] = g[2j] + i g[2j+1]  ==  g stored as consecutive f32
#pragma unroll
                for (int m = 0; m < SPL; ++m) {
                    const uint32_t j = tid + m * T;
                    if (j < L) Af[j] = (float)g[m];
                }
                __syncthreads();
                float2 *Z = fft_forward<W>(P_GENERIC, A, B, tw);
                spec = (Z == A) ? B : A;
                fft_untangle<W>(P_GENERIC, Z, spec, tw);
            } else {
#pragma unroll
                for (int m = 0; m < SPL; ++m) {
                    const uint32_t j = tid + m * T;
                    if (j < L) A[j] = make_float2((float)g[m], 0.0f);
                }
                __syncthreads();
                spec = fft_forward<W>(P_GENERIC, A, B, tw);
            }
            PH(5);
            if (!LEAN && prm.debug_stop == 4) return;
            // Order of admission: descending f32 norm = hypot(re, im) (fft.rs:88-106), ties by
            // ascending position.  W == 1: each lane keeps the norms of its KPL bins in registers
            // and the next bin is pulled by two wavefront reductions when the ladder asks for it;
            // W > 1: one sort of 64-bit keys up front.
            constexpr int KPL = (SPL * 32 + 1 + 63) / 64;
            uint32_t nb[KPL];
            float bre[KPL], bim[KPL];  // W == 1: the lane's bins, so an admitted bin is one v_readlane away
            uint64_t *keys = (uint64_t *)(spec == A ? B : A);
            bool heap_order = false;  // bit-equal norms met: the admission order is replayed from the reference's heap
            uint32_t hlen = 0;        // entries left in the heap (keys[]) once heap_order is set
            uint32_t sorted_n = 0;    // W > 1: keys[0 .. sorted_n) hold the (norm desc, position asc) order
            uint32_t nz = 0;
            if (W == 1) {
#pragma unroll
                for (int m = 0; m < KPL; ++m) {
                    const uint32_t k = tid + 64 * m;
                    nb[m] = 0;
                    bre[m] = 0.0f;
                    bim[m] = 0.0f;
                    if (k < bins) {
                        const float2 z = spec[k];
                        bre[m] = z.x;
                        bim[m] = z.y;
                        nb[m] = __float_as_uint(
                            (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y));
                        nz += (z.x != 0.0f || z.y != 0.0f) ? 1u : 0u;
                    }
                }
            } else {
                for (uint32_t k = tid; k < bins; k += T) {
                    const float2 z = spec[k];
                    const float nrm = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);
                    keys[k] = ((uint64_t)(~__float_as_uint(nrm)) << 32) | (uint64_t)k;
                    nz += (z.x != 0.0f || z.y != 0.0f) ? 1u : 0u;
                }
                __syncthreads();
            }
            const uint32_t Z = block_sum_u32<W>(nz, red, parity);  // fft.rs:249-252 zero cut-off
            // W > 1: the order of admission is built when the ladder asks for it, and in two stages.  Most frames end
            // within their first trips, so the first stage orders the bins three trips can take (mf + 2 dk1); a ladder
            // that goes on gets the full order of the kcap bins it can admit at most.  Either way: the keys are
            // distinct (the position is part of the key), so the `want` smallest ones are exactly those <= the want-th
            // smallest key.  An 8-bit radix select closes in on it digit by digit (key bits 16..31 are zero: six
            // counting passes at most) and stops as soon as the bins above the digit plus the bins in it are few enough
            // to sort -- typically after the two leading bytes of the norm; every key up to the digit moves to the
            // front and is sorted: keys[0 .. sorted_n) is the complete head of the (norm desc, position asc) order.
            bool keys_fresh = true;  // keys[] holds every bin (as the norm pass left it)
            auto build_order = [&](const uint32_t want) {
                if (!keys_fresh) {
                    __syncthreads();
                    for (uint32_t k = tid; k < bins; k += T) {
                        const float2 z = spec[k];
                        const float nrm = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);
                        keys[k] = ((uint64_t)(~__float_as_uint(nrm)) << 32) | (uint64_t)k;
                    }
                    __syncthreads();
                }
                keys_fresh = false;
                uint32_t nk = bins;
                if (bins > want && want > 0) {
                    uint32_t *hist = (uint32_t *)(smem + o_hist);  // 256 counters, then {digit, below, count, in digit}
                    const uint32_t few = 2 * want + 16;
                    uint64_t prefix = 0, resolved = 0;
                    uint32_t remaining = want;
                    for (int shift = 56; shift >= 0; shift -= 8) {
                        if (shift == 24 || shift == 16) continue;
                        for (uint32_t i = tid; i < 256; i += T) hist[i] = 0;
                        __syncthreads();
                        for (uint32_t k = tid; k < bins; k += T) {
                            const uint64_t v = keys[k];
                            if ((v & resolved) == prefix) atomicAdd(&hist[(uint32_t)(v >> shift) & 255u], 1u);
                        }
                        __syncthreads();
                        if (tid < 64) {  // first digit whose running count reaches `remaining`
                            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                            const uint32_t sum = c0 + c1 + c2 + c3;
                            const uint32_t incl = wave_incl_scan_u32(sum);
                            const uint64_t reach = __ballot(incl >= remaining);
                            if (reach && tid == (uint32_t)__builtin_ctzll(reach)) {
                                uint32_t cum = incl - sum, d = 4 * tid, in = c0;
                                if (cum + c0 < remaining) { cum += c0; ++d; in = c1;
                                    if (cum + c1 < remaining) { cum += c1; ++d; in = c2;
                                        if (cum + c2 < remaining) { cum += c2; ++d; in = c3; } } }
                                hist[256] = d;
                                hist[257] = cum;
                                hist[259] = in;
                            }
                        }
                        __syncthreads();
                        prefix |= (uint64_t)hist[256] << shift;
                        resolved |= 0xffull << shift;
                        remaining -= hist[257];
                        nk = (want - remaining) + hist[259];  // bins above the digit, bins in it
                        __syncthreads();
                        if (nk <= few) break;
                    }
                    constexpr int KPT = SPL / 2 + 1;  // bins <= 32 * W * SPL + 1
                    uint64_t mine[KPT];
#pragma unroll
                    for (int c = 0; c < KPT; ++c) {
                        const uint32_t k = tid + c * T;
                        mine[c] = k < bins ? keys[k] : ~0ull;
                    }
                    if (tid == 0) hist[258] = 0;
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < KPT; ++c)
                        if ((mine[c] & resolved) <= prefix) keys[atomicAdd(&hist[258], 1u)] = mine[c];
                    __syncthreads();
                }
                if (nk <= 1024 && 5 * nk / 2 + 2 <= bins) {
                    block_rank_sort<W>(keys, keys + nk, nk);  // (behind the nk keys the buffer is free: keys_fresh is off)
                } else {
                    uint32_t p2 = 1;
                    while (p2 < nk) p2 <<= 1;
                    block_sort<W, true>(keys, nullptr, nk, p2);
                }
                sorted_n = nk;
            };

            PH(6);
            if (!LEAN && prm.debug_stop == 5) return;
            bool fft_pruned = false;
            float acc[SPL];
#pragma unroll
            for (int m = 0; m < SPL; ++m) acc[m] = 0.0f;
            float dc = 0.0f;
            const double mxd = (double)mxf, mnd = (double)mnf;
            const double Ld = (double)L;
            // min/max far from f32 overflow: no NaN can come out of the transform, so the clamp is a plain
            // min(max()) (a NaN sample still poisons the sum through o - g); otherwise compare-select
            const bool fast_clamp = fabsf(mxf) < 1e30f && fabsf(mnf) < 1e30f;
            const double invL = 1.0 / Ld;
            const uint32_t magicL = FIX ? (uint32_t)(0x100000000ull / (FIX ? cL : 1)) + 1u : P.magicL;
            uint32_t used = 0, jump = 0, big = 0;
            double cur = prm.max_err + 1.0;
            // bounded: fft.rs:334 loop.  Unbounded (FFT::compress, fft.rs:366-388): one pass that only
            // admits the max(3, n/100) largest bins; nothing is reconstructed or measured.
            while (bounded ? (prm.max_err_m < sat_i32(cur * 1000.0)) : (fft_trips == 0)) {
                const uint32_t K = min(mf + jump, Z);
                if (prune && !can_win(1 + vlen(K) + 9 * K + 8, 0)) { fft_pruned = true; break; }
                ++fft_trips;
                if (W == 1) frame_prio(fft_trips);
                if (W > 1 && !heap_order && K > sorted_n) {
                    const uint32_t first = mf + 2 * dk1;
                    build_order((K <= first && first < kcap) ? first : kcap);
                }
                if (W > 1 && !heap_order) {
                    // Bit-equal norms among the bins this trip admits, or between the last of them and the next
                    // one in line: the reference's order there is the BinaryHeap's, not (norm, position).  The
                    // pairs up to (used - 1, used) were looked at by the trips before.  Rare -- two of the K
                    // largest f32 norms have to collide -- and then the frame replays the heap (hp_*,
                    // atsc_device.h): rebuilt level by level by the whole workgroup, popped by one wavefront,
                    // pop i left at keys[bins - 1 - i].  The bins admitted so far had distinct norms, i.e. they
                    // are the heap's first pops.
                    uint32_t tied = 0;
                    for (uint32_t i = used + tid; i < K && i + 1 < sorted_n; i += T) {
                        const uint32_t hi = (uint32_t)(keys[i] >> 32);  // (zero norms, 0xFFFFFFFF, are never admitted)
                        tied |= (hi == (uint32_t)(keys[i + 1] >> 32) && hi != 0xFFFFFFFFu) ? 1u : 0u;
                    }
                    if (K == sorted_n && sorted_n < bins && K > used) {
                        // the ladder takes the last key the select kept: a bin it left out may hold the same norm
                        const uint32_t hi = (uint32_t)(keys[sorted_n - 1] >> 32), lastpos = (uint32_t)(keys[sorted_n - 1] & 0xffffffffu);
                        if (hi != 0xFFFFFFFFu)
                            for (uint32_t k = tid; k < bins; k += T) {
                                const float2 zz = spec[k];
                                const float nrm = (float)sqrt((double)zz.x * (double)zz.x + (double)zz.y * (double)zz.y);
                                tied |= ((~__float_as_uint(nrm)) == hi && k != lastpos) ? 1u : 0u;
                            }
                    }
                    tied = block_sum_u32<W>(tied ? 1u : 0u, red, parity);
                    if (!LEAN && prm.debug_stop == -2) tied = 0;  // A/B aid: (norm, position) order throughout
                    if (tied) {
                        __syncthreads();
                        for (uint32_t k = tid; k < bins; k += T) {
                            const float2 zz = spec[k];
                            const float nrm = (float)sqrt((double)zz.x * (double)zz.x + (double)zz.y * (double)zz.y);
                            keys[k] = ((uint64_t)__float_as_uint(nrm) << 32) | (uint64_t)k;
                        }
                        __syncthreads();
                        hp_rebuild_parallel(keys, bins, tid, (uint32_t)T);
                        hlen = bins;
                        if (tid < 64)
                            for (uint32_t i = 0; i < used; ++i) (void)hp_pop(keys, hlen);
                        hlen = bins - used;
                        heap_order = true;
                    }
                }
                if (W > 1 && heap_order) {
                    if (tid < 64)
                        for (uint32_t i = used; i < K; ++i) (void)hp_pop(keys, hlen);
                    hlen = bins - K;
                    __syncthreads();
                }
                if constexpr (W >= 8) {  // (W == 4: the extra registers would cost a workgroup per CU)
                    // The trip's bins are staged first: thread i looks bin i up (order -> position -> spectrum), files
                    // it in sel[] and leaves its position, its coefficient (fft.rs:401-422 mirror: bin 0 and, for even
                    // L, bin L/2 contribute once; the 1/L of fft.rs:343 is folded in) and its twiddle stride in LDS.
                    // Every thread then walks the staged list: one uniform LDS read per bin instead of three dependent
                    // round trips, and the f64 coefficient arithmetic once per bin instead of once per wavefront.
                    struct Stg { uint32_t pos, stp; float a, b; };
                    Stg *stg = (Stg *)(smem + o_hist);  // 1040 bytes: 64 entries (the radix select is done with them)
                    // (a handful of bins -- the later trips of a 1024-sample frame -- are quicker one by one, below)
                    while (K - used >= 8) {
                        const uint32_t cnt = min(K - used, 64u);
                        __syncthreads();
                        if (tid < cnt) {
                            const uint32_t i = used + tid;
                            const uint32_t pos = (uint32_t)(keys[heap_order ? bins - 1 - i : i] & 0xffffffffu);
                            const float2 z = spec[pos];
                            sel[i].pos = pos; sel[i].re = z.x; sel[i].im = z.y;
                            const double cf = ((pos == 0 || 2 * pos == L) ? 1.0 : 2.0) * invL;
                            Stg e;
                            e.pos = pos;
                            e.stp = mod_magic(pos * (uint32_t)T, L, magicL);
                            e.a = (float)(cf * (double)z.x);
                            e.b = (float)(cf * (double)z.y);
                            stg[tid] = e;
                        }
                        __syncthreads();
#pragma unroll 2
                        for (uint32_t i = 0; i < cnt; ++i) {
                            const Stg e = stg[i];
                            const bool isdc = e.pos == 0;  // its share is added at evaluation time (acc + dc)
                            dc = isdc ? e.a : dc;
                            const float a = isdc ? 0.0f : e.a, b = isdc ? 0.0f : e.b;
                            uint32_t idx = mod_magic(e.pos * tid, L, magicL);
#pragma unroll
                            for (int m = 0; m < SPL; ++m) {
                                const float2 w = tw[idx];
                                acc[m] = fmaf(a, w.x, acc[m]);
                                acc[m] = fmaf(-b, w.y, acc[m]);
                                idx += e.stp;
                                idx = min(idx, idx - L);  // idx < 2L: unsigned wrap picks the reduced value
                            }
                        }
                        used += cnt;
                    }
                }
                for (; used < K; ++used) {
                    uint32_t pos;
                    float2 z;
                    if (W == 1) {
                        pos = 0;
                        if (!heap_order) {
                            uint32_t lm = nb[0];
#pragma unroll
                            for (int m = 1; m < KPL; ++m) lm = max(lm, nb[m]);
                            const uint32_t wm = wave_max_u32(lm);
                            // lowest position holding the largest norm, and how many bins hold it
                            uint32_t mult = 0;
#pragma unroll
                            for (int m = KPL - 1; m >= 0; --m) {
                                const uint64_t bal = __ballot(nb[m] == wm);
                                mult += (uint32_t)__popcll(bal);
                                if (bal) pos = 64u * (uint32_t)m + (uint32_t)__builtin_ctzll(bal);
                            }
                            if (mult > 1) {
                                // Bit-equal norms: from here on the bins come out of the reference's heap.  The
                                // bins admitted so far had distinct norms, i.e. they are the heap's first pops.
                                __syncthreads();
#pragma unroll
                                for (int m = 0; m < KPL; ++m) {
                                    const uint32_t k = tid + 64 * m;
                                    if (k < bins) {
                                        const float nrm = (float)sqrt((double)bre[m] * (double)bre[m] +
                                                                      (double)bim[m] * (double)bim[m]);
                                        keys[k] = ((uint64_t)__float_as_uint(nrm) << 32) | (uint64_t)k;
                                    }
                                }
                                __syncthreads();
                                hp_rebuild_parallel(keys, bins, tid, 64u);
                                hlen = bins;
                                for (uint32_t i = 0; i < used; ++i) (void)hp_pop(keys, hlen);
                                heap_order = true;
                            }
                        }
                        if (heap_order) pos = (uint32_t)(hp_pop(keys, hlen) & 0xffffffffu);
                        float zr = 0.0f, zi = 0.0f;
#pragma unroll
                        for (int m = 0; m < KPL; ++m) {
                            if (tid + 64 * m == pos) nb[m] = 0;
                            if ((pos >> 6) == (uint32_t)m) { zr = bre[m]; zi = bim[m]; }  // uniform select
                        }
                        z = make_float2(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(zr), pos & 63)),
                                        __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(zi), pos & 63)));
                    } else {
                        pos = (uint32_t)(keys[heap_order ? bins - 1 - used : used] & 0xffffffffu);
                        z = spec[pos];
                    }
                    if (tid == 0) { sel[used].pos = pos; sel[used].re = z.x; sel[used].im = z.y; }
                    big += (pos >= 251) ? 1u : 0u;
                    // fft.rs:401-422 mirror: bin 0 and (for even L) bin L/2 contribute once;
                    // the 1/L of fft.rs:343 is folded into the coefficient
                    const double cf = ((pos == 0 || 2 * pos == L) ? 1.0 : 2.0) * invL;
                    const float a = (float)(cf * (double)z.x);
                    const float b = (float)(cf * (double)z.y);
                    if (pos == 0) {
                        dc = a;
                    } else {
                        uint32_t idx = mod_magic(pos * tid, L, magicL);
                        const uint32_t stp = mod_magic(pos * (uint32_t)T, L, magicL);
#pragma unroll
                        for (int m = 0; m < SPL; ++m) {
                            // lanes beyond L accumulate too (idx < L always): their acc is never
                            // looked at, or weighs 0 in the error sum, and no exec masking is needed
                            const float2 w = tw[idx];
                            acc[m] = fmaf(a, w.x, acc[m]);
                            acc[m] = fmaf(-b, w.y, acc[m]);
                            idx += stp;
                            idx = min(idx, idx - L);  // idx < 2L: unsigned wrap picks the reduced value
                        }
                    }
                }
                PH(7);
                if (!bounded) { cur = 0.0; break; }
                double s = 0.0;
                // fft.rs:208-218.  v is an f32, so v * 1e5 is exact (<= 41 significant bits) and
                // round-half-away equals trunc(x + copysign(0.5, x)) (checked over the f32 range)
                if (fast_clamp) {
                    // every o is finite, and the lanes beyond L carry g = 1, 1/|g| = 0: no guard
#pragma unroll
                    for (int m = 0; m < SPL; ++m) {
                        const double x5 = (double)(acc[m] + dc) * 100000.0;
                        double o = div1e5(trunc(x5 + copysign(0.5, x5)));
                        o = fmin(fmax(o, mnd), mxd);
                        s += fabs(o - g[m]) * inv[m];  // utils/error.rs:104-116
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < SPL; ++m) {
                        if (tid + m * T < L) {
                            const double x5 = (double)(acc[m] + dc) * 100000.0;
                            double o = div1e5(trunc(x5 + copysign(0.5, x5)));
                            if (o > mxd) o = mxd;
                            if (o < mnd) o = mnd;
                            s += fabs(o - g[m]) * inv[m];
                        }
                    }
                }
                s = block_sum_f64<W>(s, red, parity);
                cur = s * invL;  // mean over the L padded samples (utils/error.rs:115); 1/L rounded once
                PH(8);
                if (fft_trips <= 17) jump += dk1;       // fft.rs:348-352
                else if (fft_trips <= 22) jump += dk2;
                else break;
            }
            fft_err = cur;
            fft_k = used;
            if (W > 1) {  // positions that take a three-byte varint
                uint32_t b3 = 0;
                for (uint32_t i = tid; i < used; i += T) b3 += (sel[i].pos >= 251) ? 1u : 0u;
                big = block_sum_u32<W>(b3, red, parity);
            }
            fft_size = 1 + vlen(used) + 9 * used + 2 * big + 8;
            fft_done = !fft_pruned;
            __syncthreads();  // sel[] is complete; AB may be reused from here on
        }
        if (prune && fft_done && fft_err <= me) offer(fft_size, 0);
        dg.fft_size = fft_size; dg.fft_trips = (uint16_t)fft_trips; dg.fft_k = (uint16_t)fft_k;
        dg.fft_err = fft_err;
    }


    PH(9);
    if (!LEAN && prm.debug_stop == 14) return;
    if (!poly_first || (POLY2GO && poly_paused)) eval_poly(false);
    PH(10);
    if (!LEAN && prm.debug_stop == 15) return;

    // ---- RLE with many runs: exact size only if its bound can still win ----
    if (run_rle && rle_pending) {
        if (!prune || can_win(rle_lb, 2)) {
            // Exact size without sorting: count the distinct run values and their multiplicities
            // in an LDS hash table (AB is free: ab_bytes / 4 >= 2n slots of run-end indices).
            uint32_t *tab = (uint32_t *)AB;
            const uint32_t H = ab_bytes >> 2;
            __syncthreads();
            for (uint32_t i = tid; i < H; i += T) tab[i] = 0xFFFFFFFFu;
            for (uint32_t j = tid; j < n; j += T) aux[j] = 0;
            __syncthreads();
            uint32_t dnew = 0;
            for (uint32_t j = tid; j < n; j += T) {
                if (j + 1 >= n || xs[j + 1] != xs[j]) {
                    const uint64_t key = (uint64_t)__double_as_longlong(xs[j]);
                    uint32_t h = __umulhi(((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u, H);
                    for (;;) {
                        const uint32_t old = atomicCAS(&tab[h], 0xFFFFFFFFu, j);
                        if (old == 0xFFFFFFFFu) { atomicAdd(&aux[j], 1u); ++dnew; break; }
                        if ((uint64_t)__double_as_longlong(xs[old]) == key) { atomicAdd(&aux[old], 1u); break; }
                        h = (h + 1 == H) ? 0 : h + 1;
                    }
                }
            }
            __syncthreads();
            uint32_t hb = 0;
            for (uint32_t j = tid; j < n; j += T)
                if (aux[j]) hb += value_bytes(bitdepth, xs[j]) + vlen(aux[j]);
            // (distinct << 18 | header bytes): header bytes <= 11 * 4096 < 2^18
            const uint32_t pk2 = block_sum_u32<W>((dnew << 18) | hb, red, parity);
            rle_D = pk2 >> 18;
            rle_size = 2 + vlen(rle_D) + (pk2 & 0x3ffffu) + rle_ib;
            if (prune) offer(rle_size, 2);
        } else {
            rle_size = rle_lb;  // a lower bound that already cannot win
        }
    }
    dg.rle_size = rle_size;

    PH(11);
    if (!LEAN && prm.debug_stop == 8) return;
    // =========================================================================================
    // selection: frame/mod.rs:113-147
    // =========================================================================================
    int chosen;
    double chosen_err;
    if (mode == ATSC_AUTO) {
        if (prune) {
            chosen = best_owner == 0 ? ATSC_FFT : best_owner == 1 ? ATSC_POLYNOMIAL : ATSC_RLE;
        } else {
            // max_error below zero (or NaN): nothing passes, not even RLE's 0.0, and the reference takes
            // the smallest payload of all three (frame/mod.rs:128-135); everything above ran unpruned
            chosen = ATSC_FFT;
            uint32_t bs = fft_size;
            if (poly_size < bs) { chosen = ATSC_POLYNOMIAL; bs = poly_size; }
            if (rle_size < bs) { chosen = ATSC_RLE; bs = rle_size; }
        }
        chosen_err = chosen == ATSC_FFT ? fft_err : chosen == ATSC_POLYNOMIAL ? poly_err : 0.0;
    } else {
        chosen = mode;
        chosen_err = mode == ATSC_FFT ? fft_err : (mode == ATSC_POLYNOMIAL || mode == ATSC_IDW) ? poly_err : 0.0;
    }

    // =========================================================================================
    // emit the chosen payload into the frame's slot
    // =========================================================================================
    uint32_t out_len = 0;
    if (!LEAN && prm.trial) {
        if (tid == 0) {
            res[fid].err = chosen_err;
            res[fid].len = 0;
            res[fid].chosen = (uint32_t)chosen;
        }
        return;
    }
    if (chosen == ATSC_FFT) {  // fft.rs:119-130
        const uint32_t hdr = 1 + vlen(fft_k);
        for (uint32_t i = tid; i < fft_k; i += T) aux[i] = vlen(sel[i].pos) + 8;
        __syncthreads();
        const uint32_t body = block_excl_scan<W>(aux, fft_k, wsum);
        for (uint32_t i = tid; i < fft_k; i += T) {
            uint8_t *p = out + hdr + aux[i];
            p += put_varint(p, sel[i].pos & 0xffffu);  // `pos as u16` (fft.rs:242)
            put_f32(p, sel[i].re);
            put_f32(p + 4, sel[i].im);
        }
        if (tid == 0) {
            out[0] = 15;
            put_varint(out + 1, fft_k);
            put_f32(out + hdr + body, mxf);
            put_f32(out + hdr + body + 4, mnf);
        }
        out_len = hdr + body + 8;
    } else if (chosen == ATSC_POLYNOMIAL || chosen == ATSC_IDW) {  // polynomial.rs:54-87
        const uint32_t hdr = 2 + vlen(poly_K);
        uint32_t body;
        if (bitdepth == 0 || bitdepth == 3) {
            const uint32_t vbytes = bitdepth == 0 ? 8u : 1u;
            body = poly_K * vbytes;
            for (uint32_t k = tid; k < poly_K; k += T) {
                const uint32_t t = (k == poly_K - 1) ? (n - 1) : k * poly_step;
                put_value(out + hdr + k * vbytes, bitdepth, xs[t]);
            }
        } else {
            for (uint32_t k = tid; k < poly_K; k += T) {
                const uint32_t t = (k == poly_K - 1) ? (n - 1) : k * poly_step;
                aux[k] = value_bytes(bitdepth, xs[t]);
            }
            __syncthreads();
            body = block_excl_scan<W>(aux, poly_K, wsum);
            for (uint32_t k = tid; k < poly_K; k += T) {
                const uint32_t t = (k == poly_K - 1) ? (n - 1) : k * poly_step;
                put_value(out + hdr + aux[k], bitdepth, xs[t]);
            }
        }
        if (tid == 0) {
            out[0] = idw ? 1 : 0;  // PolynomialType::{Polynomial, Idw}
            out[1] = (uint8_t)bitdepth;
            put_varint(out + 2, poly_K);
            put_f64(out + hdr + body, smin);
            put_f64(out + hdr + body + 8, smax);
            out[hdr + body + 16] = (uint8_t)poly_step;  // `step as u8`
        }
        out_len = hdr + body + 17;
    } else if (rle_small) {  // RLE with one run per lane: rle.rs:40-67
        if (!rle_early) rle_small_build();  // a ladder overwrote the sorted records
        const RleLane g = rle_small_group();
        const uint32_t hdr = 2 + vlen(g.D);
        const uint32_t st = g.rec >> 16, vl = g.live ? vlen(st) : 0u;
        const uint32_t vls = wave_incl_scan_u32(vl);  // index varint bytes up to and including this run
        const uint32_t hbs = wave_incl_scan_u32(g.hb);  // header bytes up to and including this run's group
        if (g.head) {
            uint8_t *p = out + hdr + (hbs - g.hb) + (vls - vl);
            p += put_value(p, bitdepth, xs[g.rec & 0xffffu]);
            put_varint(p, g.cnt);
        }
        if (g.live) put_varint(out + hdr + hbs + (vls - vl), st);
        if (tid == 0) {
            out[0] = 60;
            out[1] = (uint8_t)bitdepth;
            put_varint(out + 2, g.D);
        }
        out_len = hdr + g.hb_total + (uint32_t)__builtin_amdgcn_readlane((int)vls, 63);
    } else {  // RLE: rle.rs:40-67
        // runs are sorted by (value bits, start); rhp = hp[]; rph = hb[]
        uint32_t *rrec = rrec_std, *rhp = rhp_std, *rph = rph_std;
        if (rle_early) {  // sized before the ladders, and no ladder touched AB since
            rrec = (uint32_t *)AB;
            rhp = rrec + rle_R;
            rph = rrec + 2 * rle_R + 1;
        } else if (!rle_sorted) {
            rle_sort_and_group(rrec_std, rhp_std, rph_std);
        }
        const uint32_t R = rle_R, D = rle_D;
        const uint32_t hdr = 2 + vlen(D);
        // one scan for both prefixes: (group heads before i) << 16 | index varint bytes before i
        // (R <= 4096 heads, 3 * 4096 bytes: neither half overflows)
        for (uint32_t i = tid; i < R; i += T) {
            const uint32_t rec = rrec[i];
            const bool head = (i == 0 || run_key(rec) != run_key(rrec[i - 1]));
            aux[i] = (head ? 0x10000u : 0u) | vlen(rec >> 16);
        }
        __syncthreads();
        block_excl_scan<W>(aux, R, wsum);
        const uint32_t hb = block_excl_scan<W>(rph, D, wsum);
        uint32_t ibt = 0;
        for (uint32_t i = tid; i < R; i += T) {
            const uint32_t rec = rrec[i], st = rec >> 16;
            const bool head = (i == 0 || run_key(rec) != run_key(rrec[i - 1]));
            const uint32_t pk = aux[i], ps = pk & 0xffffu;
            const uint32_t gi = head ? (pk >> 16) : (pk >> 16) - 1;
            const uint32_t ghb = (gi + 1 < D ? rph[gi + 1] : hb);  // header bytes up to and incl. gi
            if (head) {
                uint8_t *p = out + hdr + rph[gi] + ps;
                p += put_value(p, bitdepth, xs[rec & 0xffffu]);
                put_varint(p, rhp[gi + 1] - rhp[gi]);
            }
            put_varint(out + hdr + ghb + ps, st);
            if (i == R - 1) ibt = ps + vlen(st);
        }
        ibt = block_sum_u32<W>(ibt, red, parity);
        if (tid == 0) {
            out[0] = 60;
            out[1] = (uint8_t)bitdepth;
            put_varint(out + 2, D);
        }
        out_len = hdr + hb + ibt;
    }
    PH(12);
#ifdef ATSC_STAMPS
    __syncthreads();
    if (FN != 0 && tid < 16) atomicAdd(&g_phase_cyc[(fid & 63u) * 16 + tid], ph_acc[tid]);
    if (FN != 0 && tid == 0 && bid < 65536) g_frame_span[span0 + 3 * bid + 1] = wall_clock64();
#endif
    if (tid == 0) {
        res[fid].err = chosen_err;
        res[fid].len = out_len;
        res[fid].chosen = (uint32_t)chosen;
        if (!LEAN && diag) diag[fid] = dg;
        if (prm.cost) prm.cost[fid] = (uint32_t)min((unsigned long long)(clock64() - t_start) >> 6, 0xFFFFFFFFull);
    }
}

// One frame per workgroup: the grid is the class's frame list.
template <int W, int SPL, bool IDW, int FN, bool LEAN_ = (FN != 0)>
__global__ __launch_bounds__(64 * W, (W == 1 && SPL <= 5 && !IDW) ? 6 : 1) void k_compress(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames,
    const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, const KParams prm, uint8_t *__restrict__ slots,
    DevResult *__restrict__ res, atsc_frame_diag *__restrict__ diag, const UniArgs uni)
{
    compress_frame<W, SPL, IDW, FN, LEAN_>(samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, blockIdx.x);
}

// Resident workgroups: as many as the GPU holds at once, each taking the next frame of the class from a counter
// until the list is done.  A workgroup slot that has to be refilled by the dispatcher stays empty for 1-2.5 us
// (tools/dispatch_probe.hip: 40960 workgroups of 10 us each with 7008 bytes of LDS keep 18.7 of a CU's 23 slots
// busy); a 256-sample frame lives 13 us.  queue[0] counts the frames handed out since the counter was made; the
// launch's share starts at q_base (the host adds every launch's count), so nothing resets it.
struct ResidentArgs {
    const double *samples;
    const DevFrame *frames;
    const uint32_t *ids;
    const DevPlan *plans;
    const float2 *twpool;
    KParams prm;
    uint8_t *slots;
    DevResult *res;
    atsc_frame_diag *diag;
    UniArgs uni;
    uint32_t *queue;  // RESIDENT_Q_WORDS words: head x at word 32 x (a 128-byte line each), x < 8; exits at word 256
    uint32_t count;
};
// Resident workgroups: as many as the GPU holds at once (resident_grid), each taking frame after frame of the
// class until none is left.  A workgroup slot the dispatcher has to refill stays empty for 1-2.5 us
// (tools/dispatch_probe.hip: 40960 workgroups that live 10 us each with 7008 bytes of LDS keep 18.7 of a CU's 23
// slots busy), a 256-sample frame lives 13 us, and a second launch's workgroups only get on the GPU as these leave:
// its head fills this launch's tail.
// The frame list is cut into eight contiguous shares with a head counter each, and a workgroup starts on the share
// of the XCD it runs on: one counter word takes ~88 returning atomics per microsecond (MI355X_MICROARCH.md,
// "dequeue") where this kernel asks for ~400; when its share is done it goes on to the next XCD's, so the shares'
// different costs even out.  The last workgroup to leave zeroes the counters for the next launch on the stream.
// STATUS: an experiment behind ATSC_RESIDENT=1, not the default.  Measured on the 10.5 M-sample batch (two chains):
// 170 us per step against 106 for one workgroup per frame.  The slots do stay full (4830 frames in flight against
// 4700) and the refill gap shrinks from 2.7 to 1.1 us, but the returning device-scope adds of ~5000 pullers cost
// more than that: a launch ends in a 70-us tail in which the last few hundred frames live 80 us each, fewer
// resident workgroups are FASTER (16 per CU: 158 us, 22: 176), and a relaxed agent-scope load of the head in front
// of every add (to spare the failing adds) took the whole launch to 480 us with every frame's life doubled --
// the queue traffic slows the frames' own memory operations.  DESIGN.md section 3 has the numbers.
template <int W, int SPL, bool IDW, int FN, bool LEAN_ = (FN != 0)>
__global__ __launch_bounds__(64 * W, 5) void k_compress_resident(const ResidentArgs args)
{
    // The arguments are read where they are used, through the kernel-argument segment's own pointer made opaque once
    // per frame: held in registers across the loop they would take ~60 scalar registers from the frame's code.
    typedef const __attribute__((address_space(4))) ResidentArgs *ArgPtr;
    ArgPtr ka = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ uint32_t q_bid, q_cur, q_tried;
    if (threadIdx.x == 0) {
        q_cur = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;  // XCC_ID
        q_tried = 0;
    }
    for (;;) {
        asm volatile("" : "+s"(ka));
        if (threadIdx.x == 0) {
            const uint32_t count = ka->count, per = (count + 7u) >> 3;
            uint32_t *q = ka->queue;
            uint32_t cur = q_cur, tried = q_tried, b = 0xFFFFFFFFu;
            while (tried < 8) {
                const uint32_t start = cur * per;
                const uint32_t size = start < count ? min(per, count - start) : 0u;
                const uint32_t i = size ? atomicAdd(&q[32 * cur], 1u) : 0xFFFFFFFFu;
                if (i < size) { b = start + i; break; }
                cur = (cur + 1) & 7u;
                ++tried;
            }
            q_cur = cur;
            q_tried = tried;
            q_bid = b;
        }
        __syncthreads();
        const uint32_t bid = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_bid);
        __syncthreads();
        if (bid == 0xFFFFFFFFu) break;
        compress_frame<W, SPL, IDW, FN, LEAN_>(ka->samples, ka->frames, ka->ids, ka->plans, ka->twpool, *(const KParams *)&ka->prm,
                                               ka->slots, ka->res, ka->diag, *(const UniArgs *)&ka->uni, bid);
    }
    if (threadIdx.x == 0) {
        uint32_t *q = ka->queue;
        if (atomicAdd(&q[256], 1u) == gridDim.x - 1) {  // everybody else has made its last request
            for (int x = 0; x < 8; ++x) q[32 * x] = 0;
            q[256] = 0;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Scheduling hint.  A frame costs between a few hundred and tens of thousands of instruction slots
// depending on how far its ladders run, and workgroups start in grid order: when the costliest
// frames start last the GPU drains half empty.  k_order_by_cost rewrites a class's launch order,
// costliest first, from the clocks each frame took in the previous batch of the same plan (slot i
// of a recurring batch is the same series one window later).  Only the order of execution changes:
// every result is written at its frame's own position.
// --------------------------------------------------------------------------------------------
struct ClassSpans {
    uint32_t first[8], count[8];  // spans of the small-frame classes in ids[] order, ascending `first`
    uint32_t n_spans, total;      // total = entries covered (the spans are contiguous from 0)
};
DEVI uint32_t cost_bucket(uint32_t c)  // 4 buckets per octave, 0..63
{
    if (c == 0) return 0;
    const uint32_t msb = 31u - (uint32_t)__clz((int)c);
    const uint32_t sub = msb >= 2 ? (c >> (msb - 2)) & 3u : 0u;
    return min(63u, msb * 4u + sub);
}
DEVI uint32_t span_of(const ClassSpans &sp, uint32_t i)
{
    uint32_t c = 0;
    for (uint32_t k = 1; k < sp.n_spans; ++k)
        if (i >= sp.first[k]) c = k;
    return c;
}
// pass 1: bucket of every entry (fixed here: cost[] may already be rewritten by the next batch while
// pass 2 runs) and the per-span histogram, pre-aggregated in LDS.  hist is zero on entry.
__global__ __launch_bounds__(256) void k_cost_hist(const uint32_t *__restrict__ ids_src,
                                                   const uint32_t *cost, uint8_t *__restrict__ bkt,
                                                   uint32_t *__restrict__ hist, const ClassSpans sp)
{
    __shared__ uint32_t h[8 * 64];
    for (uint32_t k = threadIdx.x; k < 8 * 64; k += 256) h[k] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < sp.total) {
        const uint32_t b = cost_bucket(cost[ids_src[i]]);
        bkt[i] = (uint8_t)b;
        atomicAdd(&h[span_of(sp, i) * 64 + b], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 8 * 64; k += 256)
        if (h[k]) atomicAdd(&hist[k], h[k]);
}
// pass 2: every workgroup derives the bucket bases of its spans from the histogram (costliest bucket
// first), reserves room for its own entries with one atomic per non-empty bucket, and scatters.
// The order inside a bucket is whatever the atomics give: it only affects scheduling.
__global__ __launch_bounds__(256) void k_cost_scatter(const uint32_t *__restrict__ ids_src,
                                                      uint32_t *__restrict__ ids_dst,
                                                      const uint8_t *__restrict__ bkt,
                                                      const uint32_t *__restrict__ hist,
                                                      uint32_t *__restrict__ cursor, const ClassSpans sp)
{
    __shared__ uint32_t base[8 * 64], cnt[8 * 64], res[8 * 64];
    for (uint32_t k = threadIdx.x; k < 8 * 64; k += 256) cnt[k] = 0;
    if (threadIdx.x < sp.n_spans) {
        uint32_t acc = sp.first[threadIdx.x];
        for (int b = 63; b >= 0; --b) {
            base[threadIdx.x * 64 + b] = acc;
            acc += hist[threadIdx.x * 64 + b];
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = 0, rank = 0;
    const bool live = i < sp.total;
    if (live) {
        key = span_of(sp, i) * 64 + bkt[i];
        rank = atomicAdd(&cnt[key], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 8 * 64; k += 256)
        if (cnt[k]) res[k] = atomicAdd(&cursor[k], cnt[k]);
    __syncthreads();
    if (live) ids_dst[base[key] + res[key] + rank] = ids_src[i];
}

// --------------------------------------------------------------------------------------------
// packing: frame records -> contiguous BRO body   (frame/mod.rs:25-33, data.rs:79-85)
// --------------------------------------------------------------------------------------------
constexpr int PACK_CHUNK = 1024;  // frames per scan block

DEVI uint32_t rec_header_len(uint32_t n, uint32_t tag, uint32_t len)
{
    return 1 /* varint(41) */ + vlen(n) + vlen(tag) + vlen(len);
}

__global__ __launch_bounds__(256) void k_pack_scan1(const DevFrame *__restrict__ frames,
                                                    const DevResult *__restrict__ res,
                                                    uint64_t n_frames, uint32_t *__restrict__ local,
                                                    uint64_t *__restrict__ blocksum)
{
    __shared__ uint32_t sh[PACK_CHUNK];
    __shared__ uint32_t ws[8];
    const uint64_t base = (uint64_t)blockIdx.x * PACK_CHUNK;
    for (uint32_t i = threadIdx.x; i < PACK_CHUNK; i += 256) {
        const uint64_t f = base + i;
        uint32_t v = 0;
        if (f < n_frames) {
            const DevResult r = res[f];
            v = rec_header_len(frames[f].n, r.chosen, r.len) + r.len;
        }
        sh[i] = v;
    }
    __syncthreads();
    const uint32_t tot = block_excl_scan<4>(sh, PACK_CHUNK, ws);
    for (uint32_t i = threadIdx.x; i < PACK_CHUNK; i += 256)
        if (base + i < n_frames) local[base + i] = sh[i];
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// single block: exclusive scan of the per-chunk totals (u64), in place; total -> blocksum[nb]
__global__ __launch_bounds__(256) void k_pack_scan2(uint64_t *__restrict__ blocksum, uint32_t nb)
{
    __shared__ uint64_t carry;
    __shared__ uint64_t ws[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? blocksum[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(incl, o);
            if ((threadIdx.x & 63) >= (uint32_t)o) incl += t;
        }
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t add = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) add += ws[w];
        if (i < nb) blocksum[i] = incl - v + add;
        __syncthreads();
        if (threadIdx.x == 255) carry = incl + add;
        __syncthreads();
    }
    if (threadIdx.x == 0) blocksum[nb] = carry;
}

// 8 lanes per frame (8 frames per wavefront, 32 per workgroup): header varints + payload copy in
// 8-byte pieces (slots are 16-byte aligned; the destination is not, gfx950 stores it unaligned); also
// the user-visible side arrays.  FUSED: blocksum[] still holds the per-chunk totals (k_pack_scan2 was
// skipped, at most 64 chunks); every wavefront adds up the totals of the chunks before its own (a
// workgroup never straddles a chunk: PACK_CHUNK is a multiple of 32).
constexpr int EMIT_LANES = 8;
constexpr int EMIT_FRAMES = 256 / EMIT_LANES;  // frames per workgroup
constexpr uint32_t EMIT_BIG_N = 4096;  // frames of the large tier: payloads of kilobytes
template <bool FUSED, bool BIGSEP>
__global__ __launch_bounds__(256) void k_pack_emit(
    const DevFrame *__restrict__ frames, const DevResult *__restrict__ res, uint64_t n_frames,
    const uint32_t *__restrict__ local, const uint64_t *__restrict__ blocksum,
    const uint8_t *__restrict__ slots, uint8_t *__restrict__ body, uint64_t body_cap,
    uint64_t *__restrict__ rec_off, uint8_t *__restrict__ chosen, double *__restrict__ err,
    const uint64_t *__restrict__ chain_in, uint64_t *__restrict__ chain_out)
{
    // chain_in / chain_out (optional): this call's records start at *chain_in in `body` and their end goes to
    // *chain_out -- consecutive calls of one stream lay their records end to end (atsc_compress_frames)
    const uint64_t f = (uint64_t)blockIdx.x * EMIT_FRAMES + (threadIdx.x / EMIT_LANES);
    const uint32_t lane = threadIdx.x & 63, l8 = threadIdx.x & (EMIT_LANES - 1);
    const uint32_t chunk = (uint32_t)(((uint64_t)blockIdx.x * EMIT_FRAMES) / PACK_CHUNK);
    uint64_t base;
    if (FUSED) {
        const uint64_t v = lane < chunk ? blocksum[lane] : 0;
        const uint32_t lo = wave_sum_u32((uint32_t)(v & 0xffffffu));
        const uint32_t hi = wave_sum_u32((uint32_t)(v >> 24));
        base = (uint64_t)lo + ((uint64_t)hi << 24);
    } else {
        base = blocksum[chunk];
    }
    if (chain_in) base += *chain_in;
    if (f >= n_frames) return;
    const DevFrame fr = frames[f];
    const DevResult r = res[f];
    const uint64_t off = base + local[f];
    const uint32_t hl = rec_header_len(fr.n, r.chosen, r.len);
    if (l8 == 0) {
        rec_off[f] = off;
        if (chosen) chosen[f] = (uint8_t)r.chosen;
        if (err) err[f] = r.err;
        if (f == n_frames - 1) {
            rec_off[n_frames] = off + hl + r.len;
            if (chain_out) *chain_out = off + hl + r.len;
        }
    }
    if (off + hl + r.len > body_cap) return;  // caller sized d_body too small; rec_off tells
    uint8_t *dst = body + off;
    if (l8 == 0) {
        uint8_t *p = dst;
        *p++ = 41;  // frame_size: size_of_val sum, always 41 on 64-bit (frame/mod.rs:50-56)
        p += put_varint(p, fr.n);
        p += put_varint(p, r.chosen);
        p += put_varint(p, r.len);
    }
    if (BIGSEP && fr.n > EMIT_BIG_N) return;  // k_pack_emit_big copies this payload with a workgroup
    const uint8_t *src = slots + fr.slot_off;
    uint8_t *pd = dst + hl;
    const uint32_t whole = r.len & ~7u;
    for (uint32_t b = l8 * 8; b < whole; b += 8 * EMIT_LANES) {
        const uint64_t v = *(const uint64_t *)(src + b);
        __builtin_memcpy(pd + b, &v, 8);
    }
    if (whole + l8 < r.len) pd[whole + l8] = src[whole + l8];  // at most 7 trailing bytes
}

// Payloads of large frames: one workgroup per frame, 16 bytes per lane (source 16-byte aligned).
// The record header was written by k_pack_emit; rec_off[f] is the record's offset.
__global__ __launch_bounds__(256) void k_pack_emit_big(const DevFrame *__restrict__ frames,
                                                       const DevResult *__restrict__ res,
                                                       const uint32_t *__restrict__ big_ids,
                                                       const uint64_t *__restrict__ rec_off,
                                                       const uint8_t *__restrict__ slots,
                                                       uint8_t *__restrict__ body, uint64_t body_cap)
{
    const uint32_t f = big_ids[blockIdx.x];
    const DevFrame fr = frames[f];
    const DevResult r = res[f];
    const uint64_t off = rec_off[f];
    const uint32_t hl = rec_header_len(fr.n, r.chosen, r.len);
    if (off + hl + r.len > body_cap) return;
    const uint8_t *src = slots + fr.slot_off;
    uint8_t *pd = body + off + hl;
    const uint32_t whole = r.len & ~15u;
    for (uint32_t b = threadIdx.x * 16; b < whole; b += 256 * 16) {
        const uint4 v = *(const uint4 *)(src + b);
        __builtin_memcpy(pd + b, &v, 16);
    }
    if (whole + threadIdx.x < r.len) pd[whole + threadIdx.x] = src[whole + threadIdx.x];
}


// Few frames (a batch of large frames: 80 records of kilobytes): one launch instead of three.  A workgroup per frame
// adds up the record lengths of the frames in front of its own -- at most PACK_SMALL_MAX of them, 16 bytes each, L2
// resident -- and copies its payload 16 bytes per lane.  Same bytes as k_pack_scan1 + k_pack_emit (+ _big).
constexpr uint32_t PACK_SMALL_MAX = 2048;  // (1280 frames of 8192 samples: one launch of 6 us instead of three, 14 us)
__global__ __launch_bounds__(256) void k_pack_small(const DevFrame *__restrict__ frames, const DevResult *__restrict__ res,
                                                    uint32_t n_frames, const uint8_t *__restrict__ slots,
                                                    uint8_t *__restrict__ body, uint64_t body_cap,
                                                    uint64_t *__restrict__ rec_off, uint8_t *__restrict__ chosen,
                                                    double *__restrict__ err, const uint64_t *__restrict__ chain_in,
                                                    uint64_t *__restrict__ chain_out)
{
    __shared__ uint32_t ws[8];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    uint32_t before = 0;
    for (uint32_t g = tid; g < f; g += 256) {
        const DevResult r = res[g];
        before += rec_header_len(frames[g].n, r.chosen, r.len) + r.len;
    }
    before = wave_sum_u32(before);
    if ((tid & 63) == 0) ws[tid >> 6] = before;
    __syncthreads();
    uint64_t off = (uint64_t)ws[0] + ws[1] + ws[2] + ws[3];  // (a wavefront's share: <= 512 records of < 2.3 MB, the 32-bit partial sums hold)
    if (chain_in) off += *chain_in;
    const DevFrame fr = frames[f];
    const DevResult r = res[f];
    const uint32_t hl = rec_header_len(fr.n, r.chosen, r.len);
    if (tid == 0) {
        rec_off[f] = off;
        if (chosen) chosen[f] = (uint8_t)r.chosen;
        if (err) err[f] = r.err;
        if (f == n_frames - 1) {
            rec_off[n_frames] = off + hl + r.len;
            if (chain_out) *chain_out = off + hl + r.len;
        }
    }
    if (off + hl + r.len > body_cap) return;  // caller sized d_body too small; rec_off tells
    uint8_t *dst = body + off;
    if (tid == 0) {
        uint8_t *p = dst;
        *p++ = 41;
        p += put_varint(p, fr.n);
        p += put_varint(p, r.chosen);
        p += put_varint(p, r.len);
    }
    const uint8_t *src = slots + fr.slot_off;
    uint8_t *pd = dst + hl;
    const uint32_t whole = r.len & ~15u;
    for (uint32_t b = tid * 16; b < whole; b += 256 * 16) {
        const uint4 v = *(const uint4 *)(src + b);
        __builtin_memcpy(pd + b, &v, 16);
    }
    if (whole + tid < r.len) pd[whole + tid] = src[whole + tid];
}

// --------------------------------------------------------------------------------------------
// Test hook (declared in atsc_internal.h, not part of the public ABI): the pop order of the heap replay
// (hp_*) for a given array of norms, so that tests can hold it against the oracle's restatement of
// std::collections::BinaryHeap on arbitrary tie patterns.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_test_heap_order(const float *__restrict__ norms, uint32_t bins, uint32_t k,
                                                        uint32_t *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *h = (uint64_t *)smem;
    for (uint32_t i = threadIdx.x; i < bins; i += 64) h[i] = ((uint64_t)__float_as_uint(norms[i]) << 32) | (uint64_t)i;
    __syncthreads();
    hp_rebuild(h, bins);
    uint32_t len = bins;
    for (uint32_t i = 0; i < k && len > 0; ++i) {
        const uint32_t pos = (uint32_t)(hp_pop(h, len) & 0xffffffffu);
        if (threadIdx.x == 0) order[i] = pos;
    }
}
}  // namespace atsc
extern "C" int atsc_internal_heap_order(const float *norms, uint32_t bins, uint32_t k, uint32_t *order)
{
    if (!norms || !order || bins == 0 || k > bins || bins > 8000) return ATSC_E_INVALID;
    float *d_n = nullptr;
    uint32_t *d_o = nullptr;
    int rc = ATSC_E_HIP;
    const uint32_t lds = bins * 8;
    if (hipMalloc((void **)&d_n, bins * sizeof(float)) != hipSuccess) return ATSC_E_HIP;
    if (hipMalloc((void **)&d_o, (k ? k : 1) * sizeof(uint32_t)) == hipSuccess &&
        hipMemcpy(d_n, norms, bins * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
        (lds <= 48 * 1024 || atsc::ensure_dyn_lds((const void *)atsc::k_test_heap_order, lds) == hipSuccess)) {
        hipLaunchKernelGGL(atsc::k_test_heap_order, dim3(1), dim3(64), lds, nullptr, d_n, bins, k, d_o);
        if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(order, d_o, k * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess)
            rc = ATSC_OK;
    }
    (void)hipFree(d_n);
    (void)hipFree(d_o);
    return rc;
}
namespace atsc {

// --------------------------------------------------------------------------------------------
// OptimizerPlan::clean_data (optimizer/mod.rs:64-71) drops NaN and infinite samples before chunking.  A series
// almost never holds one, and finding out on the host means a pass over all of it at one core's memory speed
// (tens of milliseconds for 84 MB that was not in cache); here the samples are on their way to the GPU anyway:
// the flag is raised on the device, and only a series that does hold such a sample takes the host's cleaning pass.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nonfinite_flag(const double *__restrict__ x, uint64_t n, uint32_t *__restrict__ flag)
{
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t b = (uint64_t)__double_as_longlong(x[i]);
        bad |= ((b & 0x7ff0000000000000ull) == 0x7ff0000000000000ull) ? 1u : 0u;
    }
    if (__ballot(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
hipError_t launch_nonfinite_flag(const double *x, uint64_t n, uint32_t *flag, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 2047) / 2048, 4096);
    hipLaunchKernelGGL(k_nonfinite_flag, dim3(grid), dim3(256), 0, s, x, n, flag);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------
template <int W, int SPL, bool IDW, int FN, bool LEAN_ = (FN != 0)>
static hipError_t launch_class2(uint32_t count, uint32_t lds, const double *samples,
                               const DevFrame *frames, const uint32_t *ids, const DevPlan *plans,
                               const float2 *twpool, const KParams &prm, uint8_t *slots,
                               DevResult *res, atsc_frame_diag *diag, const UniArgs &uni, hipStream_t s,
                               hipEvent_t ev0, hipEvent_t ev1)
{
    if (count == 0) return hipSuccess;
    auto kern = k_compress<W, SPL, IDW, FN, LEAN_>;
    static const uint32_t lds_pad = [] {  // ATSC_DEBUG_LDS_PAD: occupancy experiments only, read once
        const char *pad = getenv("ATSC_DEBUG_LDS_PAD");
        return pad ? (uint32_t)atoi(pad) : 0u;
    }();
    lds += lds_pad;
    if (lds > 48 * 1024) {
        hipError_t e = ensure_dyn_lds((const void *)kern, lds);
        if (e != hipSuccess) return e;
    }
    // ev0 / ev1 (optional) take the start / end timestamps of this dispatch itself: no separate
    // event packets, hence no bubbles around the kernel when it is being timed
    if constexpr (W == 1 && FN == 256) {
        if (uni.queue && uni.q_grid) {
            ResidentArgs ra;
            ra.samples = samples; ra.frames = frames; ra.ids = ids; ra.plans = plans; ra.twpool = twpool; ra.prm = prm;
            ra.slots = slots; ra.res = res; ra.diag = diag; ra.uni = uni; ra.queue = uni.queue;
            ra.count = count;
            hipExtLaunchKernelGGL((k_compress_resident<W, SPL, IDW, FN, LEAN_>), dim3(uni.q_grid), dim3(64 * W), lds, s, ev0, ev1, 0, ra);
            return hipGetLastError();
        }
    }
    hipExtLaunchKernelGGL(kern, dim3(count), dim3(64 * W), lds, s, ev0, ev1, 0, samples, frames, ids,
                          plans, twpool, prm, slots, res, diag, uni);
    return hipGetLastError();
}

uint32_t resident_grid(int cls, uint32_t n, uint32_t lds)
{
    if (cls != 1 || n != 256) return 0;
    static int per_cu = 0, cus = 0;
    if (!per_cu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_compress_resident<1, 5, false, 256>, 64, lds) != hipSuccess || nb <= 0)
            return 0;
        if (const char *e = getenv("ATSC_RESIDENT_PER_CU")) nb = atoi(e);
        per_cu = nb;
        cus = pr.multiProcessorCount;
    }
    return (uint32_t)(per_cu * cus);
}

template <int W, int SPL>
static hipError_t launch_class(uint32_t count, uint32_t lds, const double *samples,
                               const DevFrame *frames, const uint32_t *ids, const DevPlan *plans,
                               const float2 *twpool, const KParams &prm, uint8_t *slots,
                               DevResult *res, atsc_frame_diag *diag, const UniArgs &uni, hipStream_t s,
                               hipEvent_t ev0, hipEvent_t ev1)
{
    if (prm.mode == ATSC_IDW)
        return launch_class2<W, SPL, true, 0>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    // uniform launches of the power-of-two frame lengths the reference chunker emits (256 is also
    // BASELINE's framing) take the instantiation with the frame geometry folded in
    const bool lean_ok = diag == nullptr && (prm.debug_stop & 0xffffff) == 0 && !prm.trial && prm.trial_res == nullptr &&
                         prm.mode == ATSC_AUTO && prm.bounded && 0.0 <= prm.max_err;
    if (uni.enabled && lean_ok) {
        if constexpr (W == 1 && SPL == 5) {
            if (uni.n == 256) return launch_class2<1, 5, false, 256>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
            if (uni.n == 128) return launch_class2<1, 5, false, 128>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
        }
        if constexpr (W == 1 && SPL == 9) {
            if (uni.n == 512) return launch_class2<1, 9, false, 512>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
        }
        if constexpr (W == 4 && SPL == 5) {
            if (uni.n == 1024) return launch_class2<4, 5, false, 1024>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
        }
        if constexpr (W == 8 && SPL == 5) {
            if (uni.n == 2048) return launch_class2<8, 5, false, 2048>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
        }
        if constexpr (W == 16 && SPL == 5) {
            if (uni.n == 4096) return launch_class2<16, 5, false, 4096>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
        }
    }
    // any other length under the auto selector: the table-driven kernel without the forced-codec / trial /
    // diagnostics paths
    if (lean_ok)
        return launch_class2<W, SPL, false, 0, true>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    return launch_class2<W, SPL, false, 0>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
}

hipError_t launch_compress_class(int cls, uint32_t count, uint32_t lds, const double *samples,
                                 const DevFrame *frames, const uint32_t *ids, const DevPlan *plans,
                                 const float2 *twpool, const KParams &prm, uint8_t *slots,
                                 DevResult *res, atsc_frame_diag *diag, const UniArgs &uni, hipStream_t s,
                                 hipEvent_t ev0, hipEvent_t ev1)
{
    switch (cls) {
    case 0: return launch_class<1, 2>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    case 1: return launch_class<1, 5>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    case 2: return launch_class<1, 9>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    case 3: return launch_class<4, 5>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    // (class 4, L <= 2304, ran as <4, 9> in round 1: 72 KB of LDS per frame leave two workgroups on a CU, and two
    // workgroups of 4 wavefronts are 2 waves per SIMD; 8 wavefronts with 5 samples per lane double that: 2048-sample
    // frames 33 -> 36 Gsamples/s.  The same step for class 3 -- <8, 3> instead of <4, 5> -- halves its rate.)
    case 4: return launch_class<8, 5>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    case 5: return launch_class<16, 5>(count, lds, samples, frames, ids, plans, twpool, prm, slots, res, diag, uni, s, ev0, ev1);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pack(const DevFrame *frames, const DevResult *res, uint64_t n_frames,
                       uint32_t *local, uint64_t *blocksum, const uint8_t *slots, uint8_t *body,
                       uint64_t body_cap, uint64_t *rec_off, uint8_t *chosen, double *err,
                       const uint32_t *big_ids, uint32_t n_big, hipStream_t s, uint64_t *chain)
{
    const uint64_t *chain_in = chain;
    uint64_t *chain_out = chain ? chain + 1 : nullptr;
    if (n_frames <= 256 || (n_big * 4 >= n_frames && n_frames <= PACK_SMALL_MAX)) {
        // few records (or mostly large frames' kilobyte payloads): one launch
        hipLaunchKernelGGL(k_pack_small, dim3((uint32_t)n_frames), dim3(256), 0, s, frames, res, (uint32_t)n_frames, slots,
                           body, body_cap, rec_off, chosen, err, chain_in, chain_out);
        return hipGetLastError();
    }
    const uint32_t nb = (uint32_t)((n_frames + PACK_CHUNK - 1) / PACK_CHUNK);
    hipLaunchKernelGGL(k_pack_scan1, dim3(nb), dim3(256), 0, s, frames, res, n_frames, local,
                       blocksum);
    const dim3 eg((uint32_t)((n_frames + EMIT_FRAMES - 1) / EMIT_FRAMES));
    const bool big = n_big != 0;
    if (nb <= 64) {
        if (big)
            hipLaunchKernelGGL((k_pack_emit<true, true>), eg, dim3(256), 0, s, frames, res, n_frames, local, blocksum,
                               slots, body, body_cap, rec_off, chosen, err, chain_in, chain_out);
        else
            hipLaunchKernelGGL((k_pack_emit<true, false>), eg, dim3(256), 0, s, frames, res, n_frames, local, blocksum,
                               slots, body, body_cap, rec_off, chosen, err, chain_in, chain_out);
    } else {
        hipLaunchKernelGGL(k_pack_scan2, dim3(1), dim3(256), 0, s, blocksum, nb);
        if (big)
            hipLaunchKernelGGL((k_pack_emit<false, true>), eg, dim3(256), 0, s, frames, res, n_frames, local, blocksum,
                               slots, body, body_cap, rec_off, chosen, err, chain_in, chain_out);
        else
            hipLaunchKernelGGL((k_pack_emit<false, false>), eg, dim3(256), 0, s, frames, res, n_frames, local, blocksum,
                               slots, body, body_cap, rec_off, chosen, err, chain_in, chain_out);
    }
    if (big)
        hipLaunchKernelGGL(k_pack_emit_big, dim3(n_big), dim3(256), 0, s, frames, res, big_ids, rec_off, slots,
                           body, body_cap);
    return hipGetLastError();
}

hipError_t launch_order_by_cost(const uint32_t *ids_src, uint32_t *ids_dst, const uint32_t *cost, uint8_t *bkt,
                                uint32_t *hist_cursor /* 2 * 8 * 64 u32 */, const uint32_t *class_first,
                                const uint32_t *class_count, int n_classes, hipStream_t s)
{
    ClassSpans sp;
    memset(&sp, 0, sizeof(sp));
    for (int c = 0; c < n_classes && sp.n_spans < 8; ++c)
        if (class_count[c]) {
            if (class_first[c] != sp.total) return hipErrorInvalidValue;  // spans must be contiguous from 0
            sp.first[sp.n_spans] = class_first[c];
            sp.count[sp.n_spans] = class_count[c];
            sp.total += class_count[c];
            ++sp.n_spans;
        }
    if (!sp.total) return hipSuccess;
    hipError_t e = hipMemsetAsync(hist_cursor, 0, 2 * 8 * 64 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const dim3 grid((sp.total + 255) / 256);
    hipLaunchKernelGGL(k_cost_hist, grid, dim3(256), 0, s, ids_src, cost, bkt, hist_cursor, sp);
    hipLaunchKernelGGL(k_cost_scatter, grid, dim3(256), 0, s, ids_src, ids_dst, bkt, hist_cursor,
                       hist_cursor + 8 * 64, sp);
    return hipGetLastError();
}

}  // namespace atsc

#ifdef ATSC_STAMPS
extern "C" __attribute__((visibility("default"))) int atsc_dev_phase_read(unsigned long long *out16, int reset)
{
    static unsigned long long rows[64 * 16];
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(atsc::g_phase_cyc), sizeof(rows)) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) {
        out16[i] = 0;
        for (int r = 0; r < 64; ++r) out16[i] += rows[r * 16 + i];
    }
    if (reset) {
        for (auto &v : rows) v = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(atsc::g_phase_cyc), rows, sizeof(rows)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int atsc_dev_span_read(unsigned long long *out, unsigned n_frames)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(atsc::g_frame_span), sizeof(unsigned long long) * 3 * n_frames) == hipSuccess ? 0 : -1;
}
// the stamps of scratch set `set` (pipelined calls: set = turn % (2 * chains))
extern "C" __attribute__((visibility("default"))) int atsc_dev_span_read_set(unsigned long long *out, unsigned n_frames, unsigned set)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(atsc::g_frame_span), sizeof(unsigned long long) * 3 * n_frames,
                               sizeof(unsigned long long) * 3 * 65536 * (set & 3u)) == hipSuccess ? 0 : -1;
}
#endif
