// atsc_large.hip -- frames of 4097 .. 131072 samples (the sizes OptimizerPlan::get_chunks_sizes emits
// for long series, atsc/src/optimizer/mod.rs:78-98; transform lengths up to 139968 = 2^6 3^7).
//
// Same algorithms and reference semantics as k_compress (atsc_kernels.hip); what changes is where the
// data lives and how the FFT error ladder reconstructs:
//   * one workgroup of 16 wavefronts per frame; samples are read from the caller's buffer, everything
//     else (FFT ping-pong buffers, admitted spectrum, spline tangents, RLE records) sits in a
//     per-workgroup workspace in HBM (L2 resident: ~3 MB for a 131072-sample frame);
//   * each ladder trip admits up to n/200 new bins, so the incremental direct sum of the small-frame
//     kernel (O(new bins * L)) loses to one inverse FFT of the Hermitian spectrum per trip:
//     real-output trick, one complex FFT of length L/2 (fft.rs:338-344);
//   * bin admission order: radix-select of the kcap largest f32 norms, then one LDS sort of
//     (norm, position) keys (fft.rs:231-257).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "atsc_device.h"

namespace atsc {

// ATSC_DEBUG_STOP=-1: workgroup 0 prints the constant 100 MHz clock at the phase boundaries of k_compress_large
// (tools/large_stamp_probe.py turns the lines into a phase table); everything runs as usual otherwise
#define LT_STAMP(name)                                                                                   \
    do {                                                                                                 \
        if (prm.debug_stop == -1 && blockIdx.x == 0 && threadIdx.x == 0)                                 \
            printf("LTSTAMP part%d %s %llu\n", PART, name, (unsigned long long)wall_clock64());           \
    } while (0)

constexpr int LW = 16;          // wavefronts per workgroup (8 were tried: no register spills, but 25 % slower)
constexpr int LT = 64 * LW;     // threads

DEVI float2 cmulc(float2 v, float2 w)  // v * (w.x - i w.y)
{
    return make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
}
DEVI float2 cmulp(float2 v, float2 w)  // v * (w.x + i w.y)
{
    return make_float2(v.x * w.x - v.y * w.y, v.y * w.x + v.x * w.y);
}

// Stockham stages over global buffers (same butterflies as fft_forward in atsc_kernels.hip).  One
// workgroup walks a frame, so a stage is bound by memory latency, not bandwidth: the radix switch
// sits outside the butterfly loop and every thread keeps U butterflies in flight (all their loads are
// issued before the first result is needed).
template <int R>
DEVI void butterfly_g(const float2 (&a)[R], const float2 (&w)[R], float2 (&y)[R])
{
    if (R == 4) {
        const float2 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3 % R];
        const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y);
        const float2 t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
        const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
        const float2 d = make_float2(a1.x - a3.x, a1.y - a3.y);
        const float2 t3 = make_float2(d.y, -d.x);
        y[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
        y[1] = cmulc(make_float2(t1.x + t3.x, t1.y + t3.y), w[1]);
        y[2 % R] = cmulc(make_float2(t0.x - t2.x, t0.y - t2.y), w[2 % R]);
        y[3 % R] = cmulc(make_float2(t1.x - t3.x, t1.y - t3.y), w[3 % R]);
    } else if (R == 2) {
        const float2 a0 = a[0], a1 = a[1];
        y[0] = make_float2(a0.x + a1.x, a0.y + a1.y);
        y[1] = cmulc(make_float2(a0.x - a1.x, a0.y - a1.y), w[1]);
    } else {
        const float2 a0 = a[0], a1 = a[1], a2 = a[2 % R];
        const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
        const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
        const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
        const float h = 0.8660254037844386f;
        const float2 t3 = make_float2(h * d.y, -h * d.x);
        y[0] = make_float2(a0.x + t1.x, a0.y + t1.y);
        y[1] = cmulc(make_float2(t2.x + t3.x, t2.y + t3.y), w[1]);
        y[2 % R] = cmulc(make_float2(t2.x - t3.x, t2.y - t3.y), w[2 % R]);
    }
}
// One butterfly per thread and iteration, 8-byte accesses (odd strides: transform lengths 3^k).
template <int R>
DEVI void fft_stage_g(const float2 *__restrict__ X, float2 *__restrict__ Y, const float2 *__restrict__ tw,
                      uint32_t nb, uint32_t st, uint32_t sm, uint32_t sc, uint32_t magic)
{
    const uint32_t mg = (st == 1) ? 0u : magic;  // __umulhi(t, 0) + t below: no select in the loop
    for (uint32_t t = threadIdx.x; t < nb; t += LT) {
        const uint32_t p = (st == 1) ? t : __umulhi(t, mg);
        const uint32_t q = t - p * st;
        const uint32_t ib = q + st * p, ob = q + st * (R * p), tb = p * st * sc;
        float2 a[R], w[R], y[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = X[ib + j * sm];
        w[0] = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int j = 1; j < R; ++j) w[j] = tw[j * tb];
        butterfly_g<R>(a, w, y);
#pragma unroll
        for (int k = 0; k < R; ++k) Y[ob + k * st] = y[k];
    }
}
// Two adjacent butterflies per thread and iteration, 16-byte accesses.  Even stride: butterflies
// (p, q) and (p, q + 1), q even, read and write neighbouring points and share their twiddles.
// Stride 1 (first stage): butterflies p and p + 1 read neighbouring points and each writes R
// consecutive ones.  nb and sm are even in both cases.
template <int R, bool FIRST>
DEVI void fft_stage_g2(const float2 *__restrict__ X, float2 *__restrict__ Y, const float2 *__restrict__ tw,
                       uint32_t nb, uint32_t st, uint32_t sm, uint32_t sc, uint32_t magic)
{
    // FIRST (st == 1) is a template parameter: as a run-time select inside the loop it turned every
    // twiddle load into a branch with its own s_waitcnt, i.e. one memory round trip after the other.
    // (Requesting the next pair's operands before storing the current results was tried as well: no
    // gain, and the extra live registers spill at 1024 threads per workgroup.  A 70 K-point stage is
    // about 33 K wavefront-instructions on the CU's four SIMDs: half of the stage time is issue.)
    for (uint32_t t = 2 * threadIdx.x; t < nb; t += 2 * LT) {
        const uint32_t p = FIRST ? t : __umulhi(t, magic);
        const uint32_t q = t - p * st;
        const uint32_t ib = q + st * p, ob = q + st * (R * p), tb = p * st * sc;
        float2 a0[R], a1[R], w0[R], w1[R], y0[R], y1[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float4 v = *(const float4 *)(X + ib + j * sm);
            a0[j] = make_float2(v.x, v.y);
            a1[j] = make_float2(v.z, v.w);
        }
        w0[0] = w1[0] = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int j = 1; j < R; ++j) {
            w0[j] = tw[j * tb];
            w1[j] = FIRST ? tw[j * (tb + sc)] : w0[j];
        }
        butterfly_g<R>(a0, w0, y0);
        butterfly_g<R>(a1, w1, y1);
        if (FIRST) {
            float2 yy[2 * R];
#pragma unroll
            for (int k = 0; k < R; ++k) { yy[k] = y0[k]; yy[R + k] = y1[k]; }
#pragma unroll
            for (int k = 0; k < R; ++k)  // 2 R consecutive points from ob = R p (even)
                *(float4 *)(Y + ob + 2 * k) = make_float4(yy[2 * k].x, yy[2 * k].y, yy[2 * k + 1].x, yy[2 * k + 1].y);
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k)
                *(float4 *)(Y + ob + k * st) = make_float4(y0[k].x, y0[k].y, y1[k].x, y1[k].y);
        }
    }
}
DEVI float2 *fft_forward_g(const DevPlan &P, float2 *X, float2 *Y, const float2 *tw)
{
    const uint32_t M = P.M, sc = P.sc;
    uint32_t ncur = M, st = 1;
    for (uint32_t s = 0; s < P.nstages; ++s) {
        const uint32_t r = P.radix[s];
        const uint32_t m = ncur / r;
        const uint32_t nb = M / r;
        const uint32_t magic = P.stmagic[s];
        const uint32_t sm = st * m;
        // pairs need neighbouring butterflies to be neighbours in memory and every 16-byte access
        // aligned: an even stride (then nb and sm are even too), or the first stage with an even
        // number of butterflies; transform lengths 3^k run the single-butterfly form
        const bool pairs = (st % 2 == 0) || (st == 1 && nb % 2 == 0 && sm % 2 == 0);
        if (pairs) {
            if (st == 1) {
                if (r == 4) fft_stage_g2<4, true>(X, Y, tw, nb, st, sm, sc, magic);
                else if (r == 2) fft_stage_g2<2, true>(X, Y, tw, nb, st, sm, sc, magic);
                else fft_stage_g2<3, true>(X, Y, tw, nb, st, sm, sc, magic);
            } else {
                if (r == 4) fft_stage_g2<4, false>(X, Y, tw, nb, st, sm, sc, magic);
                else if (r == 2) fft_stage_g2<2, false>(X, Y, tw, nb, st, sm, sc, magic);
                else fft_stage_g2<3, false>(X, Y, tw, nb, st, sm, sc, magic);
            }
        } else {
            if (r == 4) fft_stage_g<4>(X, Y, tw, nb, st, sm, sc, magic);
            else if (r == 2) fft_stage_g<2>(X, Y, tw, nb, st, sm, sc, magic);
            else fft_stage_g<3>(X, Y, tw, nb, st, sm, sc, magic);
        }
        __syncthreads();
        float2 *tmp = X; X = Y; Y = tmp;
        ncur = m;
        st *= r;
    }
    return X;
}

// --------------------------------------------------------------------------------------------
// The same transform in two LDS-tiled passes, M = M1 * M2 (n = M2 n1 + n2, k = k1 + M1 k2):
//   W_M^{nk} = W_M1^{n1 k1} . W_M^{n2 k1} . W_M2^{n2 k2}
//   pass 1  FB columns n2 at a time: x[M2 n1 + n2] -> LDS, FB transforms of length M1 along n1,
//           times W_M^{n2 k1}, out to a[k1 M2 + n2]
//   pass 2  FB rows k1 at a time: a[k1 M2 + .] -> LDS, FB transforms of length M2 along n2, out to
//           X[k1 + M1 k2]
// Global memory is touched in runs of FB * 8 = 128 bytes or whole rows, twice in and twice out,
// instead of once per radix stage with 8-byte scatters; the butterflies run on LDS.
// --------------------------------------------------------------------------------------------
constexpr uint32_t FB = 16;            // sequences per LDS tile
constexpr uint32_t F4_MAX = 480;       // longest sub-transform the tile buffers hold
constexpr uint32_t F4_TILE = FB * (F4_MAX + 1);  // float2 per tile buffer

// Stockham over `nseq` sequences of length N held in LDS, element (i, c) at i * si + c * sq.
// CF: consecutive work items walk the sequences first (si = FB, sq = 1), else the butterflies first.
template <bool CF, uint32_t FBX = FB>
DEVI float2 *lds_fft(float2 *T, float2 *U, const float2 *wN, uint32_t N, uint32_t nseq, uint32_t si, uint32_t sq,
                     uint32_t nt = LT)
{
    uint32_t ncur = N, st = 1;
    while (ncur > 1) {
        // radix 9 first: two radix-3 levels in registers halve the LDS round trips, twiddle loads,
        // index arithmetic and barriers of the 3^k part (transform lengths are 2^a 3^b, b up to 7)
        const uint32_t r = (ncur % 9 == 0) ? 9u : (ncur % 4 == 0) ? 4u : (ncur % 2 == 0) ? 2u : 3u;
        const uint32_t m = ncur / r, nbf = N / r, sm = st * m;
        const uint32_t mg_st = st > 1 ? (uint32_t)(0x100000000ull / st) + 1u : 0u;
        const uint32_t mg_nb = nbf > 1 ? (uint32_t)(0x100000000ull / nbf) + 1u : 0u;
        const uint32_t total = CF ? nbf * FBX : nbf * nseq;
        for (uint32_t w = threadIdx.x; w < total; w += nt) {
            uint32_t b, c;
            if (CF) { c = w & (FBX - 1); b = w / FBX; if (c >= nseq) continue; }
            else { c = nbf > 1 ? __umulhi(w, mg_nb) : w; b = w - c * nbf; }
            const uint32_t p = st > 1 ? __umulhi(b, mg_st) : b;
            const uint32_t q = b - p * st;
            const float2 *x = T + (q + st * p) * si + c * sq;
            float2 *y = U + (q + st * (r * p)) * si + c * sq;
            const uint32_t e = p * st;  // W_N^{e k}
            if (r == 9) {
                // j = 3 j1 + j2, k = k1 + 3 k2:  W9^{jk} = W3^{j1 k1} W9^{j2 k1} W3^{j2 k2}
                float2 a[9];
#pragma unroll
                for (uint32_t j = 0; j < 9; ++j) a[j] = x[j * sm * si];
                auto dft3 = [](float2 &a0, float2 &a1, float2 &a2) {
                    const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
                    const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
                    const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
                    const float h = 0.8660254037844386f;
                    const float2 t3 = make_float2(h * d.y, -h * d.x);
                    a0 = make_float2(a0.x + t1.x, a0.y + t1.y);
                    a1 = make_float2(t2.x + t3.x, t2.y + t3.y);
                    a2 = make_float2(t2.x - t3.x, t2.y - t3.y);
                };
                dft3(a[0], a[3], a[6]);  // over j1, for j2 = 0, 1, 2: a[3 k1 + j2]
                dft3(a[1], a[4], a[7]);
                dft3(a[2], a[5], a[8]);
                const float2 w1 = make_float2(0.766044443118978f, 0.642787609686539f);   // (cos, sin) 40 deg
                const float2 w2 = make_float2(0.17364817766693f, 0.984807753012208f);    // 80 deg
                const float2 w4 = make_float2(-0.939692620785908f, 0.342020143325669f);  // 160 deg
                a[4] = cmulc(a[4], w1);  // j2 = 1, k1 = 1
                a[7] = cmulc(a[7], w2);  // j2 = 1, k1 = 2
                a[5] = cmulc(a[5], w2);  // j2 = 2, k1 = 1
                a[8] = cmulc(a[8], w4);  // j2 = 2, k1 = 2
                dft3(a[0], a[1], a[2]);  // over j2, for k1 = 0: outputs k = 0, 3, 6
                dft3(a[3], a[4], a[5]);  // k1 = 1: k = 1, 4, 7
                dft3(a[6], a[7], a[8]);  // k1 = 2: k = 2, 5, 8
                y[0] = a[0];
#pragma unroll
                for (uint32_t k1 = 0; k1 < 3; ++k1)
#pragma unroll
                    for (uint32_t k2 = 0; k2 < 3; ++k2) {
                        const uint32_t k = k1 + 3 * k2;
                        if (k) y[k * st * si] = cmulc(a[3 * k1 + k2], wN[k * e]);
                    }
            } else if (r == 4) {
                const float2 a0 = x[0], a1 = x[sm * si], a2 = x[2 * sm * si], a3 = x[3 * sm * si];
                const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y);
                const float2 t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
                const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
                const float2 d = make_float2(a1.x - a3.x, a1.y - a3.y);
                const float2 t3 = make_float2(d.y, -d.x);
                y[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
                y[st * si] = cmulc(make_float2(t1.x + t3.x, t1.y + t3.y), wN[e]);
                y[2 * st * si] = cmulc(make_float2(t0.x - t2.x, t0.y - t2.y), wN[2 * e]);
                y[3 * st * si] = cmulc(make_float2(t1.x - t3.x, t1.y - t3.y), wN[3 * e]);
            } else if (r == 2) {
                const float2 a0 = x[0], a1 = x[sm * si];
                y[0] = make_float2(a0.x + a1.x, a0.y + a1.y);
                y[st * si] = cmulc(make_float2(a0.x - a1.x, a0.y - a1.y), wN[e]);
            } else {
                const float2 a0 = x[0], a1 = x[sm * si], a2 = x[2 * sm * si];
                const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
                const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
                const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
                const float h = 0.8660254037844386f;
                const float2 t3 = make_float2(h * d.y, -h * d.x);
                y[0] = make_float2(a0.x + t1.x, a0.y + t1.y);
                y[st * si] = cmulc(make_float2(t2.x + t3.x, t2.y + t3.y), wN[e]);
                y[2 * st * si] = cmulc(make_float2(t2.x - t3.x, t2.y - t3.y), wN[2 * e]);
            }
        }
        __syncthreads();
        float2 *tmp = T; T = U; U = tmp;
        ncur = m;
        st *= r;
    }
    return T;
}

// Forward transform of X (length P.M) through the scratch buffer Y; the result lands back in X.
// `lds` = 2 * F4_TILE + 2 * F4_MAX float2 of LDS that nothing else uses meanwhile.
DEVI float2 *fft_tiled_g(const DevPlan &P, float2 *X, float2 *Y, const float2 *tw, float2 *lds)
{
    const uint32_t M1 = P.f4_m1, M2 = P.f4_m2, sc = P.sc;  // sc = L / M
    float2 *T = lds, *U = lds + F4_TILE, *w1 = lds + 2 * F4_TILE, *w2 = w1 + F4_MAX;
    for (uint32_t e = threadIdx.x; e < M1; e += LT) w1[e] = tw[e * (M2 * sc)];  // W_M1^e = W_L^{e L / M1}
    for (uint32_t e = threadIdx.x; e < M2; e += LT) w2[e] = tw[e * (M1 * sc)];
    __syncthreads();
    for (uint32_t c0 = 0; c0 < M2; c0 += FB) {  // ---- pass 1: columns
        const uint32_t nseq = min(FB, M2 - c0);
        for (uint32_t w = threadIdx.x; w < M1 * FB; w += LT) {
            const uint32_t c = w & (FB - 1), n1 = w >> 4;
            if (c < nseq) T[n1 * FB + c] = X[M2 * n1 + c0 + c];
        }
        __syncthreads();
        const float2 *R = lds_fft<true>(T, U, w1, M1, nseq, FB, 1);
        for (uint32_t w = threadIdx.x; w < M1 * FB; w += LT) {
            const uint32_t c = w & (FB - 1), k1 = w >> 4;
            if (c < nseq) {
                const uint32_t n2 = c0 + c;
                Y[k1 * M2 + n2] = cmulc(R[k1 * FB + c], tw[n2 * k1 * sc]);  // n2 k1 < M: no wrap
            }
        }
        __syncthreads();
    }
    const uint32_t ld = M2 + 1;  // odd leading dimension keeps the strided tile accesses off one bank
    const uint32_t mg_m2 = (uint32_t)(0x100000000ull / M2) + 1u;
    for (uint32_t r0 = 0; r0 < M1; r0 += FB) {  // ---- pass 2: rows
        const uint32_t nseq = min(FB, M1 - r0);
        for (uint32_t w = threadIdx.x; w < nseq * M2; w += LT) {
            const uint32_t r = __umulhi(w, mg_m2), n2 = w - r * M2;
            T[r * ld + n2] = Y[(r0 + r) * M2 + n2];
        }
        __syncthreads();
        const float2 *R = lds_fft<false>(T, U, w2, M2, nseq, 1, ld);
        for (uint32_t w = threadIdx.x; w < M2 * FB; w += LT) {
            const uint32_t r = w & (FB - 1), k2 = w >> 4;
            if (r < nseq) X[(r0 + r) + M1 * k2] = R[r * ld + k2];
        }
        __syncthreads();
    }
    return X;
}
// the transform the large tier uses: tiled when the plan has a split, stage by stage otherwise
DEVI float2 *fft_large(const DevPlan &P, float2 *X, float2 *Y, const float2 *tw, float2 *lds, bool tiled)
{
    if (tiled && P.f4_m1) return fft_tiled_g(P, X, Y, tw, lds);
    return fft_forward_g(P, X, Y, tw);
}

// workspace carve (bytes) -- the host uses the same function to size a slot
__host__ __device__ inline uint64_t lw_align(uint64_t v) { return (v + 255) & ~255ull; }
struct LargeWs {
    uint64_t o_a, o_b, o_c, o_x, o_nb, o_sel, o_mm, o_aux, o_rec, o_hp, o_tab, o_rps, o_rph, o_spos, o_cnt, o_front, o_part, o_tst, o_rbm, bytes;
};
__host__ __device__ inline LargeWs large_ws_layout(uint32_t n, uint32_t L, uint32_t kcap)
{
    LargeWs w;
    uint64_t o = 0;
    const uint64_t cplx = 8ull * (L + 8);
    w.o_a = o; o += lw_align(cplx);                 // FFT ping
    w.o_b = o; o += lw_align(cplx);                 // FFT pong
    w.o_c = o; o += lw_align(cplx);                 // inverse-transform partner (keeps the spectrum intact)
    w.o_x = o; o += lw_align(8ull * (L / 2 + 8));   // admitted spectrum Xsel[0..L/2] (zero elsewhere)
    w.o_nb = o; o += lw_align(4ull * (L + 8));      // norm bits per bin
    w.o_sel = o; o += lw_align(12ull * (kcap + 8)); // admitted bins in admission order
    w.o_mm = o; o += lw_align(16ull * (n / 2 + 8)); // spline tangents per segment
    w.o_aux = o; o += lw_align(4ull * ((n > 65536 ? n : 65536) + 8));  // scans; u16-position owners in the ladder
    w.o_rec = o; o += lw_align(8ull * (n + 8));     // RLE run records (start << 32 | end)
    w.o_hp = o; o += lw_align(4ull * (n + 8));
    w.o_tab = o; o += lw_align(8ull * (n + 8));     // RLE hash table, 2n slots
    w.o_rps = o; o += lw_align(4ull * (n + 8));
    w.o_rph = o; o += lw_align(4ull * (n + 8));
    w.o_spos = o; o += lw_align(4ull * (16384 + 8));  // admission order (bin positions) once sorted: frees the LDS
    w.o_cnt = o; o += lw_align(16);                   // pre-pass: number of ZERO bins (rare: few atomics)
    w.o_front = o; o += lw_align(256);                // k_compress_large<1> -> <2>: TripState
    w.o_part = o; o += lw_align(4096);                // first polynomial trip: MAPE sums of the 4096-sample chunks (k_large_poly1);
                                                      // fast path: [512] sums and [1536] run counts of the 1024-sample pieces
    w.o_tst = o; o += lw_align(32 * 32);              // k_large_cols243: statistics of each column tile (TileStats)
    w.o_rbm = o; o += lw_align(n / 8 + 64);           // fast path: one bit per sample, set where a run starts (poly1_piece)
    w.bytes = o;
    return w;
}
// A slot ends with LARGE_WS_TAIL bytes that belong to the LAUNCH, not to the slot's frame (so their place does not depend on
// the frame's layout): the list of the frames the grid path leaves to the general kernel -- entry i in the tail of slot i,
// the count behind entry 0 -- see fb_append / k_compress_large.
constexpr uint64_t LARGE_WS_TAIL = 256;
constexpr uint32_t FB_GRID_MAX = 256;  // workgroups of the general kernel's launch behind the grid path (one per CU)
uint64_t large_ws_bytes(uint32_t n, uint32_t L, uint32_t kcap) { return large_ws_layout(n, L, kcap).bytes + LARGE_WS_TAIL; }
DEVI uint32_t *fb_entry(unsigned char *ws_base, uint64_t ws_stride, uint32_t i)
{
    return (uint32_t *)(ws_base + (uint64_t)i * ws_stride + (ws_stride - LARGE_WS_TAIL));
}
DEVI uint32_t *fb_count(unsigned char *ws_base, uint64_t ws_stride) { return fb_entry(ws_base, ws_stride, 0) + 1; }
DEVI void fb_append(unsigned char *ws_base, uint64_t ws_stride, uint32_t slot)  // one thread of the frame's workgroup
{
    *fb_entry(ws_base, ws_stride, atomicAdd(fb_count(ws_base, ws_stride), 1u)) = slot;
}

// exclusive scan over a (global) u32 array by one workgroup; returns the total
DEVI uint32_t lscan(uint32_t *arr, uint32_t count, uint32_t *wsum)
{
    return block_excl_scan<LW>(arr, count, wsum);
}

// --------------------------------------------------------------------------------------------
// Inverse transform straight from the sparse list of admitted bins.
//
// A ladder trip (and the decoder) transforms a spectrum that holds K non-zero bins out of L/2 + 1
// (K starts at n/100).  Run as a dense transform that costs two passes over the whole packed spectrum
// through the workspace plus the pass that builds it; here nothing dense ever leaves the CU.  With
// M = Mf * Md (DevPlan::sp_mf, sp_md), input index k = ka + Mf kb and output index j = Md ja + jb:
//
//   F[Md ja + jb] = sum_ka W_Mf^{ja ka} * W_M^{jb ka} * G[ka][jb],   G[ka][jb] = sum_kb Z[ka + Mf kb] W_Md^{jb kb}
//
// G is a direct sum over the list (each admitted bin gives at most two points of the packed spectrum
// Z, so K * 2 * Md multiply-adds per trip); what remains is a batch of length-Mf transforms in LDS,
// SPB output columns jb at a time, whose results feed the evaluation (error sum or decoded samples)
// directly: 2 * SPB consecutive samples per run.  The list is bucketed by ka once per trip (LDS
// counters, scan, scatter, then each bucket put in ascending (kb, kind) order by one thread so that the
// f32 sums do not depend on the order the atomics happened to serve).
// --------------------------------------------------------------------------------------------
constexpr uint32_t SPB = 8;          // output columns per LDS tile
constexpr uint32_t SP_MF_MAX = 992;  // longest LDS sub-transform (host: DevPlan::sp_mf)
constexpr uint32_t SP_MD_MAX = 256;
constexpr uint32_t RLE_LDS_RUNS = 6144;  // runs k_decompress_large<0> sorts and expands from LDS (2 x 48 KB)
constexpr uint32_t SP_LDS_BYTES = (2 * SPB * SP_MF_MAX + SP_MF_MAX + SP_MD_MAX) * 8 + (2 * SP_MF_MAX + 8) * 4;

struct SpEnt {
    uint32_t key;  // kb << 1 | kind
    float re, im;
};

// entry(i, p, x): bin position p (<= L/2) and value x of list entry i, false if the entry is void;
// gload(j): what ev wants to know about sample j of the padded signal (loaded four samples ahead of the
// arithmetic); ev(j, re, g): sample j before the division by L.  zl: 2 K entries of scratch.
// LDS carve of the sparse inverse: tile buffers T, U (SPB x Mf points each), W_Mf and W_Md tables, bucket bounds
struct SpLds {
    float2 *T, *U, *wf, *wd;
    uint32_t *beg, *end;
};
DEVI SpLds sp_lds(const DevPlan &P, unsigned char *lds)
{
    SpLds s;
    s.T = (float2 *)lds;
    s.U = s.T + SPB * P.sp_mf;
    s.wf = s.U + SPB * P.sp_mf;
    s.wd = s.wf + P.sp_mf;
    s.beg = (uint32_t *)(s.wd + P.sp_md);
    s.end = s.beg + P.sp_mf + 4;
    return s;
}

// first half: the tables and the list bucketed by ka in zl[], bucket bounds in LDS (beg / end)
template <class EntryFn>
DEVI void sparse_bucket(const DevPlan &P, uint32_t K, EntryFn entry, SpEnt *zl, const float2 *tw,
                        unsigned char *lds, uint32_t *wsum)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t Mf = P.sp_mf, Md = P.sp_md, M = P.M, L = P.L;
    const bool half = P.half != 0;
    const SpLds sl = sp_lds(P, lds);
    float2 *wf = sl.wf, *wd = sl.wd;
    uint32_t *beg = sl.beg, *end = sl.end;
    const uint32_t mg_mf = (uint32_t)(0x100000000ull / Mf) + 1u;

    for (uint32_t e = tid; e < Mf; e += LT) {
        wf[e] = tw[e * (L / Mf)];
        beg[e] = 0;
    }
    for (uint32_t e = tid; e < Md; e += LT) wd[e] = tw[e * (L / Md)];
    __syncthreads();

    // points of the packed (even L) or full (odd L) conjugated spectrum that entry i feeds
    auto points = [&](uint32_t i, uint32_t (&kk)[2], float2 (&vv)[2]) -> uint32_t {
        uint32_t p;
        float2 x;
        if (!entry(i, p, x)) return 0;
        uint32_t c = 0;
        if (half) {
            // k_compress_large's dense form, one source bin at a time:
            //   Z[k] = conj(E + i O),  E = (X[k] + conj X[M-k]) / 2,  O = (X[k] - conj X[M-k]) / 2 * w^k
            if (p < M) {
                const float2 h = make_float2(0.5f * x.x, 0.5f * x.y);
                const float2 o = cmulp(h, tw[p]);
                kk[c] = p;
                vv[c] = make_float2(h.x - o.y, -(h.y + o.x));
                ++c;
            }
            if (p >= 1) {
                const float2 e = make_float2(0.5f * x.x, -0.5f * x.y);
                const float2 d = make_float2(-0.5f * x.x, 0.5f * x.y);
                const float2 o = cmulp(d, tw[M - p]);
                kk[c] = (M - p) | 0x80000000u;
                vv[c] = make_float2(e.x - o.y, -(e.y + o.x));
                ++c;
            }
        } else {
            kk[c] = p;
            vv[c] = make_float2(x.x, -x.y);
            ++c;
            if (p >= 1 && 2 * p != L) {
                kk[c] = (L - p) | 0x80000000u;
                vv[c] = make_float2(x.x, x.y);
                ++c;
            }
        }
        return c;
    };
    for (uint32_t i = tid; i < K; i += LT) {
        uint32_t kk[2];
        float2 vv[2];
        const uint32_t c = points(i, kk, vv);
        for (uint32_t q = 0; q < c; ++q) {
            const uint32_t k = kk[q] & 0x7fffffffu;
            atomicAdd(&beg[k - __umulhi(k, mg_mf) * Mf], 1u);
        }
    }
    __syncthreads();
    (void)block_excl_scan<LW>(beg, Mf, wsum);
    for (uint32_t e = tid; e < Mf; e += LT) end[e] = beg[e];
    __syncthreads();
    for (uint32_t i = tid; i < K; i += LT) {
        uint32_t kk[2];
        float2 vv[2];
        const uint32_t c = points(i, kk, vv);
        for (uint32_t q = 0; q < c; ++q) {
            const uint32_t k = kk[q] & 0x7fffffffu;
            const uint32_t kb = __umulhi(k, mg_mf), ka = k - kb * Mf;
            const uint32_t slot = atomicAdd(&end[ka], 1u);
            SpEnt z;
            z.key = (kb << 1) | (kk[q] >> 31);
            z.re = vv[q].x;
            z.im = vv[q].y;
            zl[slot] = z;
        }
    }
    __syncthreads();
    for (uint32_t ka = tid; ka < Mf; ka += LT) {
        const uint32_t b = beg[ka], e = end[ka], c = e - b;
        if (c <= 1) continue;
        if (c <= 8) {
            // the usual bucket: all of it into registers at once (one memory round trip instead of a chain
            // of dependent ones), a 19-exchange network, back
            SpEnt v[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
                v[i].key = 0xFFFFFFFFu;
                v[i].re = v[i].im = 0.0f;
                if (i < c) v[i] = zl[b + i];
            }
            auto cx = [&](SpEnt &x, SpEnt &y) {
                if (x.key > y.key) { const SpEnt t = x; x = y; y = t; }
            };
            cx(v[0], v[1]); cx(v[2], v[3]); cx(v[4], v[5]); cx(v[6], v[7]);
            cx(v[0], v[2]); cx(v[1], v[3]); cx(v[4], v[6]); cx(v[5], v[7]);
            cx(v[1], v[2]); cx(v[5], v[6]);
            cx(v[0], v[4]); cx(v[1], v[5]); cx(v[2], v[6]); cx(v[3], v[7]);
            cx(v[2], v[4]); cx(v[3], v[5]);
            cx(v[1], v[2]); cx(v[3], v[4]); cx(v[5], v[6]);
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
                if (i < c) zl[b + i] = v[i];
            continue;
        }
        for (uint32_t a = b + 1; a < e; ++a) {
            const SpEnt t = zl[a];
            uint32_t j = a;
            while (j > b && zl[j - 1].key > t.key) { zl[j] = zl[j - 1]; --j; }
            zl[j] = t;
        }
    }
    __syncthreads();

}

// second half: output columns jb0 .. jb0 + SPB - 1 (needs the tables and bucket bounds in LDS, the list in zl[])
template <class LoadFn, class EvalFn>
DEVI void sparse_tile(const DevPlan &P, uint32_t jb0, const SpEnt *zl, const float2 *tw, unsigned char *lds,
                      LoadFn gload, EvalFn ev, uint32_t dbg = 0)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t Mf = P.sp_mf, Md = P.sp_md, sc = P.sc;
    const bool half = P.half != 0;
    const SpLds sl = sp_lds(P, lds);
    float2 *T = sl.T, *U = sl.U;
    const float2 *wf = sl.wf, *wd = sl.wd;
    const uint32_t *beg = sl.beg, *end = sl.end;
    const uint32_t mg_md = Md >= 2 ? (uint32_t)(0x100000000ull / Md) + 1u : 0u;
    {
        const uint32_t nseq = min(SPB, Md - jb0);
        for (uint32_t w = tid; w < Mf * SPB; w += LT) {
            const uint32_t c = w & (SPB - 1), ka = w / SPB;
            if (c >= nseq) continue;
            const uint32_t jb = jb0 + c;
            float2 acc = make_float2(0.0f, 0.0f);
            const uint32_t e1 = (dbg & 1) ? 0u : end[ka];
            for (uint32_t e = beg[ka]; e < e1; ++e) {
                const SpEnt z = zl[e];
                const uint32_t x = jb * (z.key >> 1);
                const float2 t = cmulc(make_float2(z.re, z.im), wd[Md >= 2 ? mod_magic(x, Md, mg_md) : 0u]);
                acc.x += t.x;
                acc.y += t.y;
            }
            T[ka * SPB + c] = cmulc(acc, tw[jb * ka * sc]);
        }
        __syncthreads();
        const float2 *R = (dbg & 2) ? T : lds_fft<true, SPB>(T, U, wf, Mf, nseq, SPB, 1);
        if (dbg & 4) { __syncthreads(); return; }
        if ((tid & (SPB - 1)) < nseq) {
            const uint32_t npt = Mf * SPB, jc = jb0 + (tid & (SPB - 1));
            for (uint32_t w0 = tid; w0 < npt; w0 += 4 * LT) {
                decltype(gload(0u)) g0[4], g1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t w = w0 + u * LT;
                    if (w < npt) {
                        const uint32_t j = Md * (w / SPB) + jc;
                        if (half) {
                            g0[u] = gload(2 * j);
                            g1[u] = gload(2 * j + 1);
                        } else {
                            g0[u] = gload(j);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t w = w0 + u * LT;
                    if (w < npt) {
                        const float2 f = R[w];
                        const uint32_t j = Md * (w / SPB) + jc;
                        if (half) {  // idft_L = 2 * idft_M: even sample -> re, odd sample -> -im
                            ev(2 * j, 2.0f * f.x, g0[u]);
                            ev(2 * j + 1, -2.0f * f.y, g1[u]);
                        } else {
                            ev(j, f.x, g0[u]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <class EntryFn, class LoadFn, class EvalFn>
DEVI void sparse_inverse(const DevPlan &P, uint32_t K, EntryFn entry, SpEnt *zl, const float2 *tw,
                         unsigned char *lds, uint32_t *wsum, LoadFn gload, EvalFn ev, uint32_t dbg = 0)
{
    sparse_bucket(P, K, entry, zl, tw, lds, wsum);
    for (uint32_t jb0 = 0; jb0 < P.sp_md; jb0 += SPB) sparse_tile(P, jb0, zl, tw, lds, gload, ev, dbg);
}

// sort of u64 run records rec = (start << 32 | end) by (bits of xs[end], start)
DEVI void sort_runs_g(uint64_t *rec, const double *xs, uint32_t count, uint32_t P2)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t npairs = P2 >> 1;
    auto ce = [&](uint32_t i, uint32_t l) {
        if (l < count) {
            const uint64_t ra = rec[i], rb = rec[l];
            const uint64_t ka = (uint64_t)__double_as_longlong(xs[(uint32_t)ra]);
            const uint64_t kb = (uint64_t)__double_as_longlong(xs[(uint32_t)rb]);
            if (ka > kb || (ka == kb && ra > rb)) { rec[i] = rb; rec[l] = ra; }
        }
    };
    uint32_t lk = 1;
    for (uint32_t k = 2; k <= P2; k <<= 1, ++lk) {
        const uint32_t half = k >> 1;
        for (uint32_t t = tid; t < npairs; t += LT) {
            const uint32_t i = ((t >> (lk - 1)) << lk) | (t & (half - 1));
            ce(i, i ^ (k - 1));
        }
        __syncthreads();
        for (uint32_t j = half >> 1; j >= 1; j >>= 1) {
            for (uint32_t t = tid; t < npairs; t += LT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                ce(i, i | j);
            }
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------------
// Pre-pass of the large tier.  A frame's forward transform does not depend on anything the frame
// decides later, and one workgroup per frame leaves most of the GPU idle when a batch has fewer large
// frames than CUs (80 frames for BASELINE's 10 M samples in the reference chunker's framing).  So the
// forward transform, the untangle step and the norms of ALL large frames of a launch run first, as
// three kernels over (tile, frame) grids that use every CU; k_compress_large (prm.prefft) then starts
// from the spectrum in its workspace slot.  Same two-pass M1 x M2 scheme as fft_tiled_g.
// --------------------------------------------------------------------------------------------
constexpr int PT = 1024;  // threads of a pre-pass workgroup (a tile stage is ~1300 butterflies)

struct PreFrame {
    const double *xs;
    unsigned char *ws;
    uint32_t n, L, M, pre, half, M1, M2, sc, bins;
};
DEVI const double *frame_samples(const double *samples, const DevFrame &fr) { return samples + fr.sample_off; }
DEVI const double *frame_samples(const double *, const DevDFrame &) { return nullptr; }  // decoder: no samples
DEVI double *frame_out(double *, const DevFrame &) { return nullptr; }                  // encoder: no decoded output
DEVI double *frame_out(double *outp, const DevDFrame &fr) { return outp + fr.out_off; }
template <class FR>
DEVI PreFrame pre_frame(const double *samples, const FR *frames, const uint32_t *ids,
                        const DevPlan *plans, unsigned char *ws_base, uint64_t ws_stride, const DevPlan *&P)
{
    const FR fr = frames[ids[blockIdx.y]];
    P = &plans[fr.plan];
    PreFrame f;
    f.xs = frame_samples(samples, fr);
    f.ws = ws_base + (uint64_t)blockIdx.y * ws_stride;
    f.n = fr.n; f.L = P->L; f.M = P->M; f.pre = P->pre; f.half = P->half;
    f.M1 = P->f4_m1; f.M2 = P->f4_m2; f.sc = P->sc; f.bins = P->bins;
    return f;
}
// State a large frame's decoder leaves for the batched inverse transform (workspace, o_cnt)
struct DecPending {
    uint32_t pending;  // 1: the conjugated packed spectrum waits in buffer A
    float mxf, mnf;
    uint32_t pad;
};
// pass 1: FB columns of the packed (even L) or complex (odd L) padded f32 signal.  FROM_WS: the
// input is what the decoder left in buffer A (frames without a pending transform are skipped).
// (8 waves per SIMD = two workgroups per CU: the register budget is 64)
template <class FR, bool FROM_WS>
__global__ __launch_bounds__(PT, 8) void k_large_pre1(const double *__restrict__ samples,
                                                    const FR *__restrict__ frames,
                                                    const uint32_t *__restrict__ ids,
                                                    const DevPlan *__restrict__ plans,
                                                    const float2 *__restrict__ twpool,
                                                    unsigned char *__restrict__ ws_base, uint64_t ws_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevPlan *P;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, P);
    const uint32_t c0 = blockIdx.x * FB;
    if (c0 >= f.M2) return;
    const LargeWs lay = large_ws_layout(f.n, f.L, P->kcap);
    float2 *Y = (float2 *)(f.ws + lay.o_b);
    const float2 *Xin = (const float2 *)(f.ws + lay.o_a);
    if (FROM_WS) {
        if (((const DecPending *)(f.ws + lay.o_cnt))->pending == 0) return;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        *(uint32_t *)(f.ws + lay.o_cnt) = 0;
    }
    const float2 *tw = twpool + P->tw_off;
    float2 *T = (float2 *)smem, *U = T + FB * f.M1, *w1 = U + FB * f.M1;
    for (uint32_t e = threadIdx.x; e < f.M1; e += PT) w1[e] = tw[e * (f.M2 * f.sc)];
    const uint32_t nseq = min(FB, f.M2 - c0);
    auto g = [&](uint32_t j) -> float {  // fft.rs:184-204, then `as f32`
        int32_t i = (int32_t)j - (int32_t)f.pre;
        i = i < 0 ? 0 : (i >= (int32_t)f.n ? (int32_t)f.n - 1 : i);
        return (float)f.xs[i];
    };
    // F4_MAX * FB / PT <= 8 points per thread: every global load of a thread is issued before the
    // first value is used (a tile is a latency-bound gather otherwise)
    constexpr uint32_t PU = (F4_MAX * FB + PT - 1) / PT;
    {
        float2 v[PU];
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            const uint32_t c = w & (FB - 1), n1 = w >> 4;
            v[u] = make_float2(0.0f, 0.0f);
            if (w < f.M1 * FB && c < nseq) {
                const uint32_t i = f.M2 * n1 + c0 + c;
                v[u] = FROM_WS ? Xin[i] : f.half ? make_float2(g(2 * i), g(2 * i + 1)) : make_float2(g(i), 0.0f);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            const uint32_t c = w & (FB - 1), n1 = w >> 4;
            if (w < f.M1 * FB && c < nseq) T[n1 * FB + c] = v[u];
        }
    }
    __syncthreads();
    const float2 *R = lds_fft<true>(T, U, w1, f.M1, nseq, FB, 1, PT);
    {
        float2 t[PU];
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            const uint32_t c = w & (FB - 1), k1 = w >> 4;
            t[u] = make_float2(1.0f, 0.0f);
            if (w < f.M1 * FB && c < nseq) t[u] = tw[(c0 + c) * k1 * f.sc];
        }
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            const uint32_t c = w & (FB - 1), k1 = w >> 4;
            if (w < f.M1 * FB && c < nseq) Y[k1 * f.M2 + c0 + c] = cmulc(R[k1 * FB + c], t[u]);
        }
    }
}
// pass 2: FB rows
template <class FR, bool FROM_WS>
__global__ __launch_bounds__(PT) void k_large_pre2(const double *__restrict__ samples,
                                                    const FR *__restrict__ frames,
                                                    const uint32_t *__restrict__ ids,
                                                    const DevPlan *__restrict__ plans,
                                                    const float2 *__restrict__ twpool,
                                                    unsigned char *__restrict__ ws_base, uint64_t ws_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevPlan *P;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, P);
    const uint32_t r0 = blockIdx.x * FB;
    if (r0 >= f.M1) return;
    const LargeWs lay = large_ws_layout(f.n, f.L, P->kcap);
    if (FROM_WS && ((const DecPending *)(f.ws + lay.o_cnt))->pending == 0) return;
    const float2 *Y = (const float2 *)(f.ws + lay.o_b);
    float2 *X = (float2 *)(f.ws + lay.o_a);
    const float2 *tw = twpool + P->tw_off;
    const uint32_t ld = f.M2 + 1;
    float2 *T = (float2 *)smem, *U = T + FB * ld, *w2 = U + FB * ld;
    for (uint32_t e = threadIdx.x; e < f.M2; e += PT) w2[e] = tw[e * (f.M1 * f.sc)];
    const uint32_t nseq = min(FB, f.M1 - r0);
    const uint32_t mg_m2 = (uint32_t)(0x100000000ull / f.M2) + 1u;
    {
        constexpr uint32_t PU = (F4_MAX * FB + PT - 1) / PT;
        float2 v[PU];
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            v[u] = make_float2(0.0f, 0.0f);
            if (w < nseq * f.M2) v[u] = Y[r0 * f.M2 + w];  // FB rows are one contiguous run
        }
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            if (w < nseq * f.M2) {
                const uint32_t r = __umulhi(w, mg_m2), n2 = w - r * f.M2;
                T[r * ld + n2] = v[u];
            }
        }
    }
    __syncthreads();
    const float2 *R = lds_fft<false>(T, U, w2, f.M2, nseq, 1, ld, PT);
    for (uint32_t w = threadIdx.x; w < f.M2 * FB; w += PT) {
        const uint32_t r = w & (FB - 1), k2 = w >> 4;
        if (r < nseq) X[(r0 + r) + f.M1 * k2] = R[r * ld + k2];
    }
}
// Pass 2 of the encoder's pre-pass with the untangle step, the norms and the zero count fused in (k_large_pre2 +
// the former k_large_pre3 in one kernel): the row transforms' results never go to memory as Z.  Untangling bin
// k = k1 + M1 k2 needs Z[k] and Z[M - k], and M - k = (M1 - k1) + M1 (M2 - 1 - k2): the partner lives in the
// mirrored row at the mirrored column.  So a workgroup takes FBH rows a0 .. a0 + FBH - 1 TOGETHER WITH their
// mirrors M1 - a (2 FBH rows in LDS, as many as k_large_pre2's tile); row 0 (its own mirror, columns k2 and
// M2 - k2, plus bin M) and, for even M1, the middle row go to workgroup 0.  Arithmetic as in fft_untangle (atsc_kernels.hip),
// operation for operation.  The spectrum lands in buffer A (buffer B still holds pass 1's output while other
// workgroups read it).
constexpr uint32_t FBH = FB / 2;
__global__ __launch_bounds__(PT) void k_large_pre23(const double *__restrict__ samples,
                                                    const DevFrame *__restrict__ frames,
                                                    const uint32_t *__restrict__ ids,
                                                    const DevPlan *__restrict__ plans,
                                                    const float2 *__restrict__ twpool,
                                                    unsigned char *__restrict__ ws_base, uint64_t ws_stride,
                                                    int sparse_inv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevPlan *P;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, P);
    const uint32_t M1 = f.M1, M2 = f.M2, M = f.M;
    const uint32_t half_pairs = (M1 - 1) / 2;  // rows 1 .. half_pairs pair with M1 - 1 .. M1 - half_pairs
    uint32_t a0 = 0, cnt = 0, nrow;
    if (blockIdx.x == 0) {
        nrow = (M1 % 2 == 0 && M1 >= 2) ? 2u : 1u;  // row 0, and the self-mirrored row M1 / 2
    } else {
        const uint32_t first = (blockIdx.x - 1) * FBH;
        if (first >= half_pairs) return;
        a0 = 1 + first;
        cnt = min(FBH, half_pairs - first);
        nrow = 2 * cnt;
    }
    auto row_of = [&](uint32_t r) -> uint32_t {
        if (blockIdx.x == 0) return r == 0 ? 0u : M1 / 2;
        return r < cnt ? a0 + r : M1 - a0 - cnt + 1 + (r - cnt);  // both groups ascending
    };
    const LargeWs lay = large_ws_layout(f.n, f.L, P->kcap);
    const float2 *Y = (const float2 *)(f.ws + lay.o_b);
    float2 *spec = (float2 *)(f.ws + lay.o_a);
    uint32_t *nbits = (uint32_t *)(f.ws + lay.o_nb);
    float2 *Xs = (float2 *)(f.ws + lay.o_x);
    const float2 *tw = twpool + P->tw_off;
    // Tile layout: point n2 of row r at n2 * SI + r (the rows interleaved, SI = FB + 1).  With a row's points
    // contiguous instead (k_large_pre2's layout) consecutive lanes are consecutive butterflies of one row, and a
    // radix-4 stage writes its outputs 4 points apart: 8 dwords, four banks for 64 lanes.  Interleaved, consecutive
    // lanes are consecutive rows -- consecutive banks in every stage -- and the odd stride keeps the transposing
    // store of the contiguous rows off a single bank as well.
    constexpr uint32_t SI = FB + 1;
    float2 *T = (float2 *)smem, *U = T + SI * M2, *w2 = U + SI * M2;
    for (uint32_t e = threadIdx.x; e < M2; e += PT) w2[e] = tw[e * (M1 * f.sc)];
    const uint32_t mg_m2 = (uint32_t)(0x100000000ull / M2) + 1u;
    {
        constexpr uint32_t PU = (F4_MAX * FB + PT - 1) / PT;
        float2 v[PU];
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            v[u] = make_float2(0.0f, 0.0f);
            if (w < nrow * M2) {
                const uint32_t r = __umulhi(w, mg_m2), n2 = w - r * M2;
                v[u] = Y[row_of(r) * M2 + n2];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < PU; ++u) {
            const uint32_t w = threadIdx.x + u * PT;
            if (w < nrow * M2) {
                const uint32_t r = __umulhi(w, mg_m2), n2 = w - r * M2;
                T[n2 * SI + r] = v[u];
            }
        }
    }
    __syncthreads();
    const float2 *R = lds_fft<true>(T, U, w2, M2, nrow, SI, 1, PT);
    const bool dense = !(sparse_inv && P->sp_mf);
    uint32_t zeros = 0;
    auto finish = [&](uint32_t k, float2 z) {
        spec[k] = z;
        nbits[k] = __float_as_uint((float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y));
        zeros += (z.x != 0.0f || z.y != 0.0f) ? 0u : 1u;
        if (dense) Xs[k] = make_float2(0.0f, 0.0f);  // the dense ladder's admitted spectrum
    };
    // A thread's row is fixed (w = tid + u PT, PT a multiple of FB): row, partner row and the bin of column 0 are
    // computed once; every twiddle load of a thread is issued before the first one is used (a dependent global
    // load per bin otherwise, with only two workgroups on the CU to hide it).
    constexpr uint32_t PU2 = (F4_MAX * FB + PT - 1) / PT;
    const uint32_t r = threadIdx.x & (FB - 1), kc0 = threadIdx.x >> 4;
    const bool live = r < nrow;
    const uint32_t k1 = live ? row_of(r) : 0u;
    const float2 *Rrow = R + r;   // point k2 of this thread's row: Rrow[k2 * SI]
    // partner row and the column map k2 -> partner column (row 0: M2 - k2 with 0 -> 0; otherwise M2 - 1 - k2)
    const bool row0 = blockIdx.x == 0 && r == 0;
    const float2 *Rpart = R + ((blockIdx.x == 0) ? r : (nrow - 1 - r));
    float2 twk[PU2];
#pragma unroll
    for (uint32_t u = 0; u < PU2; ++u) {
        const uint32_t k2 = kc0 + u * (PT / FB);
        twk[u] = make_float2(1.0f, 0.0f);
        if (live && k2 < M2 && f.half) twk[u] = tw[k1 + M1 * k2];
    }
    auto untangle = [&](float2 zk, float2 zm, float2 wk) -> float2 {  // see fft_untangle (atsc_kernels.hip)
        const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);
        const float2 b = make_float2(zk.x - zm.x, zk.y + zm.y);
        const float2 t = cmulc(make_float2(b.y, -b.x), wk);
        return make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
    };
#pragma unroll
    for (uint32_t u = 0; u < PU2; ++u) {
        const uint32_t k2 = kc0 + u * (PT / FB);
        if (!live || k2 >= M2) continue;
        const uint32_t k = k1 + M1 * k2;
        const float2 zk = Rrow[k2 * SI];
        if (!f.half) {  // complex transform of the real signal: bins 0 .. L / 2 are kept
            if (k < f.bins) finish(k, zk);
            else spec[k] = zk;
            continue;
        }
        const float2 zm = Rpart[(row0 ? (k2 == 0 ? 0u : M2 - k2) : (M2 - 1 - k2)) * SI];
        finish(k, untangle(zk, zm, twk[u]));
        if (k == 0) finish(M, untangle(zk, zk, tw[M]));  // bin M: Z[0] with Z[0]
    }
    // fft.rs:249-252 needs the number of non-zero bins; zero bins are the rare ones, so they are what gets counted
    if (__ballot(zeros != 0)) {
        zeros = wave_sum_u32(zeros);
        if ((threadIdx.x & 63) == 0) atomicAdd((uint32_t *)(f.ws + lay.o_cnt), zeros);
    }
}
// Statistics of one column tile of the forward transform's first pass (k_large_cols243 sees every sample of the frame
// exactly once): plain stores, one record per tile, combined by the readers -- no atomics, nothing to initialise.
struct TileStats {
    double mn, mx;            // +inf / -inf when the tile saw no comparable sample
    uint32_t frac, runs, ibytes, pad;
};
constexpr uint32_t TST_MAX = 32;  // records per frame (M2 / 16 tiles; 18 for 131072 samples)

// --------------------------------------------------------------------------------------------
// k_compress_large
// --------------------------------------------------------------------------------------------
constexpr uint32_t LKEYS_MAX = 16384;  // LDS sort capacity (kcap of a 131072-sample frame is 13100)

// --------------------------------------------------------------------------------------------
// Few large frames (split run, see launch_compress_large): the frame statistics and the first polynomial
// trip, both plain walks over the samples, run as (chunk, frame) grids over the whole GPU before the
// per-frame kernel instead of inside it on the frame's one CU.
// --------------------------------------------------------------------------------------------
constexpr uint32_t LCH = 4096;  // samples per chunk
struct LargeStats {             // at o_cnt; `zeros` is the pre-pass's count of zero bins (k_large_pre23)
    uint32_t zeros, frac, runs, ibytes;
    unsigned long long kmin, kmax;  // min / max as order-preserving integer keys (integer atomics combine chunks)
};
DEVI unsigned long long f64_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
DEVI double f64_unkey(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
// The frame statistics, from k_large_stats' record or from the column tiles' records (k_large_cols243<true>)
struct FrameStats {
    double mn, mx;
    uint32_t frac, runs, ibytes;
};
DEVI FrameStats frame_stats(const unsigned char *ws, const LargeWs &lay, bool tiles, uint32_t ntiles, uint32_t nchunks = 0)
{
    FrameStats r;
    if (!tiles) {
        const LargeStats *lst = (const LargeStats *)(ws + lay.o_cnt);
        r.mn = f64_unkey(lst->kmin); r.mx = f64_unkey(lst->kmax);
        r.frac = lst->frac; r.runs = lst->runs; r.ibytes = lst->ibytes;
        return r;
    }
    const TileStats *t = (const TileStats *)(ws + lay.o_tst);
    r.mn = __longlong_as_double(0x7ff0000000000000ll); r.mx = -r.mn;
    r.frac = r.runs = r.ibytes = 0;
    for (uint32_t i = 0; i < ntiles; ++i) {
        const TileStats q = t[i];
        if (q.mn < r.mn) r.mn = q.mn;
        if (q.mx > r.mx) r.mx = q.mx;
        r.frac |= q.frac;
    }
    const uint2 *pc = (const uint2 *)(ws + lay.o_part + 1536);  // the polynomial chunks' run counts (0: the caller has no use for them)
    for (uint32_t i = 0; i < nchunks; ++i) { r.runs += pc[i].x; r.ibytes += pc[i].y; }
    return r;
}
#include "atsc_large_cols.h"

__global__ void k_large_stats0(const DevFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
                               const DevPlan *__restrict__ plans, unsigned char *__restrict__ ws_base,
                               uint64_t ws_stride)
{
    const DevPlan &P = plans[frames[ids[blockIdx.x]].plan];
    const LargeWs lay = large_ws_layout(P.n, P.L, P.kcap);
    if (threadIdx.x == 0) {
        LargeStats *st = (LargeStats *)(ws_base + (uint64_t)blockIdx.x * ws_stride + lay.o_cnt);
        st->zeros = st->frac = st->runs = st->ibytes = 0;
        st->kmin = ~0ull;
        st->kmax = 0ull;
    }
}
// utils/mod.rs min / max scan with plain `<` `>`, optimizer/utils.rs split_n, rle.rs run starts -- one chunk
__global__ __launch_bounds__(LT) void k_large_stats(const double *__restrict__ samples,
                                                    const DevFrame *__restrict__ frames,
                                                    const uint32_t *__restrict__ ids,
                                                    const DevPlan *__restrict__ plans,
                                                    unsigned char *__restrict__ ws_base, uint64_t ws_stride)
{
    __shared__ double sred[64];
    __shared__ uint32_t ured[48];
    const uint32_t tid = threadIdx.x;
    const DevFrame fr = frames[ids[blockIdx.y]];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = P.n, c0 = blockIdx.x * LCH;
    if (c0 >= n) return;
    const uint32_t c1 = min(c0 + LCH, n);
    const double *xs = samples + fr.sample_off;
    double mn = __longlong_as_double(0x7ff0000000000000ll), mx = -mn;
    uint32_t frac = 0, runs = 0, ib = 0;
    double v[LCH / LT], pv[LCH / LT];
#pragma unroll
    for (uint32_t u = 0; u < LCH / LT; ++u) {
        const uint32_t j = c0 + u * LT + tid;
        v[u] = pv[u] = 0.0;
        if (j < c1) {
            v[u] = xs[j];
            if (j) pv[u] = xs[j - 1];
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < LCH / LT; ++u) {
        const uint32_t j = c0 + u * LT + tid;
        if (j < c1) {
            frac |= frac_nonzero(v[u]) ? 1u : 0u;
            if (v[u] > mx) mx = v[u];
            if (v[u] < mn) mn = v[u];
            if (j == 0 || v[u] != pv[u]) { ++runs; ib += vlen(j); }
        }
    }
    const double wmn = wave_minmax_f64<true>(mn), wmx = wave_minmax_f64<false>(mx);
    const uint32_t wfr = wave_sum_u32(frac), wru = wave_sum_u32(runs), wib = wave_sum_u32(ib);
    if ((tid & 63) == 0) {
        sred[tid >> 6] = wmn; sred[16 + (tid >> 6)] = wmx;
        ured[tid >> 6] = wfr; ured[16 + (tid >> 6)] = wru; ured[32 + (tid >> 6)] = wib;
    }
    __syncthreads();
    if (tid == 0) {
        double a = sred[0], b = sred[16];
        uint32_t f = 0, r = 0, i2 = 0;
        for (uint32_t w = 0; w < LT / 64; ++w) {
            if (sred[w] < a) a = sred[w];
            if (sred[16 + w] > b) b = sred[16 + w];
            f += ured[w]; r += ured[16 + w]; i2 += ured[32 + w];
        }
        const LargeWs lay = large_ws_layout(P.n, P.L, P.kcap);
        LargeStats *st = (LargeStats *)(ws_base + (uint64_t)blockIdx.y * ws_stride + lay.o_cnt);
        if (a == a && a <= b) {  // the chunk held at least one comparable sample
            atomicMin(&st->kmin, f64_key(a));
            atomicMax(&st->kmax, f64_key(b));
        }
        if (f) atomicOr(&st->frac, 1u);
        if (r) atomicAdd(&st->runs, r);
        if (i2) atomicAdd(&st->ibytes, i2);
    }
}
// First trip of the polynomial ladder (polynomial.rs:209-277: points = max(3, n/100), the plan's pstep[0] /
// pK[0]), Catmull-Rom with the tables of k_compress_large: one chunk's share of the MAPE sum.  Clamp range
// from k_large_stats.
__global__ __launch_bounds__(LT) void k_large_poly1(const double *__restrict__ samples,
                                                    const DevFrame *__restrict__ frames,
                                                    const uint32_t *__restrict__ ids,
                                                    const DevPlan *__restrict__ plans,
                                                    unsigned char *__restrict__ ws_base, uint64_t ws_stride,
                                                    int tile_stats)
{
    __shared__ double4 hbt[256];
    __shared__ double2 mms[LCH / 2 + 4];
    __shared__ double red[48];
    const uint32_t tid = threadIdx.x;
    const DevFrame fr = frames[ids[blockIdx.y]];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = P.n, c0 = blockIdx.x * LCH;
    if (c0 >= n) return;
    const uint32_t c1 = min(c0 + LCH, n);
    const uint32_t step = P.pstep[0], K = P.pK[0];
    if (step <= 1 || step > 256 || K < 2) return;
    unsigned char *ws = ws_base + (uint64_t)blockIdx.y * ws_stride;
    const LargeWs lay = large_ws_layout(P.n, P.L, P.kcap);
    const FrameStats st = frame_stats(ws, lay, tile_stats != 0, (P.f4_m2 + FB - 1) / FB);
    const double smin = st.mn, smax = st.mx;
    const double *xs = samples + fr.sample_off;
    const uint32_t magic = P.pmagic[0];
    const uint32_t gapL = (n - 1) - (K - 2) * step;
    const double stepd = (double)step, gapLd = (double)gapL;
    const double ry = 1.0 / stepd, ryL = 1.0 / gapLd;
    // segments this chunk touches: sgA .. sgB
    uint32_t sgA = __umulhi(c0, magic), sgB = __umulhi(c1 - 1, magic);
    if (sgA > K - 2) sgA = K - 2;
    if (sgB > K - 2) sgB = K - 2;
    for (uint32_t sg = sgA + tid; sg <= sgB; sg += LT) {
        double2 t = make_double2(0.0, 0.0);
        if (sg >= 1 && sg + 2 < K) {
            const uint32_t t0i = sg * step;
            const uint32_t t1i = (sg + 1 == K - 1) ? (n - 1) : (sg + 1) * step;
            const uint32_t tmi = (sg - 1) * step;
            const uint32_t tpi = (sg + 2 == K - 1) ? (n - 1) : (sg + 2) * step;
            const double t0 = (double)t0i, t1 = (double)t1i;
            const double v0 = xs[t0i], v1 = xs[t1i], vm = xs[tmi], vp = xs[tpi];
            t.x = (v1 - vm) / (t1 - (double)tmi) * (t1 - t0);
            t.y = (vp - v0) / ((double)tpi - t0) * (t1 - t0);
        }
        mms[sg - sgA] = t;
    }
    for (uint32_t r = tid; r < step; r += LT) {
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
        hbt[r] = h;
    }
    __syncthreads();
    double s = 0.0;
    double g[LCH / LT], v0[LCH / LT], v1[LCH / LT], pv[LCH / LT];
#pragma unroll
    for (uint32_t u = 0; u < LCH / LT; ++u) {
        const uint32_t i = c0 + u * LT + tid;
        g[u] = v0[u] = v1[u] = pv[u] = 0.0;
        if (i < c1) {
            uint32_t sg = __umulhi(i, magic);
            if (sg > K - 2) sg = K - 2;
            const uint32_t t0i = sg * step;
            g[u] = xs[i];
            v0[u] = xs[t0i];
            v1[u] = xs[(sg == K - 2) ? (n - 1) : t0i + step];
            if (tile_stats && i) pv[u] = xs[i - 1];
        }
    }
    uint32_t runs = 0, ib = 0;  // rle.rs:142-189 run starts and their index bytes (as k_large_stats counts them)
    if (tile_stats) {
#pragma unroll
        for (uint32_t u = 0; u < LCH / LT; ++u) {
            const uint32_t i = c0 + u * LT + tid;
            if (i < c1 && (i == 0 || g[u] != pv[u])) { ++runs; ib += vlen(i); }
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < LCH / LT; ++u) {
        const uint32_t i = c0 + u * LT + tid;
        if (i >= c1) continue;
        double sv;
        if (i == n - 1) {
            sv = g[u];
        } else {
            uint32_t sg = __umulhi(i, magic);
            if (sg > K - 2) sg = K - 2;
            const uint32_t t0i = sg * step;
            const bool last = (sg == K - 2);
            if (sg > 0 && !last) {
                const double2 t = mms[sg - sgA];
                const double4 h = hbt[i - t0i];
                sv = v0[u] * h.x + t.x * h.y + v1[u] * h.z + t.y * h.w;
            } else {
                const double nt = div_small((double)(i - t0i), last ? gapLd : stepd, last ? ryL : ry);
                sv = v0[u] * (1.0 - nt) + v1[u] * nt;
            }
        }
        double o = div1e5(round(sv * 100000.0));
        if (o < smin) o = smin;
        else if (o > smax) o = smax;
        s += fabs((o - g[u]) / g[u]);
    }
    int parity = 0;
    s = block_sum_f64<LW>(s, red, parity);
    if (tid == 0) ((double *)(ws + lay.o_part))[blockIdx.x] = s;
    if (tile_stats) {  // the chunk's run statistics behind the 32 chunk sums (plain stores: the reader adds them up)
        runs = block_sum_u32<LW>(runs, red, parity);
        ib = block_sum_u32<LW>(ib, red, parity);
        if (tid == 0 && blockIdx.x < 32) ((uint2 *)(ws + lay.o_part + 256))[blockIdx.x] = make_uint2(runs, ib);
    }
}

// State of the fast path (atsc_large_fast.h)
struct FastState {  // at LargeWs::o_front
    uint32_t status;  // 0: left to k_compress_large<0>, 1: the first FFT trip's tiles are pending, 2: finished
    uint32_t bitdepth, K1, big, Z;
    uint32_t best_size;
    int32_t best_owner;
    uint32_t poly_final;  // the polynomial ladder ended with its first trip
    uint32_t forced;      // 1: forced FFT (no competition: the first trip's payload is emitted when the ladder ends there)
    uint32_t poly_size, poly_K, poly_step, poly2_lb, rle_lb;
    uint32_t nlist;       // points in the bucketed list
    double smin, smax, poly_err;
    float mxf, mnf;
};
// What the first part of a split run hands to the second (PART 1 -> k_large_trip_tiles -> PART 2)
struct TripState {
    double smin, smax, poly_err, pcur;
    atsc_frame_diag dg;
    uint32_t finished;  // 1: PART 1 handled the whole frame (no first FFT trip to farm out)
    uint32_t bitdepth, rle_size, rle_R, rle_D, rle_ib, rle_lb, best_size;
    uint32_t poly_step, poly_K, poly_size, poly_trips, pjump;
    uint32_t Z, nkeys, used;
    int32_t mode, best_owner;
    uint32_t flags;     // 1 rle_sorted, 2 rle_pending, 4 poly_done, 8 poly_pruned, 16 poly_active
};
constexpr uint32_t TRIP_BOUNDS_OFF = 0, TRIP_PARTIAL_OFF = 8192;  // inside buffer C: bucket bounds, tile sums

// PART 0: the whole per-frame compressor.  PART 1 / PART 2: the same code cut inside the first FFT trip,
// after the admitted bins are bucketed: for a batch with fewer large frames than CUs the trip's tiles --
// independent of each other from there on -- run as a (tile, frame) grid over the whole GPU
// (k_large_trip_tiles) instead of one after the other on the frame's CU; PART 2 adds the tile sums up and
// goes on with the ladder.  A frame whose PART 1 never reaches that point is finished there.
template <int PART>
__device__ __forceinline__ void compress_large_frame(
    const uint32_t bx, const double *__restrict__ samples, const DevFrame *__restrict__ frames,
    const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, const KParams &prm, uint8_t *__restrict__ slots,
    DevResult *__restrict__ res, atsc_frame_diag *__restrict__ diag, unsigned char *__restrict__ ws_base,
    uint64_t ws_stride)
{
    constexpr int T = LT;
    constexpr int W = LW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t fid = ids[bx];
    const DevFrame fr = frames[fid];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = P.n, L = P.L, pre = P.pre, bins = P.bins, M = P.M;

    // LDS: [red 384][hist 256*4 + misc 64][keys 8*LKEYS_MAX | spline basis table]
    double *red = (double *)smem;
    uint32_t *wsum = (uint32_t *)(red + 32);
    uint32_t *hist = (uint32_t *)(smem + 384);
    uint32_t *bc = hist + 256;  // broadcast scalars
    uint64_t *keys = (uint64_t *)(smem + 384 + 1024 + 64);
    double4 *hbt = (double4 *)keys;
    int parity = 0;

    unsigned char *ws = ws_base + (uint64_t)bx * ws_stride;
    const LargeWs lay = large_ws_layout(n, L, P.kcap);
    float2 *A = (float2 *)(ws + lay.o_a);
    float2 *B = (float2 *)(ws + lay.o_b);
    float2 *Cb = (float2 *)(ws + lay.o_c);
    float2 *Xs = (float2 *)(ws + lay.o_x);
    uint32_t *nbits = (uint32_t *)(ws + lay.o_nb);
    Sel *sel = (Sel *)(ws + lay.o_sel);
    double2 *mm = (double2 *)(ws + lay.o_mm);
    uint32_t *aux = (uint32_t *)(ws + lay.o_aux);
    uint64_t *rrec = (uint64_t *)(ws + lay.o_rec);
    uint32_t *rhp = (uint32_t *)(ws + lay.o_hp);
    uint32_t *tab = (uint32_t *)(ws + lay.o_tab);
    uint32_t *rps = (uint32_t *)(ws + lay.o_rps);
    uint32_t *rph = (uint32_t *)(ws + lay.o_rph);
    uint32_t *spos = (uint32_t *)(ws + lay.o_spos);
    float2 *fft_lds = (float2 *)keys;  // the key buffer is idle whenever a transform runs

    const double *xs = samples + fr.sample_off;  // read in place (L2 keeps a 1 MB frame)
    const float2 *tw = twpool + P.tw_off;
    uint8_t *out = slots + fr.slot_off;
    int mode = prm.mode;
    // behind the fast path (atsc_large_fast.h): only the frames it left undecided
    if (PART == 0 && prm.fast_skip && ((const FastState *)(ws + lay.o_front))->status == 2) return;
    if (PART == 0 && prm.fast_skip && prm.debug_stop <= -3 && tid == 0)
    {
        const uint32_t *d = (const uint32_t *)(ws + lay.o_front + 200);
        printf("FASTLEFT frame %u why %u  | %u %u %u %u %x %x %u %u %u\n", fid, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9]);
    }

    auto gpad = [&](uint32_t j) -> double {  // fft.rs:184-204
        int32_t i = (int32_t)j - (int32_t)pre;
        i = i < 0 ? 0 : (i >= (int32_t)n ? (int32_t)n - 1 : i);
        return xs[i];
    };
    LT_STAMP("start");

    // ---- stats --------------------------------------------------------------------------------
    double smin, smax;
    uint32_t bitdepth;
    // the codec a sample-level trial picked replaces Auto further down; known here already, it tells
    // whether the one pass over the samples should count the RLE runs as well (rle.rs:142-189)
    const bool rle_wanted = [&] {
        int m = mode;
        if (m == ATSC_AUTO && prm.trial_res != nullptr && n >= prm.trial_min_n) m = (int)prm.trial_res[fid].chosen;
        return m == ATSC_AUTO || m == ATSC_RLE;
    }();
    uint32_t st_runs = 0, st_ibytes = 0;
    TripState *fst = (TripState *)(ws + lay.o_front);
    if (PART == 1 && tid == 0) fst->finished = 1;  // until the cut says otherwise
    if (PART == 2) {
        if (fst->finished) return;
        smin = fst->smin;
        smax = fst->smax;
        bitdepth = fst->bitdepth;
        mode = fst->mode;
    } else {
        const double x0 = xs[0];
        double mn = x0, mx = x0;
        uint32_t fr_any = 0;
        if (prm.prestats) {
            // k_large_stats walked the samples already (LargeStats); a NaN first sample keeps the reference's
            // outcome of a scan that starts from it
            const LargeStats *lst = (const LargeStats *)(ws + lay.o_cnt);
            if (x0 == x0) {
                mn = f64_unkey(lst->kmin);
                mx = f64_unkey(lst->kmax);
            }
            fr_any = lst->frac;
            if (tid == 0) { st_runs = lst->runs; st_ibytes = lst->ibytes; }
        } else {
        auto visit = [&](uint32_t j, double v, double prev) {
            fr_any |= frac_nonzero(v) ? 1u : 0u;
            if (v > mx) mx = v;
            if (v < mn) mn = v;
            if (rle_wanted && (j == 0 || v != prev)) { ++st_runs; st_ibytes += vlen(j); }
        };
        // pairs of samples, four pairs of a thread in flight (a frame that does not start on a 16-byte
        // boundary takes the scalar loop)
        uint32_t j0 = 0;
        if (((uintptr_t)xs & 15u) == 0) {
            const double2 *x2 = (const double2 *)xs;
            const uint32_t np = n >> 1;
            for (uint32_t q0 = 0; q0 < np; q0 += 4 * T) {
                double2 v[4];
                double pv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t q = q0 + u * T + tid;
                    if (q < np) {
                        v[u] = x2[q];
                        pv[u] = q ? xs[2 * q - 1] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t q = q0 + u * T + tid;
                    if (q < np) {
                        visit(2 * q, v[u].x, pv[u]);
                        visit(2 * q + 1, v[u].y, v[u].x);
                    }
                }
            }
            j0 = 2 * np;
        }
        for (uint32_t j = j0 + tid; j < n; j += T) visit(j, xs[j], j ? xs[j - 1] : 0.0);
        }
        mn = block_minmax_f64<W, true>(mn, red, parity);
        mx = block_minmax_f64<W, false>(mx, red, parity);
        // The extremes are reported with the bits of their first occurrence.  Doubles that compare
        // equal have equal bits unless they are zeros of either sign, so only a zero extreme needs
        // the second pass.
        uint32_t mni = 0xFFFFFFFFu, mxi = 0xFFFFFFFFu;
        if (mn == 0.0 || mx == 0.0) {
            for (uint32_t j = tid; j < n; j += T) {
                const double v = xs[j];
                if (v == mn) mni = min(mni, j);
                if (v == mx) mxi = min(mxi, j);
            }
            mni = block_min_u32<W>(mni, red, parity);
            mxi = block_min_u32<W>(mxi, red, parity);
            smin = (mni < n) ? xs[mni] : x0;
            smax = (mxi < n) ? xs[mxi] : x0;
        } else {
            smin = (mn == mn) ? mn : x0;
            smax = (mx == mx) ? mx : x0;
        }
        fr_any = block_or_u32<W>(fr_any, red, parity);
        int64_t maxi, mini;
        bool fz;
        split_n(smax, maxi, fz);
        split_n(smin, mini, fz);
        bitdepth = fr_any ? 0u : bitdepth_of(maxi, mini);
    }

    atsc_frame_diag dg;
    dg.fft_size = dg.poly_size = dg.rle_size = 0xFFFFFFFFu;
    dg.fft_trips = dg.fft_k = dg.poly_trips = dg.poly_step = 0;
    dg.poly_points = 0;
    dg.fft_err = dg.poly_err = 0.0;
    if (PART == 2) dg = fst->dg;

    if (PART != 2) {
    if (mode == ATSC_CONSTANT || (mode == ATSC_AUTO && !prm.trial && smin == smax)) {
        if (tid == 0) {
            out[0] = 30;
            out[1] = (uint8_t)bitdepth;
            const uint32_t vb = put_value(out + 2, bitdepth, smin);
            res[fid].err = 0.0;
            res[fid].len = 2 + vb;
            res[fid].chosen = ATSC_CONSTANT;
            if (diag) diag[fid] = dg;
        }
        return;
    }
    if (mode == ATSC_NOOP) {
        for (uint32_t j = tid; j < n; j += T) aux[j] = vlen(zigzag(sat_i64(round(xs[j]))));
        __syncthreads();
        const uint32_t tot = lscan(aux, n, wsum);
        const uint32_t hdr = 1 + vlen(n);
        for (uint32_t j = tid; j < n; j += T)
            put_varint(out + hdr + aux[j], zigzag(sat_i64(round(xs[j]))));
        if (tid == 0) {
            out[0] = 250;
            put_varint(out + 1, n);
            res[fid].err = 0.0;
            res[fid].len = hdr + tot;
            res[fid].chosen = ATSC_NOOP;
            if (diag) diag[fid] = dg;
        }
        return;
    }
    if (mode == ATSC_AUTO && prm.trial_res != nullptr && n >= prm.trial_min_n)
        mode = (int)prm.trial_res[fid].chosen;
    }

    // Candidate bookkeeping and pruning exactly as in k_compress (atsc_kernels.hip): the selector keeps
    // the smallest passing payload, first of [FFT, Polynomial, RLE] on ties, and a ladder's payload
    // only grows, so a ladder stops once its next payload cannot beat a candidate that already passes.
    const bool run_fft = (mode == ATSC_AUTO || mode == ATSC_FFT);
    const bool run_poly = (mode == ATSC_AUTO || mode == ATSC_POLYNOMIAL || mode == ATSC_IDW);
    const bool idw = (mode == ATSC_IDW);  // forced codec only (polynomial.rs:29-34,202-207)
    const bool run_rle = (mode == ATSC_AUTO || mode == ATSC_RLE);
    const double me = prm.max_err;
    const bool prune = (mode == ATSC_AUTO) && (0.0 <= me);
    uint32_t best_size = PART == 2 ? fst->best_size : 0xFFFFFFFFu;
    int best_owner = PART == 2 ? fst->best_owner : 3;
    auto can_win = [&](uint32_t size_lb, int owner) {
        return size_lb < best_size || (size_lb == best_size && owner < best_owner);
    };
    auto offer = [&](uint32_t size, int owner) {
        if (can_win(size, owner)) { best_size = size; best_owner = owner; }
    };

    if (prm.debug_stop == 1) return;
    LT_STAMP("stats done");
    // ---- RLE (rle.rs:142-189): bound first; exact right away when there are few runs ----
    uint32_t rle_size = 0xFFFFFFFFu, rle_R = 0, rle_D = 0, rle_ib = 0, rle_lb = 0xFFFFFFFFu;
    bool rle_sorted = false, rle_pending = false;
    if (PART == 2) {
        rle_size = fst->rle_size; rle_R = fst->rle_R; rle_D = fst->rle_D; rle_ib = fst->rle_ib; rle_lb = fst->rle_lb;
        rle_sorted = (fst->flags & 1u) != 0;
        rle_pending = (fst->flags & 2u) != 0;
    }
    auto run_key = [&](uint64_t rec) { return (uint64_t)__double_as_longlong(xs[(uint32_t)rec]); };
    auto rle_sort_and_group = [&]() {
        __syncthreads();
        uint32_t R;
        uint32_t *ends = rhp;
        if (rle_R <= 8192) {
            // The run count is known from the statistics.  Each wavefront walks one contiguous sixteenth
            // of the frame, 64 samples at a time, and appends its run ends in order (ballot ranks) to its
            // own stretch of a scratch list; the stretches are then laid end to end.  One coalesced pass
            // over the samples instead of a flag array, a scan over n and a scatter.
            const uint32_t lane = tid & 63u, wv = tid >> 6;
            const uint32_t seg = (((n + W - 1) / W) + 63u) & ~63u;
            const uint32_t s0 = min(wv * seg, n), s1 = min(s0 + seg, n);
            uint32_t *mine = (uint32_t *)tab + wv * min(seg, 8192u);  // <= 4 n + 4 KB of the (idle) hash-table region
            const uint64_t lt = (1ull << lane) - 1ull;
            uint32_t cnt = 0;
            for (uint32_t jb = s0; jb < s1; jb += 256) {  // four 64-sample windows' loads in flight
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t j = jb + 64 * u + lane;
                    a[u] = j < s1 ? xs[j] : 0.0;
                    b[u] = (j < s1 && j + 1 < n) ? xs[j + 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t j = jb + 64 * u + lane;
                    const bool fl = j < s1 && (j + 1 >= n || b[u] != a[u]);
                    const uint64_t m = __ballot(fl);
                    if (fl) mine[cnt + (uint32_t)__popcll(m & lt)] = j;
                    cnt += (uint32_t)__popcll(m);
                }
            }
            if (lane == 0) wsum[wv] = cnt;
            __syncthreads();
            uint32_t base = 0;
            R = 0;
            for (uint32_t w2 = 0; w2 < (uint32_t)W; ++w2) {
                if (w2 < wv) base += wsum[w2];
                R += wsum[w2];
            }
            for (uint32_t k = lane; k < cnt; k += 64) ends[base + k] = mine[k];
            __syncthreads();
        } else {
        for (uint32_t j = tid; j < n; j += T)
            aux[j] = (j + 1 >= n || xs[j + 1] != xs[j]) ? 1u : 0u;
        __syncthreads();
        R = lscan(aux, n, wsum);
        for (uint32_t j = tid; j < n; j += T)
            if (j + 1 >= n || xs[j + 1] != xs[j]) ends[aux[j]] = j;
        __syncthreads();
        }
        for (uint32_t r = tid; r < R; r += T) {
            const uint32_t st = r ? ends[r - 1] + 1 : 0;
            rrec[r] = ((uint64_t)st << 32) | ends[r];
        }
        __syncthreads();
        uint32_t p2 = 1;
        while (p2 < R) p2 <<= 1;
        if (R <= 8192) {
            // up to 8192 runs: sort (value bits, run index) in LDS -- the run index is the start order --
            // instead of a network over global memory that looks every key up through xs[]
            uint64_t *kk = keys;                      // 8 R bytes
            uint32_t *pp = (uint32_t *)(keys + R);    // 4 R bytes; 12 R <= 96 KB of the 128 KB key buffer
            for (uint32_t r = tid; r < R; r += T) {
                kk[r] = run_key(rrec[r]);
                pp[r] = r;
            }
            __syncthreads();
            block_sort<W, false>(kk, pp, R, p2);
            uint64_t *tmp = (uint64_t *)tab;          // 8 (n + 8) bytes of scratch, free here
            for (uint32_t i = tid; i < R; i += T) tmp[i] = rrec[pp[i]];
            __syncthreads();
            for (uint32_t i = tid; i < R; i += T) rrec[i] = tmp[i];
            __syncthreads();
        } else {
            sort_runs_g(rrec, xs, R, p2);
        }
        for (uint32_t i = tid; i < R; i += T)
            aux[i] = (i == 0 || run_key(rrec[i]) != run_key(rrec[i - 1])) ? 1u : 0u;
        __syncthreads();
        const uint32_t D = lscan(aux, R, wsum);
        for (uint32_t i = tid; i < R; i += T)
            if (i == 0 || run_key(rrec[i]) != run_key(rrec[i - 1])) rhp[aux[i]] = i;
        if (tid == 0) rhp[D] = R;
        __syncthreads();
        uint32_t hb = 0;
        for (uint32_t gi = tid; gi < D; gi += T) {
            const uint32_t h0 = rhp[gi], h1 = rhp[gi + 1];
            const uint32_t b = value_bytes(bitdepth, xs[(uint32_t)rrec[h0]]) + vlen(h1 - h0);
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
    if (PART != 2 && run_rle) {
        rle_R = block_sum_u32<W>(st_runs, red, parity);  // counted with the statistics
        rle_ib = block_sum_u32<W>(st_ibytes, red, parity);
        const uint32_t minval = (bitdepth == 0) ? 8u : 1u;
        rle_lb = 3 + rle_ib + (rle_R >= 2 ? 2u : 1u) * (minval + 1);
        if (mode == ATSC_RLE || rle_R <= 1024) {
            rle_sort_and_group();  // its arrays live in their own workspace regions (aux excepted)
            if (prune) offer(rle_size, 2);
        } else {
            rle_pending = true;
        }
    }

    if (prm.debug_stop == 2) return;
    LT_STAMP("rle done");
    // =========================================================================================
    // Polynomial candidate (polynomial.rs:209-277); forced Idw shares the ladder and swaps the
    // interpolation (polynomial.rs:375-393)
    // =========================================================================================
    // The ladder runs in two goes: its first trip before the FFT candidate, the rest after it.  A frame
    // whose polynomial passes at once with fewer bytes than the FFT's first trip could store never
    // builds the admission order or runs an inverse transform; otherwise the later trips are pruned by
    // the FFT's size exactly as before.  Pruning only skips candidates that cannot win, so the
    // selection does not depend on the order (frame/mod.rs:113-138).
    uint32_t poly_step = 1, poly_K = 0, poly_size = 0xFFFFFFFFu, poly_trips = 0;
    double poly_err = 0.0;
    bool poly_done = false, poly_pruned = false, poly_active = false;
    double pcur = prm.max_err + 1.0;
    uint32_t pjump = 0;
    if (PART == 2) {
        poly_step = fst->poly_step; poly_K = fst->poly_K; poly_size = fst->poly_size; poly_trips = fst->poly_trips;
        poly_err = fst->poly_err; pcur = fst->pcur; pjump = fst->pjump;
        poly_done = (fst->flags & 4u) != 0;
        poly_pruned = (fst->flags & 8u) != 0;
        poly_active = (fst->flags & 16u) != 0;
    }
    auto poly_ladder = [&](uint32_t limit) {
            const uint32_t base = (3 >= n / 100) ? 3 : n / 100;
            const uint32_t dj1 = max(n / 10, 1u), dj2 = max(n / 100, 1u);
            double cur = pcur;
            uint32_t jump = pjump, done_now = 0;
            while (true) {
                if (!(round(cur * 10000.0) > prm.poly_q_hi)) { poly_active = false; break; }
                if (done_now == limit) break;
                ++done_now;
                const uint32_t pts = base + jump;
                const uint32_t step = max(n / pts, 1u);
                const uint32_t cnt = (n + step - 1) / step;
                const uint32_t K = cnt + (((cnt - 1) * step != n - 1) ? 1u : 0u);
                // payload of this trip, exactly (F64 / U8 points) or from below (>= 1 byte per varint)
                if (prune && !can_win(2 + vlen(K) + K * (bitdepth == 0 ? 8u : 1u) + 17, 1)) { poly_pruned = true; poly_active = false; break; }
                ++poly_trips;
                poly_step = step;
                poly_K = K;
                if (idw && step > 1) {
                    // inverse_distance_weight 0.1.1, power 2 (oracle: poly_idw_to_data): a sample that
                    // sits on a point takes its value, every other sample sums w = 1 / d^2 over ALL K
                    // points in ascending order.  d is an integer, so the weights come from a per-frame
                    // table w[d] = 1.0 / (d * d) kept in the (idle) FFT buffer: O(n K) FMAs per trip.
                    double *wtab = (double *)A;
                    if (poly_trips == 1) {
                        for (uint32_t d = tid + 1; d < n; d += T) {
                            const double dd = (double)d;
                            wtab[d] = 1.0 / (dd * dd);
                        }
                        __syncthreads();
                    }
                    double s = 0.0;
                    for (uint32_t i = tid; i < n; i += T) {
                        double sv;
                        const uint32_t q = i / step;
                        if (i == n - 1 || (q * step == i && q < K - 1)) {
                            sv = xs[i];
                        } else {
                            double num = 0.0, den = 0.0;
                            for (uint32_t k = 0; k < K; ++k) {
                                const uint32_t pk = (k == K - 1) ? (n - 1) : k * step;
                                const double w = wtab[pk > i ? pk - i : i - pk];
                                num += w * xs[pk];
                                den += w;
                            }
                            sv = num / den;
                        }
                        double o = div1e5(round(sv * 100000.0));
                        if (o < smin) o = smin;
                        else if (o > smax) o = smax;
                        const double g = xs[i];
                        s += fabs((o - g) / g);
                    }
                    s = block_sum_f64<W>(s, red, parity);
                    cur = s / (double)n;
                } else if (step > 1 && prm.prestats && poly_trips == 1 && step <= 256 && xs[0] == xs[0]) {
                    // k_large_poly1 evaluated this trip chunk by chunk: the sums, in chunk order
                    double s = 0.0;
                    if (tid == 0) {
                        const double *part = (const double *)(ws + lay.o_part);
                        for (uint32_t c = 0; c < (n + LCH - 1) / LCH; ++c) s += part[c];
                    }
                    s = block_sum_f64<W>(s, red, parity);
                    cur = s / (double)n;
                } else if (step > 1) {
                    const uint32_t magic = (uint32_t)(0x100000000ull / step) + 1u;
                    const uint32_t gapL = (n - 1) - (K - 2) * step;
                    const double stepd = (double)step, gapLd = (double)gapL;
                    const double ry = 1.0 / stepd, ryL = 1.0 / gapLd;
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
                    for (uint32_t r = tid; r < step; r += T) {  // step <= 133: Hermite basis per offset
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
                        hbt[r] = h;
                    }
                    __syncthreads();
                    double s = 0.0;
                    // four samples of a thread at a time: their loads (sample, two knots, tangents) are all
                    // issued before the first value is used -- one sample after the other the loop waits
                    // out a memory round trip per sample
                    for (uint32_t i0 = tid; i0 < n; i0 += 4 * T) {
                        double g[4], v0[4], v1[4];
                        double2 tg[4];
                        uint32_t sgs[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t i = i0 + u * T;
                            g[u] = v0[u] = v1[u] = 0.0;
                            tg[u] = make_double2(0.0, 0.0);
                            sgs[u] = 0;
                            if (i < n) {
                                uint32_t sg = __umulhi(i, magic);
                                if (sg > K - 2) sg = K - 2;
                                const uint32_t t0i = sg * step;
                                const bool last = (sg == K - 2);
                                sgs[u] = sg;
                                g[u] = xs[i];
                                v0[u] = xs[t0i];
                                v1[u] = xs[last ? (n - 1) : t0i + step];
                                if (sg > 0 && !last) tg[u] = mm[sg];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t i = i0 + u * T;
                            if (i >= n) continue;
                            double sv;
                            if (i == n - 1) {
                                sv = g[u];
                            } else {
                                const uint32_t sg = sgs[u], t0i = sg * step;
                                const bool last = (sg == K - 2);
                                if (sg > 0 && !last) {
                                    const double4 h = hbt[i - t0i];
                                    sv = v0[u] * h.x + tg[u].x * h.y + v1[u] * h.z + tg[u].y * h.w;
                                } else {
                                    const double nt = div_small((double)(i - t0i), last ? gapLd : stepd,
                                                                last ? ryL : ry);
                                    sv = v0[u] * (1.0 - nt) + v1[u] * nt;
                                }
                            }
                            double o = div1e5(round(sv * 100000.0));
                            if (o < smin) o = smin;
                            else if (o > smax) o = smax;
                            s += fabs((o - g[u]) / g[u]);
                        }
                    }
                    s = block_sum_f64<W>(s, red, parity);
                    cur = s / (double)n;
                }
                if (poly_trips <= 17) jump += dj1;
                else if (poly_trips <= 22) jump += dj2;
                else if (round(cur * 10000.0) < prm.poly_q_lo) { poly_active = false; break; }
                else { poly_step = 1; poly_K = n; cur = 0.0; poly_active = false; break; }
                if (K == n) { cur = 0.0; poly_active = false; break; }
            }
            pcur = cur;
            pjump = jump;
            poly_err = cur;
    };
    auto poly_finish = [&]() {
        uint32_t vb = 0;
        if (bitdepth == 0 || bitdepth == 3) {
            vb = poly_K * (bitdepth == 0 ? 8u : 1u);
        } else {
            for (uint32_t k = tid; k < poly_K; k += T) {
                const uint32_t t = (k == poly_K - 1) ? (n - 1) : k * poly_step;
                vb += value_bytes(bitdepth, xs[t]);
            }
            vb = block_sum_u32<W>(vb, red, parity);
        }
        poly_size = 1 + 1 + vlen(poly_K) + vb + 8 + 8 + 1;
        poly_done = !poly_pruned;
        if (prune && poly_done && poly_err <= me) offer(poly_size, 1);
        dg.poly_size = poly_size; dg.poly_trips = (uint16_t)poly_trips;
        dg.poly_step = (uint16_t)poly_step; dg.poly_points = poly_K; dg.poly_err = poly_err;
    };
    if (PART != 2 && run_poly) {
        if (smax == smin) {
            poly_K = 0;
            poly_step = 1;
        } else if (!prm.bounded) {
            const uint32_t base = (3 >= n / 100) ? 3 : n / 100;
            poly_step = max(n / base, 1u);
            const uint32_t cnt = (n + poly_step - 1) / poly_step;
            poly_K = cnt + (((cnt - 1) * poly_step != n - 1) ? 1u : 0u);
        } else {
            poly_active = true;
            poly_ladder(1);
        }
        if (!poly_active) poly_finish();
    }


    LT_STAMP("poly first trip done");
    // =========================================================================================
    // FFT candidate (fft.rs:288-362)
    // =========================================================================================
    uint32_t fft_k = 0, fft_size = 0xFFFFFFFFu, fft_trips = 0;
    double fft_err = 0.0;
    bool fft_done = false;
    const float mxf = (float)smax, mnf = (float)smin;
    if (run_fft) {
        if (mxf == mnf) {
            fft_k = 0;
            fft_size = 1 + 1 + 8;
            fft_done = true;
        } else if (prune && !can_win(1 + 1 + 9 + 8, 0)) {
            // a single stored bin already loses to a payload that passes
        } else {
            rle_sorted = false;  // aux (group ids) is reused below; the emitter sorts again if RLE wins
            bool fft_pruned = false;
            // ---- forward transform of the padded f32 signal ----
            float2 *spec;
            if (prm.prefft) {
                spec = A;  // k_large_pre1 / k_large_pre23 left the spectrum, nbits, Xs = 0 and the count
            } else if (P.half) {
                float *Af = (float *)A;
                for (uint32_t j = tid; j < L; j += T) Af[j] = (float)gpad(j);
                __syncthreads();
                float2 *Z = fft_large(P, A, B, tw, fft_lds, prm.large_tiled != 0);
                spec = (Z == A) ? B : A;
                for (uint32_t k = tid; k <= M; k += T) {  // untangle (see fft_untangle)
                    const float2 zk = Z[k == M ? 0 : k];
                    const float2 zm = Z[k == 0 ? 0 : M - k];
                    const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);
                    const float2 b = make_float2(zk.x - zm.x, zk.y + zm.y);
                    const float2 t = cmulc(make_float2(b.y, -b.x), tw[k]);
                    spec[k] = make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
                }
                __syncthreads();
            } else {
                for (uint32_t j = tid; j < L; j += T) A[j] = make_float2((float)gpad(j), 0.0f);
                __syncthreads();
                spec = fft_large(P, A, B, tw, fft_lds, prm.large_tiled != 0);
            }
            float2 *work = (spec == A) ? B : A;  // free FFT buffer from here on
            if (prm.debug_stop == 4) return;

            // ---- admission order: the kcap largest norms, descending, ties by position ----
            uint32_t Z;
            if (prm.prefft) {
                Z = bins - *(const uint32_t *)(ws + lay.o_cnt);
            } else {
                uint32_t nz = 0;
                for (uint32_t k = tid; k < bins; k += T) {
                    const float2 z = spec[k];
                    nbits[k] = __float_as_uint((float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y));
                    nz += (z.x != 0.0f || z.y != 0.0f) ? 1u : 0u;
                    if (!(prm.sparse_inv && P.sp_mf)) Xs[k] = make_float2(0.0f, 0.0f);
                }
                __syncthreads();
                Z = block_sum_u32<W>(nz, red, parity);
            }
            // The order is built for the bins the first two trips can ask for (2048 keys), extended to
            // five trips' worth (4096) and to the full kcap only if the ladder gets that far: the sort is
            // most of the cost, and at e = 5 % few frames go beyond a trip or two.
            const uint32_t kcap_total = min(P.kcap, LKEYS_MAX);
            auto build_order = [&](const uint32_t kcap) -> uint32_t {
            uint32_t nkeys = 0;
            if (bins <= kcap) {
                for (uint32_t k = tid; k < bins; k += T)
                    keys[k] = ((uint64_t)(~nbits[k]) << 32) | (uint64_t)k;
                nkeys = bins;
                __syncthreads();
            } else {
                // One histogram pass over the top 11 bits of the norm patterns (exponent + 3 mantissa
                // bits: the counters of one exponent are spread over 8 addresses) finds the digit d* in
                // which the kcap-th largest norm falls; one more pass collects the bins above d* (all
                // admitted) and the bins in d* (candidates) into LDS.  Candidates sort on their own:
                // every key above d* precedes every candidate, so the two sorted lists concatenate.
                // Lists that do not fit take the byte-wise radix select below.
                uint32_t *h2 = (uint32_t *)keys;  // 2048 counters, index 2047 - digit
                for (uint32_t i = tid; i < 2048; i += T) h2[i] = 0;
                if (tid == 0) { bc[2] = 0; bc[3] = 0; bc[4] = 0xFFFFFFFFu; }
                __syncthreads();
                for (uint32_t k0 = tid; k0 < bins; k0 += 4 * T) {  // four loads in flight per thread
                    uint32_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (k0 + u * T < bins) ? nbits[k0 + u * T] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (k0 + u * T < bins) atomicAdd(&h2[2047u - (v[u] >> 20)], 1u);
                }
                __syncthreads();
                LT_STAMP("order: histogram");
                {
                    const uint32_t c0 = h2[2 * tid], c1 = h2[2 * tid + 1];
                    __syncthreads();
                    (void)block_excl_scan<W>(h2, 2048, wsum);  // h2[i] = bins with a digit above 2047 - i
                    const uint32_t a0 = h2[2 * tid], a1 = h2[2 * tid + 1];
                    if (a0 < kcap && a0 + c0 >= kcap) { bc[4] = 2 * tid; bc[5] = a0; bc[6] = c0; }
                    if (a1 < kcap && a1 + c1 >= kcap) { bc[4] = 2 * tid + 1; bc[5] = a1; bc[6] = c1; }
                    __syncthreads();
                }
                LT_STAMP("order: digit found");
                const uint32_t dstar = 2047u - bc[4], n_above = bc[5], n_cand = bc[6];
                constexpr uint32_t CAND0 = LKEYS_MAX / 2;
                __syncthreads();
                if (bc[4] != 0xFFFFFFFFu && n_above <= CAND0 && n_cand <= LKEYS_MAX - CAND0) {
                    const uint32_t lane = tid & 63u;
                    const uint64_t lt = (1ull << lane) - 1ull;
                    for (uint32_t kb = 0; kb < bins; kb += 4 * T) {
                    uint32_t vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) vv[u] = (kb + u * T + tid < bins) ? nbits[kb + u * T + tid] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k0 = kb + u * T;
                        if (k0 >= bins) break;
                        const uint32_t k = k0 + tid;
                        const uint32_t v = vv[u];
                        const uint32_t d = v >> 20;
                        const bool ab = k < bins && d > dstar, cd = k < bins && d == dstar;
                        const uint64_t key = ((uint64_t)(~v) << 32) | (uint64_t)k;
                        const uint64_t ma = __ballot(ab), mc = __ballot(cd);
                        if (ma) {
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(&bc[2], (uint32_t)__popcll(ma));
                            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                            if (ab) keys[base + (uint32_t)__popcll(ma & lt)] = key;
                        }
                        if (mc) {
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(&bc[3], (uint32_t)__popcll(mc));
                            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                            if (cd) keys[CAND0 + base + (uint32_t)__popcll(mc & lt)] = key;
                        }
                    }
                    }
                    __syncthreads();
                    LT_STAMP("order: collected");
                    uint32_t pa = 1, pc = 1;
                    while (pa < n_above) pa <<= 1;
                    while (pc < n_cand) pc <<= 1;
                    block_sort<W, true>(keys, nullptr, n_above, pa);
                    LT_STAMP("order: sorted above");
                    block_sort<W, true>(keys + CAND0, nullptr, n_cand, pc);
                    LT_STAMP("order: sorted candidates");
                    const uint32_t take = kcap - n_above;  // 1 .. n_cand
                    for (uint32_t i = tid; i < n_above; i += T) spos[i] = (uint32_t)(keys[i] & 0xffffffffu);
                    for (uint32_t i = tid; i < take; i += T) spos[n_above + i] = (uint32_t)(keys[CAND0 + i] & 0xffffffffu);
                    __syncthreads();
                    return kcap;
                }
                // radix select of the kcap-th largest norm-bit pattern, 8 bits per pass
                uint32_t prefix = 0, remaining = kcap;
                for (int shift = 24; shift >= 0; shift -= 8) {
                    for (uint32_t i = tid; i < 256; i += T) hist[i] = 0;
                    __syncthreads();
                    const uint32_t himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
                    for (uint32_t k = tid; k < bins; k += T) {
                        const uint32_t v = nbits[k];
                        if ((v & himask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        uint32_t acc = 0, b = 255;
                        for (;; --b) {
                            if (acc + hist[b] >= remaining || b == 0) break;
                            acc += hist[b];
                        }
                        bc[0] = b;
                        bc[1] = remaining - acc;
                    }
                    __syncthreads();
                    prefix |= bc[0] << shift;
                    remaining = bc[1];
                    __syncthreads();
                }
                const uint32_t thr = prefix;        // kcap-th largest value
                const uint32_t need_ties = remaining;  // how many bins equal to thr are admitted
                // bins above the threshold: any order (the sort fixes it)
                if (tid == 0) bc[2] = 0;
                __syncthreads();
                for (uint32_t k = tid; k < bins; k += T) {
                    const uint32_t v = nbits[k];
                    if (v > thr) {
                        const uint32_t slot = atomicAdd(&bc[2], 1u);
                        keys[slot] = ((uint64_t)(~v) << 32) | (uint64_t)k;
                    }
                }
                __syncthreads();
                const uint32_t above = bc[2];
                // ties: the first need_ties by ascending position (contiguous chunk per thread + scan)
                const uint32_t C = (bins + T - 1) / T;
                const uint32_t c0 = min(tid * C, bins), c1 = min(c0 + C, bins);
                uint32_t cnt = 0;
                for (uint32_t k = c0; k < c1; ++k) cnt += (nbits[k] == thr) ? 1u : 0u;
                // exclusive scan of cnt over threads
                uint32_t incl = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o);
                    if ((tid & 63) >= (uint32_t)o) incl += t;
                }
                if ((tid & 63) == 63) wsum[tid >> 6] = incl;
                __syncthreads();
                uint32_t base = 0;
                for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
                uint32_t rank = base + incl - cnt;
                __syncthreads();
                for (uint32_t k = c0; k < c1; ++k) {
                    if (nbits[k] == thr) {
                        if (rank < need_ties) keys[above + rank] = ((uint64_t)(~thr) << 32) | (uint64_t)k;
                        ++rank;
                    }
                }
                nkeys = above + need_ties;
                __syncthreads();
            }
            uint32_t p2 = 1;
            while (p2 < nkeys) p2 <<= 1;
            block_sort<W, true>(keys, nullptr, nkeys, p2);
            // the order goes to the workspace: the ladder's transforms take the LDS over
            for (uint32_t i = tid; i < nkeys; i += T) spos[i] = (uint32_t)(keys[i] & 0xffffffffu);
            __syncthreads();
            return nkeys;
            };
            // the first trip stores min(mf, Z) bins: if even that payload loses to a candidate that
            // already passes (the polynomial's first trip ran before this block), neither the order nor
            // the ladder is needed
            const uint32_t K1 = min(P.mf, Z);
            const bool fft_hopeless = prune && !can_win(1 + vlen(K1) + 9 * K1 + 8, 0);
            if (fft_hopeless) fft_pruned = true;
            uint32_t nkeys = PART == 2 ? fst->nkeys
                             : fft_hopeless ? 0u : build_order(min(kcap_total, max(2048u, P.mf + P.dk1)));
            if (prm.debug_stop == 5) return;
            LT_STAMP("order built");

            // ---- ladder ----
            const bool wraps = bins > 65536;
            const bool sparse = prm.sparse_inv != 0 && P.sp_mf != 0;
            uint32_t *own = aux;
            if (wraps && PART != 2) {
                for (uint32_t i = tid; i < 65536; i += T) own[i] = 0;
                __syncthreads();
            }
            const double mxd = (double)mxf, mnd = (double)mnf;
            const double Ld = (double)L;
            const float Lf = (float)L;
            uint32_t used = 0, jump = 0;
            double cur = prm.max_err + 1.0;
            bool resume = (PART == 2);  // PART 2 enters the loop where PART 1 left it: first trip admitted and bucketed
            if (PART == 2) { used = fst->used; fft_trips = 1; }
            auto sel_entry = [&](uint32_t i, uint32_t &p, float2 &x) -> bool {
                const Sel e = sel[i];
                p = e.pos;
                if (wraps) {
                    p &= 0xffffu;
                    if (own[p] != i + 1) return false;
                }
                x = (p == 0 || 2 * p == L) ? make_float2(e.re, 0.0f) : make_float2(e.re, e.im);
                return true;
            };
            while (!fft_hopeless && (prm.bounded ? (prm.max_err_m < sat_i32(cur * 1000.0)) : (fft_trips == 0))) {
                uint32_t K = used;
                if (!resume) {
                K = min(P.mf + jump, Z);
                // The ladder whose next payload is the smaller goes first.  A polynomial that is still climbing runs its
                // trips while they store less than this FFT trip would: once it passes, the FFT ladder is pruned by its
                // size -- left to itself it walks up to 23 trips (0.8 ms each at the far end) that cannot win (a frame
                // half ramp, half gauge at e = 1 %: 18.9 ms).  Pruning only skips candidates that cannot win: the
                // selection is unchanged.
                if (prune && fft_trips >= 1 && run_poly && poly_active && !idw) {
                    auto poly_next_lb = [&]() -> uint32_t {
                        const uint32_t base = (3 >= n / 100) ? 3 : n / 100;
                        const uint32_t pts = base + pjump;
                        const uint32_t step = max(n / pts, 1u);
                        const uint32_t cnt = (n + step - 1) / step;
                        const uint32_t Kp = cnt + (((cnt - 1) * step != n - 1) ? 1u : 0u);
                        return 2 + vlen(Kp) + Kp * (bitdepth == 0 ? 8u : 1u) + 17;
                    };
                    while (poly_active && poly_next_lb() <= 1 + vlen(K) + 9 * K + 8) poly_ladder(1);
                    if (!poly_active) poly_finish();
                }
                // (same prefix, longer: first the two trips most frames need, then five, then everything)
                if (K > nkeys && nkeys < min(kcap_total, bins))
                    nkeys = build_order(nkeys < 4096u ? min(kcap_total, max(4096u, P.mf + 4 * P.dk1)) : kcap_total);
                K = min(K, nkeys);
                if (prune && !can_win(1 + vlen(K) + 9 * K + 8, 0)) { fft_pruned = true; break; }
                ++fft_trips;
                // admit bins used..K-1 (fft.rs:401-422: bins 0 and L/2 are purely real for the
                // real output; an imaginary rounding residue there cannot reach idata[i].re)
                // `pos as u16` (fft.rs:242): in a 131072-sample frame bins >= 65536 are stored -- and
                // mirrored back by get_mirrored_freqs -- at pos - 65536, later entries overwriting
                // earlier ones.  own[] keeps, per stored position, the latest admission index.
                for (uint32_t i = used + tid; i < K; i += T) {
                    const uint32_t pos = spos[i];
                    const float2 z = spec[pos];
                    sel[i].pos = pos; sel[i].re = z.x; sel[i].im = z.y;
                    if (wraps) atomicMax(&own[pos & 0xffffu], i + 1);
                    else if (!sparse) Xs[pos] = (pos == 0 || 2 * pos == L) ? make_float2(z.x, 0.0f) : z;
                }
                __syncthreads();
                if (wraps && !sparse) {
                    for (uint32_t i = used + tid; i < K; i += T) {
                        const uint32_t p16 = sel[i].pos & 0xffffu;
                        if (own[p16] == i + 1)
                            Xs[p16] = (p16 == 0 || 2 * p16 == L) ? make_float2(sel[i].re, 0.0f)
                                                                 : make_float2(sel[i].re, sel[i].im);
                    }
                    __syncthreads();
                }
                used = K;
                if (!prm.bounded) { cur = 0.0; break; }
                LT_STAMP("bins admitted");
                }
                double s = 0.0;
                if (sparse && resume) {
                    // the tiles of this trip ran in k_large_trip_tiles: their sums, in tile order
                    if (tid == 0) {
                        const double *part = (const double *)((const unsigned char *)Cb + TRIP_PARTIAL_OFF);
                        for (uint32_t t = 0; t < (P.sp_md + SPB - 1) / SPB; ++t) s += part[t];
                    }
                } else if (sparse) {
                    if (PART == 1 && fft_trips == 1) {
                        // the cut: bucket the list, leave the bounds and the state, the tiles run elsewhere
                        sparse_bucket(P, K, sel_entry, (SpEnt *)work, tw, (unsigned char *)keys, wsum);
                        LT_STAMP("bucketed");
                        const SpLds sl = sp_lds(P, (unsigned char *)keys);
                        uint32_t *gb = (uint32_t *)((unsigned char *)Cb + TRIP_BOUNDS_OFF);
                        for (uint32_t e = tid; e < P.sp_mf; e += T) {
                            gb[e] = sl.beg[e];
                            gb[P.sp_mf + e] = sl.end[e];
                        }
                        if (tid == 0) {
                            TripState f;
                            f.smin = smin; f.smax = smax; f.poly_err = poly_err; f.pcur = pcur;
                            f.dg = dg;
                            f.finished = 0;
                            f.bitdepth = bitdepth; f.rle_size = rle_size; f.rle_R = rle_R; f.rle_D = rle_D;
                            f.rle_ib = rle_ib; f.rle_lb = rle_lb; f.best_size = best_size;
                            f.poly_step = poly_step; f.poly_K = poly_K; f.poly_size = poly_size;
                            f.poly_trips = poly_trips; f.pjump = pjump;
                            f.Z = Z; f.nkeys = nkeys; f.used = used;
                            f.mode = mode; f.best_owner = best_owner;
                            f.flags = (rle_sorted ? 1u : 0u) | (rle_pending ? 2u : 0u) | (poly_done ? 4u : 0u) |
                                      (poly_pruned ? 8u : 0u) | (poly_active ? 16u : 0u);
                            *fst = f;
                        }
                        return;
                    }
                    // evaluate: idata[j].re / L (f32), round 5, clamp, MAPE against the padded signal
                    sparse_inverse(
                        P, K, sel_entry,
                        (SpEnt *)work, tw, (unsigned char *)keys, wsum,
                        [&](uint32_t j) -> double { return gpad(j); },
                        [&](uint32_t, float re, double g) {
                            const double v = (double)(re / Lf);
                            double o = div1e5(round(v * 100000.0));
                            if (o > mxd) o = mxd;
                            if (o < mnd) o = mnd;
                            s += fabs(o - g) * recip_abs(g);  // the product form, as in k_compress
                        },
                        prm.debug_stop >= 16 ? (uint32_t)prm.debug_stop - 16u : 0u);
                } else {
                float2 *F;
                if (P.half) {
                    // Hermitian spectrum -> packed complex spectrum of (even + i odd) samples, conjugated
                    // so that the forward butterflies deliver the inverse transform:
                    //   E[k] = (X[k] + conj X[M-k]) / 2,  O[k] = (X[k] - conj X[M-k]) / 2 * conj(w^k),
                    //   Zk = E + i O ;   idft_M(Zk) = conj(dft_M(conj Zk))
                    for (uint32_t k = tid; k < M; k += T) {
                        const float2 xk = Xs[k], xm = Xs[M - k];
                        const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
                        const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
                        const float2 o = cmulp(d, tw[k]);
                        // E + iO = (e.x - o.y) + i (e.y + o.x) ; store the conjugate
                        work[k] = make_float2(e.x - o.y, -(e.y + o.x));
                    }
                    __syncthreads();
                    F = fft_large(P, work, Cb, tw, fft_lds, prm.large_tiled != 0);
                } else {
                    for (uint32_t k = tid; k < L; k += T) {
                        float2 v = make_float2(0.0f, 0.0f);
                        if (k <= L / 2) v = Xs[k];
                        else { const float2 c = Xs[L - k]; v = make_float2(c.x, -c.y); }
                        work[k] = make_float2(v.x, -v.y);  // conj for inverse-by-forward
                    }
                    __syncthreads();
                    F = fft_large(P, work, Cb, tw, fft_lds, prm.large_tiled != 0);
                }
                // evaluate: idata[j].re / L (f32), round 5, clamp, MAPE against the padded signal
                for (uint32_t j = tid; j < L; j += T) {
                    float re;
                    if (P.half) {
                        const float2 f = F[j >> 1];
                        re = 2.0f * ((j & 1) ? -f.y : f.x);  // idft_L = 2 * idft_M (even -> re, odd -> im)
                    } else {
                        re = F[j].x;
                    }
                    const double v = (double)(re / Lf);
                    double o = div1e5(round(v * 100000.0));
                    if (o > mxd) o = mxd;
                    if (o < mnd) o = mnd;
                    const double g = gpad(j);
                    s += fabs((o - g) / g);
                }
                }
                resume = false;
                s = block_sum_f64<W>(s, red, parity);
                cur = s / Ld;
                if (fft_trips <= 17) jump += P.dk1;
                else if (fft_trips <= 22) jump += P.dk2;
                else break;
            }
            fft_err = cur;
            fft_k = used;
            uint32_t big = 0;
            for (uint32_t i = tid; i < used; i += T) big += ((sel[i].pos & 0xffffu) >= 251) ? 1u : 0u;
            big = block_sum_u32<W>(big, red, parity);
            fft_size = 1 + vlen(used) + 9 * used + 2 * big + 8;
            fft_done = !fft_pruned;
            __syncthreads();
        }
        if (prune && fft_done && fft_err <= me) offer(fft_size, 0);
        dg.fft_size = fft_size; dg.fft_trips = (uint16_t)fft_trips; dg.fft_k = (uint16_t)fft_k;
        dg.fft_err = fft_err;
    }

    LT_STAMP("fft ladder done");
    if (prm.debug_stop == 6 || prm.debug_stop >= 16) return;
    // the polynomial ladder continues where its first trip stopped (see above the FFT block)
    if (run_poly && poly_active) {
        poly_ladder(0xFFFFFFFFu);
        poly_finish();
    }

    LT_STAMP("poly ladder done");
    if (prm.debug_stop == 7) return;
    // ---- RLE with many runs: exact size (hash of run values) only if its bound can still win ----
    if (run_rle && rle_pending) {
        if (!prune || can_win(rle_lb, 2)) {
            const uint32_t H = 2 * n;
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
            rle_D = block_sum_u32<W>(dnew, red, parity);
            hb = block_sum_u32<W>(hb, red, parity);
            rle_size = 2 + vlen(rle_D) + hb + rle_ib;
            if (prune) offer(rle_size, 2);
        } else {
            rle_size = rle_lb;
        }
    }
    dg.rle_size = rle_size;

    // ---- selection (frame/mod.rs:113-147) ----
    int chosen;
    double chosen_err;
    if (mode == ATSC_AUTO) {
        if (prune) {
            chosen = best_owner == 0 ? ATSC_FFT : best_owner == 1 ? ATSC_POLYNOMIAL : ATSC_RLE;
        } else {  // max_error < 0 or NaN: nothing passes, smallest of all (frame/mod.rs:128-135)
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

    uint32_t out_len = 0;
    if (prm.trial) {
        if (tid == 0) {
            res[fid].err = chosen_err;
            res[fid].len = 0;
            res[fid].chosen = (uint32_t)chosen;
        }
        return;
    }
    if (chosen == ATSC_FFT) {
        const uint32_t hdr = 1 + vlen(fft_k);
        for (uint32_t i = tid; i < fft_k; i += T) aux[i] = vlen(sel[i].pos & 0xffffu) + 8;
        __syncthreads();
        const uint32_t body = lscan(aux, fft_k, wsum);
        for (uint32_t i = tid; i < fft_k; i += T) {
            uint8_t *p = out + hdr + aux[i];
            p += put_varint(p, sel[i].pos & 0xffffu);
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
    } else if (chosen == ATSC_POLYNOMIAL || chosen == ATSC_IDW) {
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
            body = lscan(aux, poly_K, wsum);
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
            out[hdr + body + 16] = (uint8_t)poly_step;
        }
        out_len = hdr + body + 17;
    } else {
        if (!rle_sorted) rle_sort_and_group();
        const uint32_t R = rle_R, D = rle_D;
        const uint32_t hdr = 2 + vlen(D);
        for (uint32_t i = tid; i < R; i += T) rps[i] = vlen(rrec[i] >> 32);
        __syncthreads();
        lscan(rps, R, wsum);
        const uint32_t hb = lscan(rph, D, wsum);
        uint32_t ibt = 0;
        for (uint32_t i = tid; i < R; i += T) {
            const uint64_t rec = rrec[i];
            const uint32_t st = (uint32_t)(rec >> 32);
            const bool head = (i == 0 || run_key(rec) != run_key(rrec[i - 1]));
            const uint32_t gi = head ? aux[i] : aux[i] - 1;
            const uint32_t ghb = (gi + 1 < D ? rph[gi + 1] : hb);
            if (head) {
                uint8_t *p = out + hdr + rph[gi] + rps[i];
                p += put_value(p, bitdepth, xs[(uint32_t)rec]);
                put_varint(p, rhp[gi + 1] - rhp[gi]);
            }
            put_varint(out + hdr + ghb + rps[i], st);
            if (i == R - 1) ibt = rps[i] + vlen(st);
        }
        ibt = block_sum_u32<W>(ibt, red, parity);
        if (tid == 0) {
            out[0] = 60;
            out[1] = (uint8_t)bitdepth;
            put_varint(out + 2, D);
        }
        out_len = hdr + hb + ibt;
    }
    if (tid == 0) {
        res[fid].err = chosen_err;
        res[fid].len = out_len;
        res[fid].chosen = (uint32_t)chosen;
        if (diag) diag[fid] = dg;
    }
    LT_STAMP("emitted");
}

// A workgroup per frame of the launch (slot = blockIdx.x).  Behind the grid path (prm.fast_skip == 2) the launch is as wide
// as the GPU part at most and its workgroups share out the frames k_large_decide1 / decide2 listed (fb_append): a launch of
// 1280 workgroups of 1024 threads whose frames are all decided -- the usual case -- took 41 us to start and end them.
template <int PART>
__global__ __launch_bounds__(LT) void k_compress_large(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames,
    const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, const KParams prm, uint8_t *__restrict__ slots,
    DevResult *__restrict__ res, atsc_frame_diag *__restrict__ diag, unsigned char *__restrict__ ws_base,
    uint64_t ws_stride)
{
    const bool listed = PART == 0 && prm.fast_skip == 2;
    const uint32_t count = listed ? *fb_count(ws_base, ws_stride) : gridDim.x;
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {  // (not listed: one turn, it = blockIdx.x)
        const uint32_t bx = listed ? *fb_entry(ws_base, ws_stride, it) : it;
        compress_large_frame<PART>(bx, samples, frames, ids, plans, twpool, prm, slots, res, diag, ws_base, ws_stride);
        __syncthreads();  // the next frame's first LDS writes stay behind this frame's last reads
    }
}

// One tile (SPB output columns) of the first FFT trip of one frame whose k_compress_large<1> stopped at the cut:
// the tile's share of the MAPE sum goes to the frame's workspace (buffer C, TRIP_PARTIAL_OFF).
__global__ __launch_bounds__(LT) void k_large_trip_tiles(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames,
    const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, unsigned char *__restrict__ ws_base, uint64_t ws_stride, uint32_t nframes)
{
    // grid (frames padded to a multiple of 8, tiles): workgroups are dealt round-robin over the 8 XCDs by
    // their linear index, so the tiles of a frame -- which share its list, twiddles and samples -- land on one
    // XCD and its L2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t fidx = blockIdx.x;
    if (fidx >= nframes) return;
    const DevFrame fr = frames[ids[fidx]];
    const DevPlan &P = plans[fr.plan];
    const uint32_t jb0 = blockIdx.y * SPB;
    if (!P.sp_mf || jb0 >= P.sp_md) return;
    unsigned char *ws = ws_base + (uint64_t)fidx * ws_stride;
    const LargeWs lay = large_ws_layout(P.n, P.L, P.kcap);
    const TripState *fst = (const TripState *)(ws + lay.o_front);
    if (fst->finished) return;
    const uint32_t n = P.n, L = P.L, pre = P.pre, Mf = P.sp_mf, Md = P.sp_md;
    const float2 *tw = twpool + P.tw_off;
    const double *xs = samples + fr.sample_off;
    const SpLds sl = sp_lds(P, smem);
    double *red = (double *)(smem + SP_LDS_BYTES);
    unsigned char *Cb = ws + lay.o_c;
    const uint32_t *gb = (const uint32_t *)(Cb + TRIP_BOUNDS_OFF);
    for (uint32_t e = tid; e < Mf; e += LT) {
        sl.wf[e] = tw[e * (L / Mf)];
        sl.beg[e] = gb[e];
        sl.end[e] = gb[Mf + e];
    }
    for (uint32_t e = tid; e < Md; e += LT) sl.wd[e] = tw[e * (L / Md)];
    __syncthreads();
    // the list sits in the FFT buffer that does not hold the spectrum (see k_compress_large: spec / work)
    const SpEnt *zl = (const SpEnt *)(ws + lay.o_b);
    const double mxd = (double)(float)fst->smax, mnd = (double)(float)fst->smin;
    const float Lf = (float)L;
    double s = 0.0;
    sparse_tile(
        P, jb0, zl, tw, smem,
        [&](uint32_t j) -> double {  // fft.rs:184-204
            int32_t i = (int32_t)j - (int32_t)pre;
            i = i < 0 ? 0 : (i >= (int32_t)n ? (int32_t)n - 1 : i);
            return xs[i];
        },
        [&](uint32_t, float re, double g) {
            const double v = (double)(re / Lf);
            double o = div1e5(round(v * 100000.0));
            if (o > mxd) o = mxd;
            if (o < mnd) o = mnd;
            s += fabs(o - g) * recip_abs(g);
        });
    int parity = 0;
    s = block_sum_f64<LW>(s, red, parity);
    if (tid == 0) ((double *)(Cb + TRIP_PARTIAL_OFF))[blockIdx.y] = s;
}

constexpr uint32_t STG_BYTES = 16384;  // payload window of the large decoder (RdS, atsc_device.h)
#include "atsc_large_fast.h"

// k_large_trip243 for every row dimension present among the launch's frames (LargePre::rows9p holds the P = 2^LG values)
template <bool DECODE, class FR>
static hipError_t launch_trip243(uint32_t pmask, uint32_t tiles, uint32_t nb, hipStream_t s, const double *samples, const FR *frames,
                                 const uint32_t *ids, const DevPlan *plans, const float2 *twpool, unsigned char *ws,
                                 uint64_t ws_stride, int dbg, double *out)
{
    static const char *wm_env = getenv("ATSC_TRIP_WIDE_MAX");  // (A/B aid: tools/trip_width_ab.sh)
    const uint32_t wide_max = wm_env ? (uint32_t)atoi(wm_env) : (DECODE ? TRIP_WIDE_MAX_DEC : TRIP_WIDE_MAX_ENC);
#define ATSC_TRIP_T(LG_, TT_)                                                                                                 \
    {                                                                                                                          \
        hipError_t e = ensure_dyn_lds((const void *)k_large_trip243<DECODE, FR, LG_, TT_>, FAST_TILE_LDS);                     \
        if (e != hipSuccess) return e;                                                                                         \
        hipLaunchKernelGGL((k_large_trip243<DECODE, FR, LG_, TT_>), dim3(gx, nb), dim3(TT_), FAST_TILE_LDS, s, samples, frames, \
                           ids, plans, twpool, ws, ws_stride, dbg, out);                                                       \
    }
#define ATSC_TRIP(LG_)                                                                                                        \
    if (pmask & (1u << LG_)) {                                                                                                 \
        const uint32_t gx = min(tiles, (9u << LG_) / 16u + (((9u << LG_) & 15u) ? 1u : 0u));                                    \
        if (gx * nb <= wide_max) ATSC_TRIP_T(LG_, TRIP_WIDE) else ATSC_TRIP_T(LG_, CT)                                         \
        dbg &= ~4;  /* (the list of the frames left alone is closed by the first launch) */                                    \
    }
    ATSC_TRIP(5) ATSC_TRIP(4) ATSC_TRIP(3) ATSC_TRIP(2) ATSC_TRIP(1)
#undef ATSC_TRIP
#undef ATSC_TRIP_T
    return hipSuccess;
}

// The row pass of every P present among the launch's frames (LargePre::rows9p): one launch per P, whose workgroups
// leave the frames of another P alone; the polynomial pieces of a frame (extra workgroups behind the row tiles) ride on
// the launch of the frame's own P.
static void launch_rows9p(uint32_t pmask, uint32_t row_tiles, uint32_t pieces, uint32_t nb, hipStream_t s, const double *samples,
                          const DevFrame *frames, const uint32_t *ids, const DevPlan *plans, const float2 *twpool,
                          unsigned char *ws, uint64_t ws_stride, int sparse_inv)
{
    const dim3 g(row_tiles + pieces, nb);
    if (pmask & 32u) hipLaunchKernelGGL(k_large_rows9p<32>, g, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv, row_tiles);
    if (pmask & 16u) hipLaunchKernelGGL(k_large_rows9p<16>, g, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv, row_tiles);
    if (pmask & 8u) hipLaunchKernelGGL(k_large_rows9p<8>, g, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv, row_tiles);
    // rows of 18 points: one thread per row, one workgroup per frame (k_large_rows_thread)
    static const bool old_rows = getenv("ATSC_LARGE_ROWS9P_ONLY") != nullptr;
    const dim3 gt(1 + pieces, nb);
    // (36-point rows: 166 VGPRs and 70 KB of LDS leave two such workgroups on a CU: 16384-sample frames 46 -> 42 Gsamples/s)
    static const bool thread4 = getenv("ATSC_LARGE_ROWS_THREAD4") != nullptr;
    if (pmask & 4u) {
        if (!thread4) hipLaunchKernelGGL(k_large_rows9p<4>, g, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv, row_tiles);
        else hipLaunchKernelGGL(k_large_rows_thread<4>, gt, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv);
    }
    if (pmask & 2u) {
        if (old_rows) hipLaunchKernelGGL(k_large_rows9p<2>, g, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv, row_tiles);
        else hipLaunchKernelGGL(k_large_rows_thread<2>, gt, dim3(RT), 0, s, samples, frames, ids, plans, twpool, ws, ws_stride, sparse_inv);
    }
}

hipError_t launch_compress_large(uint32_t count, const double *samples, const DevFrame *frames,
                                 const uint32_t *ids, const DevPlan *plans, const float2 *twpool,
                                 const KParams &prm, uint8_t *slots, DevResult *res, atsc_frame_diag *diag,
                                 unsigned char *ws, uint64_t ws_stride, uint32_t ws_slots, hipStream_t s,
                                 const LargePre *pre)
{
    // (The caller deals the large frames of a batch over several streams, a contiguous group of frames and workspace
    // slots each: the chain below is bound by latency -- a dependent launch starts 6-10 us after its predecessor ends on
    // this system, tools/gap_probe.hip, and several links run one workgroup per frame -- so the groups' chains overlap.)
    const uint32_t lds = 384 + 1024 + 64 + max(8 * LKEYS_MAX, SP_LDS_BYTES);
    hipError_t e = ensure_dyn_lds((const void *)k_compress_large<0>, lds);
    if (e != hipSuccess) return e;
    KParams kp = prm;
    kp.prefft = (pre && pre->tiles1) ? 1u : 0u;
    // Few large frames: each has a CU to itself and most CUs idle, so the tiles of the first FFT trip run as
    // a (tile, frame) grid between the two parts of the per-frame kernel.
    const bool split = kp.prefft && kp.sparse_inv && kp.bounded && kp.debug_stop <= 0 && !kp.trial && pre->sp_tiles &&
                       count <= LARGE_SPLIT_MAX;
    const uint32_t lds_tiles = SP_LDS_BYTES + 512;
    kp.prestats = (split && pre->chunks_n) ? 1u : 0u;
    if (split) {
        e = ensure_dyn_lds((const void *)k_compress_large<1>, lds);
        if (e != hipSuccess) return e;
        e = ensure_dyn_lds((const void *)k_compress_large<2>, lds);
        if (e != hipSuccess) return e;
        e = ensure_dyn_lds((const void *)k_large_trip_tiles, lds_tiles);
        if (e != hipSuccess) return e;
    }
    // tile buffers of a pre-pass workgroup: two of FB x (sub-transform length [+ 1]) points + its twiddles
    uint32_t lds1 = 0, lds2 = 0;
    if (kp.prefft) {
        lds1 = (2 * FB * pre->m1_max + pre->m1_max) * (uint32_t)sizeof(float2);
        lds2 = (2 * (FB + 1) * pre->m2_max + pre->m2_max) * (uint32_t)sizeof(float2);
        e = ensure_dyn_lds((const void *)k_large_pre1<DevFrame, false>, lds1);
        if (e != hipSuccess) return e;
        e = ensure_dyn_lds((const void *)k_large_pre23, lds2);
        if (e != hipSuccess) return e;
    }
    // The fast path (atsc_large_fast.h): frames of 131072 samples under the auto selector, any number of them.
    const bool no_fast = getenv("ATSC_LARGE_NO_FAST") != nullptr;  // (read per launch: the tests switch it)
    const bool fast = !no_fast && kp.prefft && pre->cols243 && pre->rows9p != 0 && pre->chunks_n && kp.sparse_inv &&
                      kp.bounded && (kp.mode == ATSC_AUTO || kp.mode == ATSC_FFT || kp.mode == ATSC_POLYNOMIAL) && !kp.trial &&
                      kp.trial_res == nullptr && diag == nullptr &&
                      (kp.debug_stop == 0 || kp.debug_stop == -3 || kp.debug_stop == -4 || kp.debug_stop == -6) && 0.0 <= kp.max_err;
    if (fast) {
        e = ensure_dyn_lds(pre->m2_max >= FAST_MD ? (const void *)k_large_decide1_big : (const void *)k_large_decide1,
                           fast_d1_lds(fast_carve(pre->m2_max)));
        if (e != hipSuccess) return e;
        e = ensure_dyn_lds((const void *)k_large_decide2, FAST_D2_LDS);
        if (e != hipSuccess) return e;
        kp.prestats = 1;
        kp.fast_skip = 2;
    }
    for (uint32_t b0 = 0; b0 < count; b0 += ws_slots) {
        const uint32_t nb = min(ws_slots, count - b0);
        if (fast) {
            const uint32_t tiles23 = 1 + ((pre->m1_max - 1) / 2 + FBH - 1) / FBH;
            // the statistics ride on the column pass when every frame can be read as aligned pairs
            const bool tst = pre->even_off && (((uintptr_t)samples & 15u) == 0) && pre->tiles1 <= TST_MAX;
            if (tst) {
                hipLaunchKernelGGL(k_large_cols243<true>, dim3(pre->tiles1, nb), dim3(CT), 0, s, samples, frames, ids + b0,
                                   plans, twpool, ws, ws_stride);
            } else {
                hipLaunchKernelGGL(k_large_stats0, dim3(nb), dim3(64), 0, s, frames, ids + b0, plans, ws, ws_stride);
                hipLaunchKernelGGL(k_large_stats, dim3(pre->chunks_n, nb), dim3(LT), 0, s, samples, frames, ids + b0, plans,
                                   ws, ws_stride);
                hipLaunchKernelGGL(k_large_cols243<false>, dim3(pre->tiles1, nb), dim3(CT), 0, s, samples, frames, ids + b0,
                                   plans, twpool, ws, ws_stride);
            }
            kp.tile_stats = tst ? 1u : 0u;
            if (tst) {  // the row pass and the first polynomial trip's pieces in one launch
                // (bit 1: the column tiles' statistics are there -- a Constant frame's workgroups leave at once)
                launch_rows9p(pre->rows9p, tiles23, pre->chunks_n * (LCH / PCH), nb, s, samples, frames, ids + b0, plans, twpool,
                              ws, ws_stride, (int)(kp.sparse_inv ? 1 : 0) | 2);
            } else {
                hipLaunchKernelGGL(k_large_poly1, dim3(pre->chunks_n, nb), dim3(LT), 0, s, samples, frames, ids + b0, plans,
                                   ws, ws_stride, 0);
                launch_rows9p(pre->rows9p, tiles23, 0, nb, s, samples, frames, ids + b0, plans, twpool, ws, ws_stride,
                              (int)kp.sparse_inv);
            }
            const FastCarve cv = fast_carve(pre->m2_max);
            if (pre->m2_max >= FAST_MD)
                hipLaunchKernelGGL(k_large_decide1_big, dim3(nb), dim3(LT), fast_d1_lds(cv), s, samples, frames, ids + b0,
                                   plans, twpool, kp, slots, res, ws, ws_stride, cv);
            else
                hipLaunchKernelGGL(k_large_decide1, dim3(nb), dim3(LT), fast_d1_lds(cv), s, samples, frames, ids + b0, plans,
                                   twpool, kp, slots, res, ws, ws_stride, cv);
            e = launch_trip243<false, DevFrame>(pre->rows9p, (pre->m2_max + 15) / 16, nb, s, samples, frames, ids + b0, plans, twpool,
                                                ws, ws_stride, kp.debug_stop == -6 ? 2 : kp.debug_stop <= -3 ? 1 : 0, (double *)nullptr);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_large_decide2, dim3(nb), dim3(LT), FAST_D2_LDS, s, samples, frames, ids + b0, plans, kp,
                               slots, res, ws, ws_stride);
            // whatever those left undecided (FastState::status != 2: the list k_large_decide2 closed)
            hipLaunchKernelGGL(k_compress_large<0>, dim3(min(nb, FB_GRID_MAX)), dim3(LT), lds, s, samples, frames, ids + b0, plans,
                               twpool, kp, slots, res, diag, ws, ws_stride);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            continue;
        }
        if (split) {
            // statistics and first polynomial trip as (chunk, frame) grids; k_large_pre1 then resets the count of
            // zero bins only
            hipLaunchKernelGGL(k_large_stats0, dim3(nb), dim3(64), 0, s, frames, ids + b0, plans, ws, ws_stride);
            hipLaunchKernelGGL(k_large_stats, dim3(pre->chunks_n, nb), dim3(LT), 0, s, samples, frames, ids + b0, plans,
                               ws, ws_stride);
            if (kp.mode == ATSC_AUTO)
                hipLaunchKernelGGL(k_large_poly1, dim3(pre->chunks_n, nb), dim3(LT), 0, s, samples, frames, ids + b0,
                                   plans, ws, ws_stride, 0);
        }
        if (kp.prefft) {
            if (pre->cols243)
                hipLaunchKernelGGL(k_large_cols243<false>, dim3(pre->tiles1, nb), dim3(CT), 0, s, samples, frames, ids + b0,
                                   plans, twpool, ws, ws_stride);
            else
                hipLaunchKernelGGL((k_large_pre1<DevFrame, false>), dim3(pre->tiles1, nb), dim3(PT), lds1, s, samples,
                                   frames, ids + b0, plans, twpool, ws, ws_stride);
            const uint32_t tiles23 = 1 + ((pre->m1_max - 1) / 2 + FBH - 1) / FBH;
            if (pre->cols243 && pre->rows9p != 0)
                launch_rows9p(pre->rows9p, tiles23, 0, nb, s, samples, frames, ids + b0, plans, twpool, ws, ws_stride,
                              (int)kp.sparse_inv);
            else
                hipLaunchKernelGGL(k_large_pre23, dim3(tiles23, nb), dim3(PT), lds2, s, samples, frames, ids + b0, plans,
                                   twpool, ws, ws_stride, (int)kp.sparse_inv);
        }
        if (split) {
            hipStream_t sf = s;
            hipLaunchKernelGGL(k_compress_large<1>, dim3(nb), dim3(LT), lds, sf, samples, frames, ids + b0, plans,
                               twpool, kp, slots, res, diag, ws, ws_stride);
            hipLaunchKernelGGL(k_large_trip_tiles, dim3((nb + 7u) & ~7u, pre->sp_tiles), dim3(LT), lds_tiles, sf, samples,
                               frames, ids + b0, plans, twpool, ws, ws_stride, nb);
            hipLaunchKernelGGL(k_compress_large<2>, dim3(nb), dim3(LT), lds, sf, samples, frames, ids + b0, plans,
                               twpool, kp, slots, res, diag, ws, ws_stride);
        } else {
            hipLaunchKernelGGL(k_compress_large<0>, dim3(nb), dim3(LT), lds, s, samples, frames, ids + b0, plans,
                               twpool, kp, slots, res, diag, ws, ws_stride);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}


// --------------------------------------------------------------------------------------------
// k_decompress_large: CompressorFrame::decompress for frames of 4097 .. 131072 samples
// --------------------------------------------------------------------------------------------

// PH 0: the whole decoder.  PH 1 / PH 2: the decoder around the batched inverse transform -- PH 1 parses,
// decodes every codec but FFT completely and leaves an FFT frame's conjugated packed spectrum in buffer A
// (DecPending in the workspace); k_large_pre1 / k_large_pre2 <DevDFrame, true> transform all pending frames
// over the whole GPU; PH 2 scales, rounds and clamps.
template <int PH>
__device__ __forceinline__ void decompress_large_frame(
    const uint32_t bx, const DevDFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool,
    const uint8_t *__restrict__ body, double *__restrict__ outp, int *__restrict__ status,
    unsigned char *__restrict__ ws_base, uint64_t ws_stride, int tiled, int sparse, int sp_split, int fast_skip)
{
    constexpr int T = LT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const DevDFrame fr = frames[ids[bx]];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = fr.n, L = P.L, M = P.M, pre = P.pre;
    double *out = outp + fr.out_off;
    const uint8_t *pay = body + fr.payload_off;

    struct Hdr {
        double d0, d1;
        uint32_t u0, u1, u2, bad;
        float f0, f1;
    };
    Hdr *hdr = (Hdr *)smem;
    uint32_t *wsum = (uint32_t *)(smem + 64);

    unsigned char *ws = ws_base + (uint64_t)bx * ws_stride;
    const LargeWs lay = large_ws_layout(n, L, P.kcap);
    float2 *A = (float2 *)(ws + lay.o_a);
    float2 *Cb = (float2 *)(ws + lay.o_c);
    float2 *Xs = (float2 *)(ws + lay.o_x);
    double *vals = (double *)(ws + lay.o_tab);    // knot values / RLE group values (8n bytes)
    uint64_t *keys = (uint64_t *)(ws + lay.o_rec);  // RLE (start << 32 | group)
    Sel *ent = (Sel *)(ws + lay.o_sel);           // FFT entries parsed in parallel (<= kcap of them)
    uint32_t *own = (uint32_t *)(ws + lay.o_aux);  // per position: 1 + index of the last entry that names it
    const float2 *tw = twpool + P.tw_off;
    (void)wsum;
    DecPending *pend = (DecPending *)(ws + lay.o_cnt);
    if (PH == 2) {  // second half: only frames whose transform was pending
        if (fr.tag != ATSC_FFT || pend->pending == 0) return;
        const float mxf = pend->mxf, mnf = pend->mnf;
        const float2 *F = A;
        const double mxd = (double)mxf, mnd = (double)mnf;
        const float Lf = (float)L;
        for (uint32_t i = tid; i < n; i += T) {
            const uint32_t j = i + pre;
            float re;
            if (P.half) {
                const float2 f = F[j >> 1];
                re = 2.0f * ((j & 1) ? -f.y : f.x);
            } else {
                re = F[j].x;
            }
            const float v = re / Lf;
            double o = round((double)v * 100000.0) / 100000.0;
            if (o > mxd) o = mxd;
            if (o < mnd) o = mnd;
            out[i] = o;
        }
        return;
    }
    if ((PH == 1 || (PH == 0 && sp_split)) && tid == 0) pend->pending = 0;
    // behind the decoder's fast path (k_large_dparse + k_large_trip243<true>): only the frames it left alone
    if (PH == 0 && fast_skip && ((const FastState *)(ws + lay.o_front))->status != 0) return;

    // fixed-width point arrays (U8 / F64) are read in parallel after the header; the spectrum must be
    // empty before lane 0 starts filling it
    if (fr.tag == ATSC_FFT && !(PH == 0 && sparse && P.sp_mf)) {
        for (uint32_t k = tid; k <= L / 2; k += T) Xs[k] = make_float2(0.0f, 0.0f);
    }
    __syncthreads();

    if (tid < 64) {  // the first wavefront walks the payload in lock step: every lane writes the same values
        RdS r{pay, fr.payload_len, 0, false, smem + 256, 0, 0, STG_BYTES};
        Hdr h;
        h.d0 = h.d1 = 0.0; h.u0 = h.u1 = h.u2 = 0; h.f0 = h.f1 = 0.0f;
        switch (fr.tag) {
        case ATSC_CONSTANT: {
            (void)rds_u8(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            if (bd > 3) r.bad = true;
            else h.d0 = rds_value(r, bd);
            break;
        }
        case ATSC_NOOP: {
            (void)rds_u8(r);
            const uint64_t cnt = rds_varint(r);
            if (cnt != n) r.bad = true;
            if (!r.bad) rds_varints(r, n, [&](uint32_t i, uint64_t v) { out[i] = (double)unzig(v); });
            break;
        }
        case ATSC_IDW:
        case ATSC_POLYNOMIAL: {
            const uint32_t id = (uint32_t)rds_varint(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t cnt = rds_varint(r);
            if (id > 1 || bd > 3 || cnt > n) r.bad = true;
            h.f1 = (float)id;  // 0 Polynomial, 1 Idw
            h.u2 = bd;
            h.u0 = (uint32_t)cnt;
            if (!r.bad) {
                if (bd == 0 || bd == 3) {
                    h.u1 = r.pos;  // fixed-width points start here; read in parallel below
                    r.pos += (uint32_t)cnt * (bd == 0 ? 8u : 1u);
                    if (r.pos > r.len) r.bad = true;
                } else {
                    rds_varints(r, (uint32_t)cnt, [&](uint32_t i, uint64_t v) {
                        vals[i] = (bd == 2) ? (double)(int16_t)unzig(v) : (double)(int32_t)unzig(v);
                    });
                }
            }
            h.d0 = __longlong_as_double((long long)rds_le(r, 8));
            h.d1 = __longlong_as_double((long long)rds_le(r, 8));
            h.f0 = (float)rds_u8(r);  // point_step
            break;
        }
        case ATSC_FFT: {
            (void)rds_u8(r);
            const uint64_t cnt = rds_varint(r);
            if (cnt > L) r.bad = true;
            if (!r.bad && cnt <= P.kcap) {
                rds_fft_entries(r, (uint32_t)cnt, L, ent);  // 64 at a time; the whole workgroup applies them below
                h.u1 = 1;  // entries wait in the workspace
            } else
            // (a foreign stream with more entries than this library's encoder ever stores: one at a time)
            // get_mirrored_freqs (fft.rs:401-422): entries are applied in stream order, later ones
            // overwrite; a position above L/2 is the mirror of L - pos
            {
            if (PH == 0 && sparse && P.sp_mf)  // nobody cleared the dense spectrum for this frame yet
                for (uint32_t k = tid; k <= L / 2; k += 64) Xs[k] = make_float2(0.0f, 0.0f);
            for (uint32_t i = 0; i < cnt && !r.bad; ++i) {
                uint32_t pos = (uint32_t)rds_varint(r) & 0xffffu;
                float re = rds_f32(r), im = rds_f32(r);
                if (pos >= L) { r.bad = true; break; }
                if (pos > L / 2) { pos = L - pos; im = -im; }
                Xs[pos] = (pos == 0 || 2 * pos == L) ? make_float2(re, 0.0f) : make_float2(re, im);
            }
            }
            h.u0 = (uint32_t)cnt;
            h.f0 = rds_f32(r);
            h.f1 = rds_f32(r);
            break;
        }
        case ATSC_RLE: {
            (void)rds_u8(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t groups = rds_varint(r);
            if (bd > 3 || groups > n) r.bad = true;
            uint32_t e = 0;
            for (uint32_t gi = 0; gi < groups && !r.bad; ++gi) {
                vals[gi] = rds_value(r, bd);
                const uint64_t cnt = rds_varint(r);
                if (cnt > n - e) { r.bad = true; break; }
                for (uint32_t k = 0; k < cnt && !r.bad; ++k) {
                    const uint64_t idx = rds_varint(r);
                    if (idx >= n) { r.bad = true; break; }
                    keys[e++] = (idx << 32) | gi;
                }
            }
            h.u0 = e;
            h.u1 = (uint32_t)groups;
            break;
        }
        default: r.bad = true;
        }
        h.bad = r.bad ? 1u : 0u;
        *hdr = h;
        if (r.bad && tid == 0) atomicExch(status, 1);
    }
    __syncthreads();
    const Hdr h = *hdr;
    if (h.bad) return;

    if (fr.tag == ATSC_CONSTANT) {
        for (uint32_t j = tid; j < n; j += T) out[j] = h.d0;
        return;
    }
    if (fr.tag == ATSC_NOOP) return;
    if (fr.tag == ATSC_POLYNOMIAL || fr.tag == ATSC_IDW) {
        const double mn = h.d0, mx = h.d1;
        if (mx == mn) {
            for (uint32_t j = tid; j < n; j += T) out[j] = mx;
            return;
        }
        const uint32_t step = (uint32_t)h.f0, K = h.u0, bd = h.u2;
        bool ok = step >= 1;
        if (ok) {
            const uint32_t cnt = (n + step - 1) / step;
            const uint32_t Kp = cnt + (((cnt - 1) * step != n - 1) ? 1u : 0u);
            ok = (Kp == K) && K >= 2;
        }
        if (!ok) {
            if (tid == 0) atomicExch(status, 1);
            return;
        }
        if (bd == 0 || bd == 3) {
            const uint8_t *pp = pay + h.u1;
            for (uint32_t k = tid; k < K; k += T) {
                if (bd == 3) {
                    vals[k] = (double)pp[k];
                } else {
                    uint64_t v = 0;
                    for (int b = 0; b < 8; ++b) v |= (uint64_t)pp[8 * k + b] << (8 * b);
                    vals[k] = __longlong_as_double((long long)v);
                }
            }
            __syncthreads();
        }
        const uint32_t magic = (uint32_t)(0x100000000ull / step) + 1u;
        const bool idw = (h.f1 != 0.0f);  // polynomial.rs:395-404 picks the interpolation by the id
        double *wtab = (double *)A;       // w[d] = 1 / d^2, as in k_compress_large
        if (idw) {
            for (uint32_t d = tid + 1; d < n; d += T) {
                const double dd = (double)d;
                wtab[d] = 1.0 / (dd * dd);
            }
            __syncthreads();
        }
        if (!idw && step > 1) {
            // Catmull-Rom with the tables of k_compress_large (bit-identical there to the oracle's
            // polynomial_to_data): tangents once per segment, Hermite basis once per in-segment offset,
            // exact r / step by the reciprocal with one FMA correction; linear first and last segment.
            double2 *mm = (double2 *)(ws + lay.o_mm);
            double4 *hbt = (double4 *)(smem + 256);  // step <= 255 entries of 32 B: the payload window is done with
            const uint32_t gapL = (n - 1) - (K - 2) * step;
            const double stepd = (double)step, gapLd = (double)gapL;
            const double ry = 1.0 / stepd, ryL = 1.0 / gapLd;
            for (uint32_t sg = tid + 1; sg + 2 < K; sg += T) {
                const uint32_t t0i = sg * step;
                const uint32_t t1i = (sg + 1 == K - 1) ? (n - 1) : (sg + 1) * step;
                const uint32_t tmi = (sg - 1) * step;
                const uint32_t tpi = (sg + 2 == K - 1) ? (n - 1) : (sg + 2) * step;
                const double t0 = (double)t0i, t1 = (double)t1i;
                const double v0 = vals[sg], v1 = vals[sg + 1], vm = vals[sg - 1], vp = vals[sg + 2];
                double2 t;
                t.x = (v1 - vm) / (t1 - (double)tmi) * (t1 - t0);
                t.y = (vp - v0) / ((double)tpi - t0) * (t1 - t0);
                mm[sg] = t;
            }
            for (uint32_t r = tid; r < step; r += T) {
                const double nt = div_small((double)r, stepd, ry);
                const double t2 = nt * nt;
                const double t3 = t2 * nt;
                const double two_t3 = t3 * 2.0;
                const double two_t2 = t2 * 2.0;
                const double three_t2 = t2 * 3.0;
                double4 hh;
                hh.x = two_t3 - three_t2 + 1.0;
                hh.y = t3 - two_t2 + nt;
                hh.z = three_t2 - two_t3;
                hh.w = t3 - t2;
                hbt[r] = hh;
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += T) {
                double sv;
                if (i == n - 1) {
                    sv = vals[K - 1];
                } else {
                    uint32_t sg = __umulhi(i, magic);
                    if (sg > K - 2) sg = K - 2;
                    const uint32_t t0i = sg * step;
                    const bool last = (sg == K - 2);
                    const double v0 = vals[sg], v1 = vals[sg + 1];
                    if (sg > 0 && !last) {
                        const double2 t = mm[sg];
                        const double4 hh = hbt[i - t0i];
                        sv = v0 * hh.x + t.x * hh.y + v1 * hh.z + t.y * hh.w;
                    } else {
                        const double nt = div_small((double)(i - t0i), last ? gapLd : stepd, last ? ryL : ry);
                        sv = v0 * (1.0 - nt) + v1 * nt;
                    }
                }
                double o = div1e5(round(sv * 100000.0));
                if (o < mn) o = mn;
                else if (o > mx) o = mx;
                out[i] = o;
            }
            return;
        }
        for (uint32_t j = tid; j < n; j += T) {
            double sv;
            if (idw) {
                const uint32_t q = j / step;
                if (j == n - 1 || (q * step == j && q < K - 1)) {
                    sv = vals[j == n - 1 ? K - 1 : q];
                } else {
                    double num = 0.0, den = 0.0;
                    for (uint32_t k = 0; k < K; ++k) {
                        const uint32_t pk = (k == K - 1) ? (n - 1) : k * step;
                        const double w = wtab[pk > j ? pk - j : j - pk];
                        num += w * vals[k];
                        den += w;
                    }
                    sv = num / den;
                }
            } else {
                sv = spline_eval([&](uint32_t k) { return vals[k]; }, j, n, step, K, magic);
            }
            double o = round(sv * 100000.0) / 100000.0;
            if (o < mn) o = mn;
            else if (o > mx) o = mx;
            out[j] = o;
        }
        return;
    }
    if (fr.tag == ATSC_FFT) {
        const float mxf = h.f0, mnf = h.f1;
        if (mxf == mnf) {
            for (uint32_t j = tid; j < n; j += T) out[j] = (double)mxf;
            return;
        }
        if (h.u1) {
            // entries are applied in stream order, later ones overwrite (fft.rs:401-422): the last entry
            // naming a position owns it
            const uint32_t cnt = h.u0, nbin = L / 2 + 1;
            for (uint32_t k = tid; k < nbin; k += T) own[k] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += T) atomicMax(&own[ent[i].pos], i + 1);
            __syncthreads();
            if (PH == 0 && sparse && P.sp_mf && sp_split) {
                // The list is bucketed here; the tiles -- independent of each other from here on -- go to
                // k_decompress_large_tiles, a (tile, frame) grid over the whole GPU: a batch of few frames
                // is bound by one workgroup's 9+ tiles in a row otherwise.
                sparse_bucket(
                    P, cnt,
                    [&](uint32_t i, uint32_t &p, float2 &x) -> bool {
                        const Sel e = ent[i];
                        p = e.pos;
                        x = make_float2(e.re, e.im);
                        return own[p] == i + 1;
                    },
                    (SpEnt *)Cb, tw, smem + 256, wsum);
                const SpLds sl = sp_lds(P, smem + 256);
                uint32_t *gb = (uint32_t *)(ws + lay.o_nb);  // 2 Mf words of the (idle) norm-bit region
                for (uint32_t e = tid; e < P.sp_mf; e += T) {
                    gb[e] = sl.beg[e];
                    gb[P.sp_mf + e] = sl.end[e];
                }
                if (tid == 0) { pend->mxf = mxf; pend->mnf = mnf; pend->pending = 2; }
                return;
            }
            if (PH == 0 && sparse && P.sp_mf) {
                const double mxd = (double)mxf, mnd = (double)mnf;
                const float Lf = (float)L;
                sparse_inverse(
                    P, cnt,
                    [&](uint32_t i, uint32_t &p, float2 &x) -> bool {
                        const Sel e = ent[i];
                        p = e.pos;
                        x = make_float2(e.re, e.im);
                        return own[p] == i + 1;
                    },
                    (SpEnt *)Cb, tw, smem + 256, wsum,
                    [](uint32_t) -> int { return 0; },
                    [&](uint32_t j, float re, int) {
                        const uint32_t i = j - pre;
                        if (i < n) {
                            const float v = re / Lf;
                            double o = round((double)v * 100000.0) / 100000.0;
                            if (o > mxd) o = mxd;
                            if (o < mnd) o = mnd;
                            out[i] = o;
                        }
                    });
                return;
            }
            for (uint32_t i = tid; i < cnt; i += T) {
                const Sel e = ent[i];
                if (own[e.pos] == i + 1) Xs[e.pos] = make_float2(e.re, e.im);
            }
            __syncthreads();
        }
        float2 *F;
        if (P.half) {
            for (uint32_t k = tid; k < M; k += T) {
                const float2 xk = Xs[k], xm = Xs[M - k];
                const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
                const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
                const float2 o = cmulp(d, tw[k]);
                A[k] = make_float2(e.x - o.y, -(e.y + o.x));
            }
        } else {
            for (uint32_t k = tid; k < L; k += T) {
                float2 v;
                if (k <= L / 2) v = Xs[k];
                else { const float2 c = Xs[L - k]; v = make_float2(c.x, -c.y); }
                A[k] = make_float2(v.x, -v.y);
            }
        }
        __syncthreads();
        if (PH == 1) {  // the batched kernels take it from here
            if (tid == 0) { pend->mxf = mxf; pend->mnf = mnf; pend->pending = 1; }
            return;
        }
        F = fft_large(P, A, Cb, tw, (float2 *)(smem + 256), tiled != 0);
        const double mxd = (double)mxf, mnd = (double)mnf;
        const float Lf = (float)L;
        for (uint32_t i = tid; i < n; i += T) {
            const uint32_t j = i + pre;
            float re;
            if (P.half) {
                const float2 f = F[j >> 1];
                re = 2.0f * ((j & 1) ? -f.y : f.x);
            } else {
                re = F[j].x;
            }
            const float v = re / Lf;
            double o = round((double)v * 100000.0) / 100000.0;
            if (o > mxd) o = mxd;
            if (o < mnd) o = mnd;
            out[i] = o;
        }
        return;
    }
    {  // RLE (rle.rs:204-236)
        const uint32_t E = h.u0;
        uint32_t p2 = 1;
        while (p2 < E) p2 <<= 1;
        if (PH == 0 && E >= 1 && E <= RLE_LDS_RUNS && h.u1 <= RLE_LDS_RUNS) {
            // Up to RLE_LDS_RUNS runs (a 131072-sample frame of a slowly changing gauge has about a thousand): the run
            // starts are sorted in LDS and every run is written by one wavefront, 64 samples a store, its value read
            // once -- instead of a sort in the workspace and a binary search of it per sample (ten dependent L2 round
            // trips for each of the frame's samples: 530 us for six such frames).  Runs that share a start keep the
            // reference's outcome: sorted by (start, group), all but the last of them are empty.
            uint64_t *lk = (uint64_t *)(smem + 256 + STG_BYTES);
            double *lv = (double *)(lk + RLE_LDS_RUNS);
            for (uint32_t i = tid; i < E; i += T) lk[i] = keys[i];
            for (uint32_t i = tid; i < h.u1; i += T) lv[i] = vals[i];
            __syncthreads();
            block_sort<LW, true>(lk, nullptr, E, p2);
            const uint32_t first = (uint32_t)(lk[0] >> 32);
            for (uint32_t j = tid; j < first; j += T) out[j] = 0.0;
            const uint32_t lane = tid & 63u;
            for (uint32_t i = tid >> 6; i < E; i += LW) {
                const uint64_t k = lk[i];
                const uint32_t start = (uint32_t)(k >> 32);
                const uint32_t end = (i + 1 < E) ? (uint32_t)(lk[i + 1] >> 32) : n;
                const double v = lv[(uint32_t)(k & 0xffffffffu)];
                for (uint32_t j = start + lane; j < end; j += 64) out[j] = v;
            }
            return;
        }
        block_sort<LW, true>(keys, nullptr, E, p2);
        for (uint32_t j = tid; j < n; j += T) {
            uint32_t lo = 0, hi = E;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)(keys[mid] >> 32) <= j) lo = mid + 1;
                else hi = mid;
            }
            out[j] = lo ? vals[(uint32_t)(keys[lo - 1] & 0xffffffffu)] : 0.0;
        }
    }
}

// A workgroup per frame of the launch; behind the decoder's grid path with no tile grid of its own to feed (fast_skip == 2)
// the launch is as wide as the GPU part at most and shares out the frames k_large_dparse left alone (listed by the first
// k_large_trip243<true> launch), as k_compress_large does.
template <int PH>
__global__ __launch_bounds__(LT) void k_decompress_large(
    const DevDFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool,
    const uint8_t *__restrict__ body, double *__restrict__ outp, int *__restrict__ status,
    unsigned char *__restrict__ ws_base, uint64_t ws_stride, int tiled, int sparse, int sp_split, int fast_skip)
{
    const bool listed = PH == 0 && fast_skip == 2;
    const uint32_t count = listed ? *fb_count(ws_base, ws_stride) : gridDim.x;
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {  // (not listed: one turn, it = blockIdx.x)
        const uint32_t bx = listed ? *fb_entry(ws_base, ws_stride, it) : it;
        decompress_large_frame<PH>(bx, frames, ids, plans, twpool, body, outp, status, ws_base, ws_stride, tiled, sparse, sp_split,
                                   fast_skip);
        __syncthreads();
    }
}

// One tile (SPB output columns) of one FFT frame whose list k_decompress_large<0> bucketed (DecPending::pending == 2)
__global__ __launch_bounds__(LT) void k_decompress_large_tiles(
    const DevDFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool, double *__restrict__ outp,
    unsigned char *__restrict__ ws_base, uint64_t ws_stride, uint32_t nframes)
{
    // grid (frames padded to a multiple of 8, tiles): a frame's tiles share one XCD (see k_large_trip_tiles)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t fidx = blockIdx.x;
    if (fidx >= nframes) return;
    const DevDFrame fr = frames[ids[fidx]];
    if (fr.tag != ATSC_FFT) return;
    const DevPlan &P = plans[fr.plan];
    const uint32_t jb0 = blockIdx.y * SPB;
    if (!P.sp_mf || jb0 >= P.sp_md) return;
    unsigned char *ws = ws_base + (uint64_t)fidx * ws_stride;
    const LargeWs lay = large_ws_layout(fr.n, P.L, P.kcap);
    const DecPending *pend = (const DecPending *)(ws + lay.o_cnt);
    if (pend->pending != 2) return;
    const uint32_t n = fr.n, L = P.L, pre = P.pre, Mf = P.sp_mf, Md = P.sp_md;
    const float2 *tw = twpool + P.tw_off;
    const SpLds sl = sp_lds(P, smem);
    const uint32_t *gb = (const uint32_t *)(ws + lay.o_nb);
    for (uint32_t e = tid; e < Mf; e += LT) {
        sl.wf[e] = tw[e * (L / Mf)];
        sl.beg[e] = gb[e];
        sl.end[e] = gb[Mf + e];
    }
    for (uint32_t e = tid; e < Md; e += LT) sl.wd[e] = tw[e * (L / Md)];
    __syncthreads();
    double *out = outp + fr.out_off;
    const double mxd = (double)pend->mxf, mnd = (double)pend->mnf;
    const float Lf = (float)L;
    sparse_tile(
        P, jb0, (const SpEnt *)(ws + lay.o_c), tw, smem, [](uint32_t) -> int { return 0; },
        [&](uint32_t j, float re, int) {
            const uint32_t i = j - pre;
            if (i < n) {
                const float v = re / Lf;
                double o = round((double)v * 100000.0) / 100000.0;
                if (o > mxd) o = mxd;
                if (o < mnd) o = mnd;
                out[i] = o;
            }
        });
}

hipError_t launch_decompress_large(uint32_t count, const DevDFrame *frames, const uint32_t *ids,
                                   const DevPlan *plans, const float2 *twpool, const uint8_t *body,
                                   double *out, int *status, unsigned char *ws, uint64_t ws_stride,
                                   uint32_t ws_slots, int tiled, int sparse, hipStream_t s, const LargePre *pre,
                                   uint32_t sp_tiles)
{
    // 256 B of header scratch + the tile buffers of the LDS-tiled inverse transform
    const uint32_t lds = 256 + max((2 * F4_TILE + 2 * F4_MAX) * (uint32_t)sizeof(float2), SP_LDS_BYTES);
    // the batched inverse transform serves the dense form only: from the sparse bin list a frame's
    // decoder needs no pass over the workspace at all
    const bool split = pre && pre->tiles1 && !sparse;
    hipError_t e;
    if (sparse && sp_tiles) {
        e = ensure_dyn_lds((const void *)k_decompress_large_tiles, SP_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    uint32_t lds1 = 0, lds2 = 0;
    if (split) {
        lds1 = (2 * FB * pre->m1_max + pre->m1_max) * (uint32_t)sizeof(float2);
        lds2 = (2 * FB * (pre->m2_max + 1) + pre->m2_max) * (uint32_t)sizeof(float2);
        e = ensure_dyn_lds((const void *)k_large_pre1<DevDFrame, true>, lds1);
        if (e != hipSuccess) return e;
        e = ensure_dyn_lds((const void *)k_large_pre2<DevDFrame, true>, lds2);
        if (e != hipSuccess) return e;
    } else {
        e = ensure_dyn_lds((const void *)k_decompress_large<0>, lds);
        if (e != hipSuccess) return e;
    }
    for (uint32_t b0 = 0; b0 < count; b0 += ws_slots) {
        const uint32_t nb = min(ws_slots, count - b0);
        if (split) {
            // parse + every codec but the FFT transform; the transform of all pending frames over the whole
            // GPU; scale / round / clamp
            hipLaunchKernelGGL(k_decompress_large<1>, dim3(nb), dim3(LT), 256 + STG_BYTES, s, frames, ids + b0, plans, twpool,
                               body, out, status, ws, ws_stride, tiled, sparse, 0, 0);
            hipLaunchKernelGGL((k_large_pre1<DevDFrame, true>), dim3(pre->tiles1, nb), dim3(PT), lds1, s,
                               (const double *)nullptr, frames, ids + b0, plans, twpool, ws, ws_stride);
            hipLaunchKernelGGL((k_large_pre2<DevDFrame, true>), dim3(pre->tiles2, nb), dim3(PT), lds2, s,
                               (const double *)nullptr, frames, ids + b0, plans, twpool, ws, ws_stride);
            hipLaunchKernelGGL(k_decompress_large<2>, dim3(nb), dim3(LT), 256, s, frames, ids + b0, plans, twpool,
                               body, out, status, ws, ws_stride, tiled, sparse, 0, 0);
        } else {
            // (with every CU busy on its own frame the grid only repeats the table loads per tile: e = 1 %,
            // 512 frames: 1.53 vs 1.12 ms)
            const int sp_split = (sparse && sp_tiles && count <= LARGE_SPLIT_MAX) ? 1 : 0;
            // FFT frames of 131072 samples: k_large_dparse + the tile grid (atsc_large_fast.h); the rest, and whatever
            // the parser leaves alone, behind them
            const bool no_fast = getenv("ATSC_LARGE_NO_FAST") != nullptr;  // (read per launch: the tests switch it)
            const int fast = (!no_fast && sparse && pre && pre->cols243 && pre->rows9p != 0) ? 1 : 0;
            if (fast) {
                e = ensure_dyn_lds((const void *)k_large_dparse, FAST_DP_LDS);
                if (e != hipSuccess) return e;
                static const int dbg = getenv("ATSC_DEBUG_DPARSE") ? 1 : 0;
                hipLaunchKernelGGL(k_large_dparse, dim3(nb), dim3(LT), FAST_DP_LDS, s, frames, ids + b0, plans, twpool, body,
                                   ws, ws_stride, dbg);
                e = launch_trip243<true, DevDFrame>(pre->rows9p, (pre->m2_max + 15) / 16, nb, s, (const double *)nullptr, frames, ids + b0,
                                                    plans, twpool, ws, ws_stride, 4, out);  // (4: the first launch lists the frames left alone)
                if (e != hipSuccess) return e;
            }
            // (listed: see k_decompress_large; with a tile grid behind it every frame's workgroup has to clear its pending mark)
            const bool listed = fast && !sp_split;
            hipLaunchKernelGGL(k_decompress_large<0>, dim3(listed ? min(nb, FB_GRID_MAX) : nb), dim3(LT), lds, s, frames, ids + b0,
                               plans, twpool, body, out, status, ws, ws_stride, tiled, sparse, sp_split, listed ? 2 : fast);
            if (sp_split)
                hipLaunchKernelGGL(k_decompress_large_tiles, dim3((nb + 7u) & ~7u, sp_tiles), dim3(LT), SP_LDS_BYTES, s,
                                   frames, ids + b0, plans, twpool, out, ws, ws_stride, nb);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace atsc
