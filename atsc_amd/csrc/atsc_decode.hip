// atsc_decode.hip -- gfx950 kernels for CompressorFrame::decompress (atsc/src/frame/mod.rs:152-158
// -> Compressor::decompress, atsc/src/compressor/mod.rs:109-119).
//
// One workgroup of W wavefronts per frame.  The payload is walked by the first wavefront through an
// LDS window (varint fields are sequential by construction); the per-sample reconstruction is parallel:
//   FFT         fft.rs:426-462      K-sparse Hermitian spectrum -> direct sum over the stored bins
//                                   (f64 accumulation, f32 table twiddles), f32 result / L, round 5, clamp
//   Polynomial  polynomial.rs:395-404,342-373  Catmull-Rom/linear pieces, exact f64 op order
//   Constant    constant.rs:141-144 ; RLE rle.rs:204-236 ; Noop noop.rs:79-83
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "atsc_device.h"

namespace atsc {

#ifndef ATSC_DEC_FFT_MIN_K
#define ATSC_DEC_FFT_MIN_K 16
#endif
constexpr uint32_t DEC_FFT_MIN_K = ATSC_DEC_FFT_MIN_K;  // stored bins from which a multi-wavefront frame decodes by inverse FFT

template <int W, int SPL>
__global__ __launch_bounds__(64 * W) void k_decompress(
    const DevDFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool,
    const uint8_t *__restrict__ body, double *__restrict__ outp, int *__restrict__ status)
{
    constexpr int T = 64 * W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const DevDFrame fr = frames[ids ? ids[blockIdx.x] : blockIdx.x];  // ids == nullptr: the class is every frame, in order
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = fr.n;
    double *out = outp + fr.out_off;
    const uint8_t *pay = body + fr.payload_off;

    double *xs = (double *)(smem + P.o_xs);      // 8n : knot values / rle group values
    float2 *tw = (float2 *)(smem + P.o_tw);      // 8L
    unsigned char *AB = smem + P.o_ab;           // >= 16L + 8 bytes, contiguous A|B
    uint32_t *aux = (uint32_t *)(smem + P.o_aux);  // 4(n+2)
    double *red = (double *)(smem + P.o_red);
    // header scalars broadcast through LDS (written by lane 0)
    struct Hdr {
        double d0, d1;
        uint32_t u0, u1, u2, bad;
        float f0, f1;
    };
    Hdr *hdr = (Hdr *)(red + 40);

    // The first wavefront walks the payload in lock step (every lane computes and writes the same
    // values) through an LDS window refilled by its 64 lanes -- the twiddle region, idle until the header is
    // known -- instead of one lane pulling bytes out of global memory one dependent read at a time.  Runs of
    // variable-width fields are parsed 64 at a time when the window holds a whole group (atsc_device.h).
    if (tid < 64) {
        const uint32_t cap = (P.o_ab - P.o_tw) & ~15u;
        RdS r{pay, fr.payload_len, 0, false, smem + P.o_tw, 0, 0, cap};
        const bool wide = cap >= 11 * 64 + 16;
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
            if (r.bad) break;
            if (wide) rds_varints(r, n, [&](uint32_t i, uint64_t v) { xs[i] = (double)unzig(v); });
            else for (uint32_t i = 0; i < n && !r.bad; ++i) xs[i] = (double)unzig(rds_varint(r));
            break;
        }
        case ATSC_IDW:
        case ATSC_POLYNOMIAL: {
            const uint32_t id = (uint32_t)rds_varint(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t cnt = rds_varint(r);
            if (id > 1 || bd > 3 || cnt > n) r.bad = true;
            h.u2 = id;  // PolynomialType decides the interpolation (polynomial.rs:400-403), not the frame tag
            if (!r.bad) {
                if (bd == 0 || bd == 3) {
                    // fixed-width points (F64 / U8): skipped here, read by the whole workgroup below
                    h.f0 = __uint_as_float(r.pos);
                    r.pos += (uint32_t)cnt * (bd == 0 ? 8u : 1u);
                    if (r.pos > r.len) r.bad = true;
                } else if (wide) {
                    rds_varints(r, (uint32_t)cnt, [&](uint32_t i, uint64_t v) {
                        xs[i] = (bd == 2) ? (double)(int16_t)unzig(v) : (double)(int32_t)unzig(v);
                    });
                } else {
                    for (uint32_t i = 0; i < cnt && !r.bad; ++i) xs[i] = rds_value(r, bd);
                }
            }
            h.f1 = __uint_as_float(bd);
            h.d0 = __longlong_as_double((long long)rds_le(r, 8));  // min
            h.d1 = __longlong_as_double((long long)rds_le(r, 8));  // max
            h.u0 = (uint32_t)cnt;
            h.u1 = rds_u8(r);  // point_step
            break;
        }
        case ATSC_FFT: {
            (void)rds_u8(r);
            const uint64_t cnt = rds_varint(r);
            if (cnt > P.bins) r.bad = true;
            Sel *sel = (Sel *)AB;
            if (!r.bad) {
                if (wide) {
                    rds_fft_entries(r, (uint32_t)cnt, P.L, sel);
                } else {
                    for (uint32_t i = 0; i < cnt && !r.bad; ++i) {
                        uint32_t pos = (uint32_t)rds_varint(r) & 0xffffu;
                        const float re = rds_f32(r);
                        float im = rds_f32(r);
                        if (pos >= P.L) { r.bad = true; break; }
                        if (pos > P.L / 2) { pos = P.L - pos; im = -im; }  // fft.rs:401-422
                        if (pos == 0 || 2 * pos == P.L) im = 0.0f;
                        sel[i].pos = pos; sel[i].re = re; sel[i].im = im;
                    }
                }
            }
            h.u0 = (uint32_t)cnt;
            h.f0 = rds_f32(r);  // max_value
            h.f1 = rds_f32(r);  // min_value
            break;
        }
        case ATSC_RLE: {
            (void)rds_u8(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t groups = rds_varint(r);
            if (bd > 3 || groups > n) r.bad = true;
            uint64_t *keys = (uint64_t *)AB;  // (run start << 32) | group
            uint32_t e = 0;
            for (uint32_t gi = 0; gi < groups && !r.bad; ++gi) {
                xs[gi] = rds_value(r, bd);
                const uint64_t cnt = rds_varint(r);
                if (cnt > n - e) { r.bad = true; break; }
                for (uint32_t k = 0; k < cnt && !r.bad; ++k) {
                    const uint64_t idx = rds_varint(r);
                    if (idx >= n) { r.bad = true; break; }
                    keys[e++] = (idx << 32) | gi;
                }
            }
            h.u0 = e;
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
    if (fr.tag == ATSC_NOOP) {
        for (uint32_t j = tid; j < n; j += T) out[j] = xs[j];
        return;
    }
    if (fr.tag == ATSC_POLYNOMIAL || fr.tag == ATSC_IDW) {
        const double mn = h.d0, mx = h.d1;
        if (mx == mn) {  // polynomial.rs:396-399
            for (uint32_t j = tid; j < n; j += T) out[j] = mx;
            return;
        }
        const uint32_t step = h.u1, K = h.u0, bd = __float_as_uint(h.f1);
        if (bd == 0 || bd == 3) {
            const uint8_t *pp = pay + __float_as_uint(h.f0);
            for (uint32_t k = tid; k < K; k += T) {
                if (bd == 3) {
                    xs[k] = (double)pp[k];
                } else {
                    uint64_t v = 0;
                    for (int b = 0; b < 8; ++b) v |= (uint64_t)pp[8 * k + b] << (8 * b);
                    xs[k] = __longlong_as_double((long long)v);
                }
            }
            __syncthreads();
        }
        bool ok = step >= 1;
        if (ok) {
            const uint32_t cnt = (n + step - 1) / step;
            const uint32_t Kp = cnt + (((cnt - 1) * step != n - 1) ? 1u : 0u);
            ok = (Kp == K) && K >= 2;
        }
        if (!ok) {  // a stream the reference's encoder cannot produce
            if (tid == 0) atomicExch(status, 1);
            return;
        }
        if (h.u2 == 1) {  // idw_to_data: polynomial.rs:375-393
            for (uint32_t j = tid; j < n; j += T) {
                const double x = (double)j;
                double num = 0.0, den = 0.0, sv = 0.0;
                bool hit = false;
                for (uint32_t k = 0; k < K && !hit; ++k) {
                    const uint32_t pk = (k == K - 1) ? (n - 1) : k * step;
                    const double d = fabs((double)pk - x);
                    if (d == 0.0) { hit = true; sv = xs[k]; }
                    else { const double w = 1.0 / (d * d); num += w * xs[k]; den += w; }
                }
                if (!hit) sv = num / den;
                double o = round(sv * 100000.0) / 100000.0;
                if (o < mn) o = mn;
                else if (o > mx) o = mx;
                out[j] = o;
            }
            return;
        }
        const uint32_t magic = (uint32_t)(0x100000000ull / step) + 1u;
        // the twiddle region and the work buffers behind it are contiguous and idle here: basis table, then tangents
        const uint32_t cap_tab = (P.o_ab - P.o_tw) + P.ab_bytes;
        if (step > 1 && 32u * step + 16u * K <= cap_tab) {
            // Catmull-Rom with the tables of the compressor's ladder (bit-identical there to the oracle's
            // polynomial_to_data): tangents once per segment, Hermite basis once per in-segment offset,
            // exact r / step by the reciprocal with one FMA correction; linear first and last segment.
            double4 *hbt = (double4 *)(smem + P.o_tw);  // the payload window is done with
            double2 *mm = (double2 *)(smem + P.o_tw + 32u * step);
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
                const double v0 = xs[sg], v1 = xs[sg + 1], vm = xs[sg - 1], vp = xs[sg + 2];
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
                    sv = xs[K - 1];
                } else {
                    uint32_t sg = __umulhi(i, magic);
                    if (sg > K - 2) sg = K - 2;
                    const uint32_t t0i = sg * step;
                    const bool last = (sg == K - 2);
                    const double v0 = xs[sg], v1 = xs[sg + 1];
                    if (sg > 0 && !last) {
                        const double2 t = mm[sg];
                        const double4 hh = hbt[i - t0i];
                        sv = v0 * hh.x + t.x * hh.y + v1 * hh.z + t.y * hh.w;
                    } else {
                        const double nt = div_small((double)(i - t0i), last ? gapLd : stepd, last ? ryL : ry);
                        sv = v0 * (1.0 - nt) + v1 * nt;
                    }
                }
                double o = div1e5(round(sv * 100000.0));  // utils/mod.rs:66-74 (min first)
                if (o < mn) o = mn;
                else if (o > mx) o = mx;
                out[i] = o;
            }
            return;
        }
        for (uint32_t j = tid; j < n; j += T) {
            const double sv = spline_eval([&](uint32_t k) { return xs[k]; }, j, n, step, K, magic);
            double o = round(sv * 100000.0) / 100000.0;  // utils/mod.rs:66-74 (min first)
            if (o < mn) o = mn;
            else if (o > mx) o = mx;
            out[j] = o;
        }
        return;
    }
    if (fr.tag == ATSC_FFT) {
        const float mxf = h.f0, mnf = h.f1;
        if (mxf == mnf) {  // fft.rs:427-430
            for (uint32_t j = tid; j < n; j += T) out[j] = (double)mxf;
            return;
        }
        // fft.rs:432-444: the decoder recomputes the Gibbs padding from the frame size
        const uint32_t L = P.L, pre = P.pre;
        const float2 *twp = twpool + P.tw_off;
        for (uint32_t j = tid; j < L; j += T) tw[j] = twp[j];
        __syncthreads();
        const Sel *sel = (const Sel *)AB;
        const uint32_t K = h.u0;
        // entries are applied in stream order, later ones overwrite (fft.rs:401-422): the last entry naming
        // a (mirrored) position owns it
        uint32_t *own = aux;  // bins = L/2 + 1 <= n + 2 words
        for (uint32_t k = tid; k <= L / 2; k += T) own[k] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < K; i += T) atomicMax(&own[sel[i].pos], i + 1);
        __syncthreads();
        const double mxd = (double)mxf, mnd = (double)mnf;
        const float Lf = (float)L;
        const uint32_t magicL = P.magicL;
        // Multi-wavefront frames with more than a handful of stored bins: one inverse transform in LDS -- what the
        // reference runs (fft.rs:446-456: an f32 inverse FFT of the mirrored spectrum) -- instead of the direct sum,
        // which is K x L multiply-adds in f64 (a 4096-sample frame with 80 bins: 16 us on its CU, which holds one
        // such workgroup; `tools/decode_length_probe.py`: 59 Gsamples/s where 2048-sample frames decode at 98).
        // Even L: the Hermitian spectrum X folds into Zf = E + i O of length M = L / 2 (E, O: the transforms of the
        // even and the odd samples, E[k] = (X[k] + conj X[M-k]) / 2, O[k] = w^-k (X[k] - conj X[M-k]) / 2) and
        // idft_L = 2 idft_M: even sample -> re, odd -> im (the inverse of the encoder's untangle step, as in
        // k_large_trip243<true>).  Odd L: the full mirrored spectrum.  The inverse runs as conj(FFT(conj(.))) through
        // the encoder's forward stages.  sel[] (in A) is consumed before A is reused.
        if constexpr (W > 1) {
            const uint32_t M = P.M;
            if (!P.direct && K >= DEC_FFT_MIN_K && 12u * K <= P.ab_half) {
                float2 *A = (float2 *)AB, *B = (float2 *)(AB + P.ab_half);
                const float2 *Y;
                if (P.half) {
                    for (uint32_t k = tid; k <= M; k += T) B[k] = make_float2(0.0f, 0.0f);
                    __syncthreads();
                    for (uint32_t i = tid; i < K; i += T) {
                        const Sel e = sel[i];
                        if (own[e.pos] == i + 1) B[e.pos] = make_float2(e.re, e.im);  // (mirrored positions: <= L / 2 = M)
                    }
                    __syncthreads();
                    for (uint32_t k = tid; k < M; k += T) {
                        const float2 xk = B[k], xm = B[M - k];
                        const float2 w = tw[k];
                        const float2 E = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
                        const float2 D = make_float2(xk.x - xm.x, xk.y + xm.y);
                        const float2 O = make_float2(0.5f * (D.x * w.x - D.y * w.y), 0.5f * (D.x * w.y + D.y * w.x));
                        A[k] = make_float2(E.x - O.y, -(E.y + O.x));  // conj(E + i O)
                    }
                    __syncthreads();
                    Y = fft_forward<W>(P, A, B, tw);
                } else {
                    for (uint32_t k = tid; k < L; k += T) B[k] = make_float2(0.0f, 0.0f);
                    __syncthreads();
                    for (uint32_t i = tid; i < K; i += T) {
                        const Sel e = sel[i];
                        if (own[e.pos] == i + 1) {
                            B[e.pos] = make_float2(e.re, -e.im);                    // conj X[pos]
                            if (e.pos != 0) B[L - e.pos] = make_float2(e.re, e.im);  // conj X[L - pos] = X[pos]
                        }
                    }
                    __syncthreads();
                    Y = fft_forward<W>(P, B, A, tw);
                }
#pragma unroll
                for (int m = 0; m < SPL; ++m) {
                    const uint32_t j = tid + m * T;
                    if (j < n) {
                        const uint32_t jj = j + pre;
                        float re;
                        if (P.half) {
                            const float2 y = Y[jj >> 1];
                            re = (jj & 1u) ? -2.0f * y.y : 2.0f * y.x;
                        } else {
                            re = Y[jj].x;
                        }
                        const float v = re / Lf;  // fft.rs:460  f.re / len (f32)
                        double o = round((double)v * 100000.0) / 100000.0;
                        if (o > mxd) o = mxd;
                        if (o < mnd) o = mnd;
                        out[j] = o;
                    }
                }
                return;
            }
        }
        // A thread owns samples tid, tid + T, ... (at most SPL of them: n <= L <= 64 W SPL); the entries are the
        // outer loop so that an entry's constants and its twiddle index walk (pos * (j + pre) mod L,
        // advanced by pos * T mod L) are set up once.  Per sample the terms still add up in stream order,
        // and 2 (re wx - im wy) = (2 re) wx - (2 im) wy exactly, so the sums are those of the sample-outer form.
        constexpr int MAXS = SPL;
        double acc[MAXS];
#pragma unroll
        for (int m = 0; m < MAXS; ++m) acc[m] = 0.0;
        auto add_entry = [&](const uint32_t pos, const float re, const float im) {
            if (pos == 0) {
#pragma unroll
                for (int m = 0; m < MAXS; ++m) acc[m] += (double)re;
                return;
            }
            const double cf = (2 * pos == L) ? 1.0 : 2.0;  // fft.rs:401-422
            const double a = cf * (double)re, bq = cf * (double)im;
            uint32_t idx = mod_magic(pos * (tid + pre), L, magicL);  // pos * jj < 2^32
            const uint32_t stp = mod_magic(pos * (uint32_t)T, L, magicL);
#pragma unroll
            for (int m = 0; m < MAXS; ++m) {
                const float2 w = tw[idx];
                acc[m] += a * (double)w.x - bq * (double)w.y;
                idx += stp;
                idx = min(idx, idx - L);  // idx < 2L: the unsigned wrap picks the reduced value
            }
        };
        if (W == 1 && K <= 64) {
            // one wavefront, at most one entry per lane: lane i looks entry i and its owner up once, and the walk
            // takes position and value from lane i (v_readlane) -- two dependent LDS round trips per entry less
            uint32_t e_pos = 0;
            float e_re = 0.0f, e_im = 0.0f;
            bool valid = false;
            if (tid < K) {
                const Sel e = sel[tid];
                e_pos = e.pos; e_re = e.re; e_im = e.im;
                valid = own[e.pos] == tid + 1;
            }
            uint64_t todo = __ballot(valid);
            while (todo) {  // ascending entry index: the terms add up in stream order
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                add_entry((uint32_t)__builtin_amdgcn_readlane((int)e_pos, i),
                          __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(e_re), i)),
                          __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(e_im), i)));
            }
        } else {
            for (uint32_t i = 0; i < K; ++i) {
                const Sel e = sel[i];
                if (own[e.pos] != i + 1) continue;
                add_entry(e.pos, e.re, e.im);
            }
        }
#pragma unroll
        for (int m = 0; m < MAXS; ++m) {
            const uint32_t j = tid + m * T;
            if (j < n) {
                const float v = (float)acc[m] / Lf;  // fft.rs:460  f.re / len (f32)
                double o = round((double)v * 100000.0) / 100000.0;  // fft.rs:208-218 (max first)
                if (o > mxd) o = mxd;
                if (o < mnd) o = mnd;
                out[j] = o;
            }
        }
        return;
    }
    // RLE: rle.rs:204-236 -- sort (start, value) by start, then each sample takes the last run
    // starting at or before it; samples before the first run stay 0.0
    {
        uint64_t *keys = (uint64_t *)AB;
        const uint32_t E = h.u0;
        uint32_t p2 = 1;
        while (p2 < E) p2 <<= 1;
        block_sort<W, true>(keys, nullptr, E, p2);
        for (uint32_t j = tid; j < n; j += T) {
            uint32_t lo = 0, hi = E;  // first entry with start > j
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)(keys[mid] >> 32) <= j) lo = mid + 1;
                else hi = mid;
            }
            out[j] = lo ? xs[(uint32_t)(keys[lo - 1] & 0xffffffffu)] : 0.0;
        }
        (void)aux;
    }
}

template <int W, int SPL>
static hipError_t launch_d(uint32_t count, uint32_t lds, const DevDFrame *frames,
                           const uint32_t *ids, const DevPlan *plans, const float2 *twpool,
                           const uint8_t *body, double *out, int *status, hipStream_t s)
{
    auto kern = k_decompress<W, SPL>;
    if (lds > 48 * 1024) {
        hipError_t e = ensure_dyn_lds((const void *)kern, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(count), dim3(64 * W), lds, s, frames, ids, plans, twpool, body, out,
                       status);
    return hipGetLastError();
}

// A small table from page-locked host memory into device memory by a kernel on the caller's stream instead of a copy
// engine's queue: a synchronous hipMemcpy issued while a large device-to-host transfer is in flight waits behind it
// (atsc_host.cpp: decompress_frames_halves).  src: a hipHostMalloc'd buffer (mapped into the device's address space).
__global__ void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
hipError_t launch_copy_words(void *dst, const void *src_host_mapped, uint64_t n_words, hipStream_t s)
{
    if (n_words == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(512, (n_words + 255) / 256);
    hipLaunchKernelGGL(k_copy_words, dim3(blocks), dim3(256), 0, s, (uint32_t *)dst, (const uint32_t *)src_host_mapped,
                       n_words);
    return hipGetLastError();
}

// frame classes as in the compressor (class_of, atsc_host.cpp): L <= 64 W SPL samples per workgroup
hipError_t launch_decompress(const DevDFrame *frames, uint64_t n_frames, const uint32_t *ids, int cls,
                             uint32_t count, uint32_t lds, const DevPlan *plans,
                             const float2 *twpool, const uint8_t *body, double *out, int *status,
                             hipStream_t s)
{
    (void)n_frames;
    if (count == 0) return hipSuccess;
    switch (cls) {
    case 0: return launch_d<1, 2>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    case 1: return launch_d<1, 5>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    case 2: return launch_d<1, 9>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    case 3: return launch_d<4, 5>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    case 4: return launch_d<4, 9>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    case 5: return launch_d<16, 5>(count, lds, frames, ids, plans, twpool, body, out, status, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace atsc
