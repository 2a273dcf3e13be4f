// atsc_device.h -- device-side helpers shared by the compress and decompress kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atsc_hip.h"
#include "atsc_internal.h"

#include <map>
#include <mutex>
#include <utility>

namespace atsc {

#define DEVI __device__ __forceinline__
// threadIdx.x through an opaque move: code built on it cannot be hoisted out of a loop over frames (k_compress_resident),
// where every per-lane address and constant would otherwise stay in registers across the whole frame body
template <int W = 0>
DEVI uint32_t tid_now()
{
    // one-wavefront workgroups: the lane count below this lane (two instructions) instead of the thread-id register,
    // which the resident kernel's register allocation would rather spill than keep for the whole loop
    uint32_t t = (W == 1) ? __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel) and size step instead of once per
// launch: in a process that has many code objects loaded (PyTorch) one such call was measured at ~5 ms, and
// the large tier made six per batch.
inline hipError_t ensure_dyn_lds(const void *fn, uint32_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, uint32_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    uint32_t &have = done[std::make_pair(dev, fn)];
    if (bytes <= have) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// --------------------------------------------------------------------------------------------
// Rust `as` casts (saturating, NaN -> 0) and bincode varints
// --------------------------------------------------------------------------------------------
DEVI int64_t sat_i64(double x)
{
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
DEVI int32_t sat_i32(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
DEVI int32_t sat_i16(double x)
{
    if (x != x) return 0;
    if (x >= 32767.0) return 32767;
    if (x <= -32768.0) return -32768;
    return (int32_t)x;
}
DEVI uint32_t sat_u8(double x)
{
    if (x != x) return 0;
    if (x >= 255.0) return 255;
    if (x <= 0.0) return 0;
    return (uint32_t)x;
}
DEVI uint32_t vlen(uint64_t v)
{
    return v < 251 ? 1u : v < (1ull << 16) ? 3u : v < (1ull << 32) ? 5u : 9u;
}
DEVI uint64_t zigzag(int64_t v)
{
    return v >= 0 ? ((uint64_t)v << 1) : ((~(uint64_t)v << 1) | 1ull);
}
DEVI uint32_t put_varint(uint8_t *p, uint64_t v)
{
    if (v < 251) {
        p[0] = (uint8_t)v;
        return 1;
    }
    uint32_t nb;
    if (v < (1ull << 16)) { p[0] = 251; nb = 2; }
    else if (v < (1ull << 32)) { p[0] = 252; nb = 4; }
    else { p[0] = 253; nb = 8; }
    for (uint32_t i = 0; i < nb; ++i) p[1 + i] = (uint8_t)(v >> (8 * i));
    return nb + 1;
}
DEVI void put_u32(uint8_t *p, uint32_t u)
{
    p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24);
}
DEVI void put_f32(uint8_t *p, float f) { put_u32(p, __float_as_uint(f)); }
DEVI void put_u64(uint8_t *p, uint64_t u)
{
    put_u32(p, (uint32_t)u);
    put_u32(p + 4, (uint32_t)(u >> 32));
}
DEVI void put_f64(uint8_t *p, double d) { put_u64(p, (uint64_t)__double_as_longlong(d)); }

// optimizer/utils.rs:115-160  split_n: integer part and "fraction is non-zero"
DEVI void split_n(double x, int64_t &ip, bool &frac_nz)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(x);
    const bool neg = ((int64_t)bits) < 0;
    const int32_t exponent = (int32_t)((uint32_t)(bits >> 52) & 0x7ffu);
    const uint64_t mant_u = (bits & ((1ull << 52) - 1)) | (1ull << 52);
    const int64_t mant = neg ? -(int64_t)mant_u : (int64_t)mant_u;
    const int32_t shl = exponent + (64 - 53 - 1023 + 1);
    if (shl <= 0) {
        const int32_t shr = -shl;
        ip = 0;
        frac_nz = (shr < 64) ? ((((uint64_t)mant) >> shr) != 0) : false;
    } else if (shl < 64) {
        ip = mant >> (64 - shl);
        frac_nz = (((uint64_t)mant) << shl) != 0;
    } else if (shl < 128) {
        ip = (int64_t)(((uint64_t)mant) << (shl - 64));
        frac_nz = false;
    } else {
        ip = 0;
        frac_nz = false;
    }
}
// optimizer/utils.rs:91-113 ; wire ids F64 0, I32 1, I16 2, U8 3
DEVI uint32_t bitdepth_of(int64_t max_int, int64_t min_int)
{
    const int bd = max_int <= 255 ? 8 : max_int <= 32767 ? 16 : max_int <= 2147483647LL ? 32 : 64;
    const int bs = (min_int >= 0 && min_int <= 255) ? 8
                   : min_int >= -32768              ? 16
                   : min_int >= -2147483648LL       ? 32
                                                    : 64;
    const int m = bd > bs ? bd : bs;
    return m == 8 ? 3u : m == 16 ? 2u : m == 32 ? 1u : 0u;
}
// bytes one value takes at a bitdepth (constant.rs:43-60, polynomial.rs:61-81, rle.rs:46-63)
DEVI uint32_t value_bytes(uint32_t bitdepth, double v)
{
    if (bitdepth == 3) return 1;
    if (bitdepth == 2) return vlen(zigzag((int64_t)sat_i16(v)));
    if (bitdepth == 1) return vlen(zigzag((int64_t)sat_i32(v)));
    return 8;
}
DEVI uint32_t put_value(uint8_t *p, uint32_t bitdepth, double v)
{
    if (bitdepth == 3) { p[0] = (uint8_t)sat_u8(v); return 1; }
    if (bitdepth == 2) return put_varint(p, zigzag((int64_t)sat_i16(v)));
    if (bitdepth == 1) return put_varint(p, zigzag((int64_t)sat_i32(v)));
    put_f64(p, v);
    return 8;
}

// --------------------------------------------------------------------------------------------
// wavefront reductions on the DPP crossbar (no LDS): quad_perm, row_ror, row_bcast; the total
// lands in lane 63 and is broadcast through an SGPR (v_readlane).
// --------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
DEVI uint32_t dpp_u32(uint32_t v)  // lanes without a source read 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
// value of lane `l` (a compile-time lane) in every lane
DEVI double lane_f64(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// lane i receives lane i - 1's value (whole wavefront: DPP wave_shr:1); lane 0 keeps its own
DEVI double wave_shr1_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = (int)(uint32_t)b, hi = (int)(uint32_t)((unsigned long long)b >> 32);
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)h2 << 32) | l2));
}
template <int CTRL, int ROWMASK>
DEVI uint32_t dpp_u32_keep(uint32_t v, uint32_t ident)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROWMASK, 0xf, false);
}
DEVI uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_u32<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_u32<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_u32<0x124, 0xf>(v);  // row_ror:4
    v += dpp_u32<0x128, 0xf>(v);  // row_ror:8
    v += dpp_u32<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_u32<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum across the wavefront: Hillis-Steele inside each row of 16 (row_shr), then the
// row totals ripple through row_bcast:15 / row_bcast:31 -- six v_add_u32_dpp, no LDS crossbar
DEVI uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += dpp_u32<0x111, 0xf>(v);  // row_shr:1
    v += dpp_u32<0x112, 0xf>(v);  // row_shr:2
    v += dpp_u32<0x114, 0xf>(v);  // row_shr:4
    v += dpp_u32<0x118, 0xf>(v);  // row_shr:8
    v += dpp_u32<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_u32<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
DEVI uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<0xb1, 0xf>(v));
    v = max(v, dpp_u32<0x4e, 0xf>(v));
    v = max(v, dpp_u32<0x124, 0xf>(v));
    v = max(v, dpp_u32<0x128, 0xf>(v));
    v = max(v, dpp_u32<0x142, 0xa>(v));
    v = max(v, dpp_u32<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
DEVI uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }
template <int CTRL, int ROWMASK>
DEVI double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// (quad_perm / row_ror give every lane a source: v_mov_b32_dpp without an `old` operand to initialise)
template <int CTRL>
DEVI double dpp_f64_full(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
DEVI double wave_sum_f64(double v)
{
    v += dpp_f64_full<0xb1>(v);
    v += dpp_f64_full<0x4e>(v);
    v += dpp_f64_full<0x124>(v);
    v += dpp_f64_full<0x128>(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// for (j = tid; j < n; j += T) f(j), unrolled: FN trips for a compile-time length FN, else MAXIT predicated
// trips (n <= MAXIT * T by the caller's contract; MAXIT = 0: plain loop).  Unrolled, the loads of all trips
// issue together instead of one LDS round trip per (divergent) loop trip.
template <int FN, int T, int MAXIT = 0, typename F>
DEVI void for_strided(uint32_t tid, uint32_t n, F &&f)
{
    if constexpr (FN != 0) {
#pragma unroll
        for (int m = 0; m < (FN + T - 1) / T; ++m) {
            const uint32_t j = tid + (uint32_t)(m * T);
            if (FN % T == 0 || j < (uint32_t)FN) f(j);
        }
    } else if constexpr (MAXIT != 0) {
#pragma unroll
        for (int m = 0; m < MAXIT; ++m) {
            const uint32_t j = tid + (uint32_t)(m * T);
            if (j < n) f(j);
        }
    } else {
        for (uint32_t j = tid; j < n; j += T) f(j);
    }
}

// --------------------------------------------------------------------------------------------
// workgroup collectives (W wavefronts).  All return the same bits in every thread.
// --------------------------------------------------------------------------------------------
template <int W>
DEVI double block_sum_f64(double v, double *red, int &parity)
{
    v = wave_sum_f64(v);
    if (W == 1) return v;
    double *r = red + parity * 16;
    parity ^= 1;
    if ((tid_now<W>() & 63) == 0) r[tid_now<W>() >> 6] = v;
    __syncthreads();
    double s = r[0];
#pragma unroll
    for (int w = 1; w < W; ++w) s += r[w];
    return s;
}
template <int W>
DEVI uint32_t block_sum_u32(uint32_t v, double *red, int &parity)
{
    v = wave_sum_u32(v);
    if (W == 1) return v;
    uint32_t *r = (uint32_t *)(red + parity * 16);
    parity ^= 1;
    if ((tid_now<W>() & 63) == 0) r[tid_now<W>() >> 6] = v;
    __syncthreads();
    uint32_t s = r[0];
#pragma unroll
    for (int w = 1; w < W; ++w) s += r[w];
    return s;
}

// wavefront min / max of f64.  The lane values come out of the reference's scan (strict compares from data[0]):
// either none of them is a NaN or all of them are (data[0] is one), so fmin / fmax return what a compare-select
// chain would; which zero of +0.0 / -0.0 survives is open either way and the caller looks the first one up.
// Lanes without a DPP source read themselves.
template <int CTRL, int ROWMASK>
DEVI double dpp_f64_self(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool IsMin>
DEVI double wave_minmax_f64(double v)
{
#define ATSC_MM_STEP(O)                                          \
    {                                                            \
        const double o = O;                                      \
        v = IsMin ? fmin(o, v) : fmax(o, v);                     \
    }
    ATSC_MM_STEP(dpp_f64_full<0xb1>(v))
    ATSC_MM_STEP(dpp_f64_full<0x4e>(v))
    ATSC_MM_STEP(dpp_f64_full<0x124>(v))
    ATSC_MM_STEP(dpp_f64_full<0x128>(v))
    ATSC_MM_STEP((dpp_f64_self<0x142, 0xa>(v)))
    ATSC_MM_STEP((dpp_f64_self<0x143, 0xc>(v)))
#undef ATSC_MM_STEP
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// The statistics of a multi-wavefront frame in ONE exchange: minimum, maximum, "any fractional sample" and the packed
// run count / index bytes of the RLE bound (rle.rs:142-189).  Four block reductions one after the other are four
// barriers of the whole workgroup, each followed by every thread reading all W partials; here every wavefront leaves
// its four partials, one barrier, lane l combines the partials of wavefront l & 15 inside its row of 16 lanes (four DPP
// steps per value), and a second barrier frees the scratch.  Min / max / or / integer sum: nothing depends on the order.
// red: 32 doubles, wsum: 32 words behind them (the layout of the block_* helpers).  W <= 16.
struct BlockStats {
    double mn, mx;
    uint32_t frac, pk;
};
template <int W>
DEVI BlockStats block_stats4(double mn, double mx, uint32_t frac, uint32_t pk, double *red, uint32_t *wsum)
{
    static_assert(W >= 2 && W <= 16, "block_stats4: 2 .. 16 wavefronts");
    const uint32_t tid = tid_now<W>(), lane = tid & 63u, wv = tid >> 6;
    mn = wave_minmax_f64<true>(mn);
    mx = wave_minmax_f64<false>(mx);
    const uint32_t wfr = __ballot(frac != 0) ? 1u : 0u;
    pk = wave_sum_u32(pk);
    if (lane == 0) { red[wv] = mn; red[16 + wv] = mx; wsum[wv] = wfr; wsum[16 + wv] = pk; }
    __syncthreads();
    const uint32_t src = lane & 15u;
    const bool valid = src < (uint32_t)W;
    double a = red[valid ? src : 0u], b = red[16 + (valid ? src : 0u)];
    uint32_t f = valid ? wsum[src] : 0u, k = valid ? wsum[16 + src] : 0u;
#define ATSC_ROW_STEP(CTRL)                                            \
    {                                                                  \
        const double oa = dpp_f64_full<CTRL>(a), ob = dpp_f64_full<CTRL>(b); \
        a = fmin(oa, a);                                               \
        b = fmax(ob, b);                                               \
        f |= dpp_u32<CTRL, 0xf>(f);                                    \
        k += dpp_u32<CTRL, 0xf>(k);                                    \
    }
    ATSC_ROW_STEP(0xb1)
    ATSC_ROW_STEP(0x4e)
    ATSC_ROW_STEP(0x124)
    ATSC_ROW_STEP(0x128)
#undef ATSC_ROW_STEP
    __syncthreads();
    // (every lane holds the same four values; said so -- readfirstlane -- because everything the frame derives from them
    // (the clamp range, the bit depth, the RLE bound, payload sizes) is scalar work only if the compiler knows: taken
    // from DPP results it kept all of it in vector registers, 87 -> 117 VGPRs for the 1024-sample instantiation)
    BlockStats r;
    r.mn = lane_f64(a, 0);
    r.mx = lane_f64(b, 0);
    r.frac = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
    r.pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    return r;
}
template <int W, bool IsMin>
DEVI double block_minmax_f64(double v, double *red, int &parity)
{
    v = wave_minmax_f64<IsMin>(v);
    if (W == 1) return v;
    double *r = red + parity * 16;
    parity ^= 1;
    if ((tid_now<W>() & 63) == 0) r[tid_now<W>() >> 6] = v;
    __syncthreads();
    double s = r[0];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const double o = r[w];
        s = (IsMin ? (o < s) : (o > s)) ? o : s;
    }
    return s;
}
template <int W>
DEVI uint32_t block_min_u32(uint32_t v, double *red, int &parity)
{
    v = wave_min_u32(v);
    if (W == 1) return v;
    uint32_t *r = (uint32_t *)(red + parity * 16);
    parity ^= 1;
    if ((tid_now<W>() & 63) == 0) r[tid_now<W>() >> 6] = v;
    __syncthreads();
    uint32_t s = r[0];
#pragma unroll
    for (int w = 1; w < W; ++w) s = min(s, r[w]);
    return s;
}
template <int W>
DEVI uint32_t block_or_u32(uint32_t v, double *red, int &parity)
{
    if (W == 1) return __ballot(v != 0) ? 1u : 0u;  // one compare instead of a six-step DPP sum
    return block_sum_u32<W>(v ? 1u : 0u, red, parity) ? 1u : 0u;
}
// "fractional part is non-zero" exactly as optimizer/utils.rs:115-160 split_n(x).1 != 0.0 decides
// it, without the bit surgery: checked against the literal port on 16.4 M bit patterns covering
// every exponent (subnormals, |x| >= 2^52, NaN, Inf, both signs).
DEVI bool frac_nonzero(double x)
{
    return (x != trunc(x)) && (x == x) && (x >= 0x1p-64 || x <= -0x1p-75);
}

// x mod L with a precomputed magic = floor(2^32 / L) + 1 (L >= 2, any 32-bit x): the estimated
// quotient is exact or one too large.
DEVI uint32_t mod_magic(uint32_t x, uint32_t L, uint32_t magic)
{
    uint32_t r = x - __umulhi(x, magic) * L;
    if (r >= L) r += L;
    return r;
}
// a / d for small non-negative integers a <= d <= 4096 (spline parameter r / gap), correctly
// rounded: q = a*y with y = RN(1/d), one FMA residual step.  Checked exhaustively for all pairs.
DEVI double div_small(double a, double d, double y)
{
    const double q = a * y;
    const double rem = fma(-q, d, a);
    return fma(rem, y, q);
}

// In-place exclusive scan of an LDS u32 array; returns the total.  Each thread owns a
// contiguous chunk; partials are scanned across the workgroup.
template <int W>
DEVI uint32_t block_excl_scan(uint32_t *arr, uint32_t count, uint32_t *wsum)
{
    constexpr int T = 64 * W;
    const int tid = tid_now<W>();
    if (W == 1 && count <= 64) {  // one entry per lane: the payload emitters and the few-runs RLE
        const uint32_t v = (uint32_t)tid < count ? arr[tid] : 0u;
        const uint32_t in = wave_incl_scan_u32(v);
        if ((uint32_t)tid < count) arr[tid] = in - v;
        __syncthreads();
        return (uint32_t)__builtin_amdgcn_readlane((int)in, 63);
    }
    const uint32_t C = (count + T - 1) / T;
    const uint32_t b = min((uint32_t)tid * C, count), e = min(b + C, count);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += arr[i];
    uint32_t incl = wave_incl_scan_u32(s);
    if (W > 1) {
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
        incl += base;
    }
    if (tid == T - 1) wsum[W] = incl;
    uint32_t run = incl - s;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t v = arr[i];
        arr[i] = run;
        run += v;
    }
    __syncthreads();
    const uint32_t total = wsum[W];
    __syncthreads();
    return total;
}

// Ascending-only bitonic network for any count (elements >= count act as +inf and never
// move).  KeyOnly sorts u64 keys; otherwise a u32 payload array moves with the keys and
// breaks ties (key, payload).
template <int W, bool KeyOnly>
DEVI void block_sort(uint64_t *keys, uint32_t *pay, uint32_t count, uint32_t P2)
{
    constexpr int T = 64 * W;
    const uint32_t tid = tid_now<W>();
    const uint32_t npairs = P2 >> 1;
    auto ce = [&](uint32_t i, uint32_t l) {
        if (l < count) {
            const uint64_t a = keys[i], b = keys[l];
            bool sw = a > b;
            if (!KeyOnly) {
                const uint32_t pa = pay[i], pb = pay[l];
                sw = sw || (a == b && pa > pb);
                if (sw) { pay[i] = pb; pay[l] = pa; }
            }
            if (sw) { keys[i] = b; keys[l] = a; }
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

// --------------------------------------------------------------------------------------------
// forward transform, complex f32, unnormalised, e^{-i...}  (rustfft plan_fft_forward): Stockham autosort of length
// M = P.M, radices 4/2/3 from the plan; tw[t] = (cos, sin)(2 pi t / L) is the length-L table, so the stage twiddle
// w_M^e is tw[e * P.sc].  Returns the result buffer.  Shared by the encoder (atsc_kernels.hip) and the decoder's
// inverse transform (atsc_decode.hip).
// --------------------------------------------------------------------------------------------
DEVI float2 cmul_conj_tw(float2 v, float2 w)  // v * (w.x - i w.y)
{
    return make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
}
template <int W>
DEVI float2 *fft_forward(const DevPlan &P, float2 *X, float2 *Y, const float2 *tw)
{
    constexpr int T = 64 * W;
    const uint32_t M = P.M, sc = P.sc;
    uint32_t ncur = M, st = 1;
    for (uint32_t s = 0; s < P.nstages; ++s) {
        const uint32_t r = P.radix[s];
        const uint32_t m = ncur / r;
        const uint32_t nb = M / r;
        const uint32_t magic = P.stmagic[s];
        const uint32_t sm = st * m;
        for (uint32_t t = tid_now<W>(); t < nb; t += T) {
            const uint32_t p = (st == 1) ? t : __umulhi(t, magic);  // t / st
            const uint32_t q = t - p * st;
            const uint32_t ib = q + st * p;        // + st*m*j
            const uint32_t ob = q + st * (r * p);  // + st*k
            const uint32_t tb = p * st * sc;       // twiddle index step per k
            if (r == 4) {
                const float2 a0 = X[ib], a1 = X[ib + sm], a2 = X[ib + 2 * sm], a3 = X[ib + 3 * sm];
                const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y);
                const float2 t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
                const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
                const float2 d = make_float2(a1.x - a3.x, a1.y - a3.y);
                const float2 t3 = make_float2(d.y, -d.x);  // d * (-i)
                Y[ob] = make_float2(t0.x + t2.x, t0.y + t2.y);
                Y[ob + st] = cmul_conj_tw(make_float2(t1.x + t3.x, t1.y + t3.y), tw[tb]);
                Y[ob + 2 * st] = cmul_conj_tw(make_float2(t0.x - t2.x, t0.y - t2.y), tw[2 * tb]);
                Y[ob + 3 * st] = cmul_conj_tw(make_float2(t1.x - t3.x, t1.y - t3.y), tw[3 * tb]);
            } else if (r == 2) {
                const float2 a0 = X[ib], a1 = X[ib + sm];
                Y[ob] = make_float2(a0.x + a1.x, a0.y + a1.y);
                Y[ob + st] = cmul_conj_tw(make_float2(a0.x - a1.x, a0.y - a1.y), tw[tb]);
            } else {
                const float2 a0 = X[ib], a1 = X[ib + sm], a2 = X[ib + 2 * sm];
                const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
                const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
                const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
                const float h = 0.8660254037844386f;
                const float2 t3 = make_float2(h * d.y, -h * d.x);  // -i * h * d
                Y[ob] = make_float2(a0.x + t1.x, a0.y + t1.y);
                Y[ob + st] = cmul_conj_tw(make_float2(t2.x + t3.x, t2.y + t3.y), tw[tb]);
                Y[ob + 2 * st] = cmul_conj_tw(make_float2(t2.x - t3.x, t2.y - t3.y), tw[2 * tb]);
            }
        }
        __syncthreads();
        float2 *tmp = X; X = Y; Y = tmp;
        ncur = m;
        st *= r;
    }
    return X;
}

// --------------------------------------------------------------------------------------------
// Catmull-Rom / linear piecewise evaluation at integer x (splines 4.3.1 semantics, restated in
// oracle/atsc_oracle.c spline_clamped_sample) with the closed-form segment index.
// keys: T(k) = k*step for k < K-1, T(K-1) = n-1.   polynomial.rs:329-373
// --------------------------------------------------------------------------------------------
template <typename KnotFn>
DEVI double spline_eval(KnotFn V, uint32_t i, uint32_t n, uint32_t step, uint32_t K, uint32_t magic)
{
    // V(k): value of knot k;  knot k sits at x = k*step, the last one (k = K-1) at x = n-1
    if (i == n - 1) return V(K - 1);
    uint32_t seg = (step == 1) ? i : __umulhi(i, magic);  // i / step (magic = 2^32/step + 1, step >= 2)
    if (seg > K - 2) seg = K - 2;
    const uint32_t t0i = seg * step;
    const uint32_t t1i = (seg + 1 == K - 1) ? (n - 1) : (seg + 1) * step;
    const double t0 = (double)t0i, t1 = (double)t1i;
    const double v0 = V(seg), v1 = V(seg + 1);
    const double nt = ((double)i - t0) / (t1 - t0);
    if (seg > 0 && K - seg > 2) {
        const uint32_t tmi = (seg - 1) * step;
        const uint32_t tpi = (seg + 2 == K - 1) ? (n - 1) : (seg + 2) * step;
        const double tm = (double)tmi, tp = (double)tpi;
        const double vm = V(seg - 1), vp = V(seg + 2);
        const double t2 = nt * nt;
        const double t3 = t2 * nt;
        const double two_t3 = t3 * 2.0;
        const double two_t2 = t2 * 2.0;
        const double three_t2 = t2 * 3.0;
        const double m0 = (v1 - vm) / (t1 - tm) * (t1 - t0);
        const double m1 = (vp - v0) / (tp - t0) * (t1 - t0);
        return v0 * (two_t3 - three_t2 + 1.0) + m0 * (t3 - two_t2 + nt) + v1 * (three_t2 - two_t3) +
               m1 * (t3 - t2);
    }
    return v0 * (1.0 - nt) + v1 * nt;
}

// r / 100000.0 for the 5-decimal rounding (utils/mod.rs:61-74, fft.rs:208-218), correctly rounded
// without the ~30-instruction IEEE divide: q = r*y, one FMA residual correction (Markstein).
// Checked against r / 100000.0 for every integer |r| <= 2^27 and 4e8 random integers up to 2^53.
DEVI double div1e5(double r)
{
    const double y = 1.0e-5;
    const double q = r * y;
    const double rem = fma(-q, 100000.0, r);
    return fma(rem, y, q);
}

// 1 / |g| for the MAPE weights (utils/error.rs:104-116 divides by |g|; the product form is within
// an ulp of it).  v_rcp_f64 + two Newton steps instead of the ~30-instruction IEEE divide, for
// magnitudes where neither the seed nor the residuals leave the normal range; zero (-> inf),
// subnormal, huge, inf and NaN inputs take the divide.
DEVI double recip_abs(double g)
{
    const double a = fabs(g);
    if (a > 1e-290 && a < 1e290) {
        double r = __builtin_amdgcn_rcp(a);
        r = fma(fma(-a, r, 1.0), r, r);
        r = fma(fma(-a, r, 1.0), r, r);
        return r;
    }
    return 1.0 / a;
}

// --------------------------------------------------------------------------------------------
// Order of admission among bins of EQUAL norm.  FFT::fft_trim (fft.rs:231-257) pops its top-K from a
// std::collections::BinaryHeap built over all bins in position order; FrequencyPoint's Ord looks at the
// f32 norm only (fft.rs:88-106), so bins whose norms are bit-equal leave the heap in an order that
// depends on the heap's internals (Rust 1.81: rebuild = sift_down_range from len/2 - 1 down to 0;
// pop = swap_remove(0) + sift_down_to_bottom(0) + sift_up), and the order is serialised.  The sorted
// / lazy selection of the kernels is that pop sequence whenever the admitted norms are distinct; when
// they are not, the frame replays the heap itself: entries (norm bits << 32 | position) in LDS, every
// lane of one wavefront running the same sequential code (oracle/atsc_oracle.c heap_* is the same
// algorithm on the CPU).  Compares are the reference's: a <= b is !(na > nb), a >= b is na >= nb.
// --------------------------------------------------------------------------------------------
DEVI float hp_norm(uint64_t e) { return __uint_as_float((uint32_t)(e >> 32)); }
DEVI void hp_sift_down_range(uint64_t *h, uint32_t pos, uint32_t end)
{
    const uint64_t elt = h[pos];
    const float en = hp_norm(elt);
    uint32_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
        const uint64_t c0 = h[child], c1 = h[child + 1];
        const bool right = !(hp_norm(c0) > hp_norm(c1));  // hole.get(child) <= hole.get(child + 1)
        const uint64_t c = right ? c1 : c0;
        child += right ? 1u : 0u;
        if (en >= hp_norm(c)) { h[pos] = elt; return; }
        h[pos] = c;
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1) {
        const uint64_t c = h[child];
        if (!(en >= hp_norm(c))) { h[pos] = c; pos = child; }  // hole.element() < hole.get(child)
    }
    h[pos] = elt;
}
DEVI void hp_rebuild(uint64_t *h, uint32_t len)
{
    for (uint32_t n = len / 2; n > 0;) {
        --n;
        hp_sift_down_range(h, n, len);
    }
}
// The same rebuild by `nthreads` threads of a workgroup (all of them call it): BinaryHeap::rebuild sifts nodes
// len / 2 - 1 ... 0 down one after the other, i.e. every node of a level after all nodes of the levels below it, and
// the sift of a node touches its own subtree only -- the nodes of one level are independent, level by level.
DEVI void hp_rebuild_parallel(uint64_t *h, uint32_t len, uint32_t tid, uint32_t nthreads)
{
    if (len < 2) return;
    const uint32_t last_internal = len / 2 - 1;
    int level = 31 - __clz((int)(last_internal + 1));  // level of the last internal node (root = level 0)
    for (; level >= 0; --level) {
        const uint32_t first = (1u << level) - 1u;
        const uint32_t last = min(last_internal, (2u << level) - 2u);
        for (uint32_t node = first + tid; node <= last; node += nthreads) hp_sift_down_range(h, node, len);
        __syncthreads();
    }
}
// BinaryHeap::pop; the popped entry is also left at h[len - 1] (the slot the heap gives up), so K pops of
// a heap of `len` entries leave the pop sequence at h[len - 1], h[len - 2], ...
DEVI uint64_t hp_pop(uint64_t *h, uint32_t &len)
{
    uint64_t item = h[--len];
    if (len > 0) {
        const uint64_t top = h[0];
        const float en = hp_norm(item);
        // sift_down_to_bottom(0)
        uint32_t pos = 0, child = 1;
        while (len >= 2 && child <= len - 2) {
            const uint64_t c0 = h[child], c1 = h[child + 1];
            const bool right = !(hp_norm(c0) > hp_norm(c1));
            h[pos] = right ? c1 : c0;
            pos = child + (right ? 1u : 0u);
            child = 2 * pos + 1;
        }
        if (child == len - 1) {
            h[pos] = h[child];
            pos = child;
        }
        // sift_up(0, pos)
        while (pos > 0) {
            const uint32_t parent = (pos - 1) >> 1;
            const uint64_t pe = h[parent];
            if (!(en > hp_norm(pe))) break;  // hole.element() <= hole.get(parent)
            h[pos] = pe;
            pos = parent;
        }
        h[pos] = item;
        item = top;
    }
    h[len] = item;
    return item;
}

struct Sel {
    uint32_t pos;
    float re, im;
};

// ---- bincode payload reader (one lane walks the varint fields) ----
struct Rd {
    const uint8_t *p;
    uint32_t len, pos;
    bool bad;
};
DEVI uint32_t rd_u8(Rd &r)
{
    if (r.pos + 1 > r.len) { r.bad = true; return 0; }
    return r.p[r.pos++];
}
DEVI uint64_t rd_le(Rd &r, uint32_t nb)
{
    if (r.pos + nb > r.len) { r.bad = true; return 0; }
    uint64_t v = 0;
    for (uint32_t i = 0; i < nb; ++i) v |= (uint64_t)r.p[r.pos + i] << (8 * i);
    r.pos += nb;
    return v;
}
DEVI uint64_t rd_varint(Rd &r)
{
    const uint32_t t = rd_u8(r);
    if (t < 251) return t;
    if (t == 251) return rd_le(r, 2);
    if (t == 252) return rd_le(r, 4);
    if (t == 253) return rd_le(r, 8);
    r.bad = true;
    return 0;
}
DEVI int64_t unzig(uint64_t u) { return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1); }
// value at a bitdepth: constant.rs:71-92, polynomial.rs:95-116, rle.rs:76-100
DEVI double rd_value(Rd &r, uint32_t bd)
{
    if (bd == 3) return (double)rd_u8(r);
    if (bd == 2) return (double)(int16_t)unzig(rd_varint(r));
    if (bd == 1) return (double)(int32_t)unzig(rd_varint(r));
    return __longlong_as_double((long long)rd_le(r, 8));
}
DEVI float rd_f32(Rd &r) { return __uint_as_float((uint32_t)rd_le(r, 4)); }


// Payload reader of the decoders.  A payload's varint fields are sequential by construction (up to
// 13100 FFT entries, 131072 Noop values), and a byte at a time out of global memory costs a memory round
// trip per dependent read.  The first wavefront walks the payload in lock step (every lane computes the
// same thing) and keeps a window of it in LDS, refilled by the 64 lanes together.
struct RdS {
    const uint8_t *g;  // payload in global memory
    uint32_t len, pos;
    bool bad;
    uint8_t *stg;      // LDS window of `cap` bytes (cap >= 16)
    uint32_t w0, w1;   // payload bytes [w0, w1) are staged
    uint32_t cap;
};
DEVI void rds_fill(RdS &r)
{
    const uint32_t lane = threadIdx.x & 63;
    r.w0 = r.pos;
    const uint32_t nbytes = min(r.len - r.w0, r.cap);
    for (uint32_t o = lane * 4; o < nbytes; o += 64 * 4) {  // byte-granular source alignment: 4 bytes per lane
        uint32_t v = 0;
        const uint32_t left = min(nbytes - o, 4u);
        for (uint32_t b = 0; b < left; ++b) v |= (uint32_t)r.g[r.w0 + o + b] << (8 * b);
        *(uint32_t *)(r.stg + o) = v;
    }
    r.w1 = r.w0 + nbytes;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
DEVI uint32_t rds_u8(RdS &r)
{
    if (r.pos + 1 > r.len) { r.bad = true; return 0; }
    if (r.pos >= r.w1) rds_fill(r);
    return r.stg[r.pos++ - r.w0];
}
DEVI uint64_t rds_le(RdS &r, uint32_t nb)
{
    if (r.pos + nb > r.len) { r.bad = true; return 0; }
    if (r.pos + nb > r.w1) rds_fill(r);  // nb <= 8 << window
    uint64_t v = 0;
    for (uint32_t i = 0; i < nb; ++i) v |= (uint64_t)r.stg[r.pos + i - r.w0] << (8 * i);
    r.pos += nb;
    return v;
}
DEVI uint64_t rds_varint(RdS &r)
{
    const uint32_t t = rds_u8(r);
    if (t < 251) return t;
    if (t == 251) return rds_le(r, 2);
    if (t == 252) return rds_le(r, 4);
    if (t == 253) return rds_le(r, 8);
    r.bad = true;
    return 0;
}
DEVI double rds_value(RdS &r, uint32_t bd)
{
    if (bd == 3) return (double)rds_u8(r);
    if (bd == 2) return (double)(int16_t)unzig(rds_varint(r));
    if (bd == 1) return (double)(int32_t)unzig(rds_varint(r));
    return __longlong_as_double((long long)rds_le(r, 8));
}
DEVI float rds_f32(RdS &r) { return __uint_as_float((uint32_t)rds_le(r, 4)); }

// `cnt` consecutive varints, 64 at a time.  Lane l starts from the sum of the widths of the lanes
// before it, each width read off the marker byte at that lane's assumed start; the assumption is
// iterated to its fixed point (lane 0 is right at once, lane k after at most k more rounds; a run of
// equal widths settles in one or two).  emit(i, v) runs on the lane that decoded value i.
template <class Emit>
DEVI void rds_varints(RdS &r, uint32_t cnt, Emit emit)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t done = 0, excl = lane;
    while (done < cnt && !r.bad) {
        const uint32_t grp = min(64u, cnt - done);
        if (r.pos + 9 * 64 > r.w1 && r.w1 < r.len) rds_fill(r);
        uint32_t first = 0, wd = 0, incl = 0;
        for (int it = 0; it < 66; ++it) {
            const uint32_t st = r.pos + excl;
            first = (lane < grp && st < r.len) ? r.stg[st - r.w0] : 0u;
            wd = lane >= grp ? 0u : first < 251 ? 1u : first == 251 ? 3u : first == 252 ? 5u : first == 253 ? 9u : 1u;
            incl = wave_incl_scan_u32(wd);
            const uint32_t ne = incl - wd;
            const bool moved = ne != excl;
            excl = ne;
            if (__ballot(moved) == 0) break;
        }
        bool lbad = false;
        uint64_t v = first;
        if (lane < grp) {
            const uint32_t st = r.pos + excl;
            if (st + wd > r.len || first > 253) {
                lbad = true;
            } else if (wd > 1) {
                v = 0;
                for (uint32_t b = 0; b + 1 < wd; ++b) v |= (uint64_t)r.stg[st + 1 + b - r.w0] << (8 * b);
            }
        }
        if (__ballot(lbad)) { r.bad = true; break; }
        if (lane < grp) emit(done + lane, v);
        r.pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        done += grp;
    }
}

// `cnt` FFT entries -- varint(pos: u16) + f32 + f32 = 9 bytes (pos < 251) or 11 (marker 251 + 2) -- 64 at a
// time: lane l assumes 11-byte entries before it, corrected by c = the number of 9-byte entries among
// them; c is the exclusive prefix sum of the 9-byte flags read at the assumed starts, iterated to its
// fixed point (lane 0 is right at once, lane k after at most k more rounds; short entries are rare, so
// one or two rounds usually).  Entries land in ent[] with the position mirrored into [0, L/2]
// (get_mirrored_freqs, fft.rs:401-422: a position above L/2 is the mirror of L - pos) and the purely
// real bins' imaginary part cleared; a position >= L marks the stream bad.  Needs cap >= 11 * 64 + 16.
DEVI void rds_fft_entries(RdS &r, uint32_t cnt, uint32_t L, Sel *ent)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t done = 0;
    while (done < cnt && !r.bad) {
        const uint32_t grp = min(64u, cnt - done);
        if (r.pos + 11 * 64 > r.w1 && r.w1 < r.len) rds_fill(r);
        uint32_t c = 0, st = 0, first = 251, incl = 0;
        for (int it = 0; it < 66; ++it) {
            st = r.pos + 11 * lane - 2 * c;
            first = (lane < grp && st < r.len) ? r.stg[st - r.w0] : 251u;
            const uint32_t sf = (lane < grp && first < 251) ? 1u : 0u;
            incl = wave_incl_scan_u32(sf);
            const uint32_t cn = incl - sf;
            const bool moved = cn != c;
            c = cn;
            if (__ballot(moved) == 0) break;
        }
        bool lbad = false;
        if (lane < grp) {
            uint32_t pos = first, o = st + 1;
            if (first == 251) {
                if (st + 3 > r.len) lbad = true;
                else pos = (uint32_t)r.stg[st + 1 - r.w0] | ((uint32_t)r.stg[st + 2 - r.w0] << 8);
                o = st + 3;
            } else if (first > 251) {
                lbad = true;  // a u16 field: 4- and 8-byte varints cannot occur
            }
            if (!lbad && o + 8 > r.len) lbad = true;
            if (!lbad) {
                uint32_t wre = 0, wim = 0;
                for (uint32_t b = 0; b < 4; ++b) {
                    wre |= (uint32_t)r.stg[o + b - r.w0] << (8 * b);
                    wim |= (uint32_t)r.stg[o + 4 + b - r.w0] << (8 * b);
                }
                float re = __uint_as_float(wre), im = __uint_as_float(wim);
                if (pos >= L) {
                    lbad = true;
                } else {
                    if (pos > L / 2) { pos = L - pos; im = -im; }
                    if (pos == 0 || 2 * pos == L) im = 0.0f;
                    ent[done + lane].pos = pos;
                    ent[done + lane].re = re;
                    ent[done + lane].im = im;
                }
            }
        }
        if (__ballot(lbad)) { r.bad = true; break; }
        r.pos += 11 * grp - 2 * (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        done += grp;
    }
}


}  // namespace atsc
