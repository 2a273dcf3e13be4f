// atsc_large_fast.h -- included by atsc_large.hip (inside namespace atsc, after k_compress_large).
//
// Fast path of the large tier for frames whose transform splits as M = 243 x 288 (131072 samples, the frame the
// reference chunker cuts long series into, optimizer/mod.rs:78-98) under the auto selector (frame/mod.rs:71-149).
// k_compress_large runs a whole frame -- statistics, three candidates, every ladder trip, the payload -- in one
// 1024-thread workgroup whose live state no register file holds (88-183 spilled VGPRs).  Most frames are decided by
// the FIRST trip of each ladder: the polynomial's (k_large_poly1, a grid) and the FFT's, whose expensive part -- the
// inverse transform and the error sum -- is a grid as well (k_large_trip243 below).  What is left per frame is two
// short decision steps:
//   k_large_decide1  statistics -> Constant shortcut; polynomial trip 1 (sums of k_large_poly1); RLE bound; if the
//                    FFT's first payload can still win: the K1 = min(mf, Z) largest norms as a SET (histogram on the
//                    norm's top 11 bits, collection, radix select among the candidates of the threshold digit -- no
//                    sort: the trip's reconstruction does not depend on the order), the u16-wrap owners, the list of
//                    packed-spectrum points bucketed by k mod 243
//   k_large_trip243  (tile, frame) grid: 16 output columns per workgroup, direct sum over the buckets, 243-point
//                    transforms in registers (as k_large_cols243), evaluation against the padded samples
//   k_large_decide2  the trip's error, the selector, the payload (FFT: its bins sorted by norm now, once)
// A frame these steps cannot finish (a ladder that needs a second trip and is not pruned, an RLE bound that could
// still win, zero or NaN extremes, ...) is left untouched for k_compress_large<0>, launched behind them for the
// frames not marked finished: every frame takes the same path whatever batch it arrives in.
// Pruning is the selector's own arithmetic (a ladder's payload only grows, so a ladder stops once its next payload
// cannot beat a candidate that already passes): the winner and its bytes are the reference's.

constexpr uint32_t FAST_MF = 243, FAST_MD = 288;  // FAST_MD: the widest row dimension (131072 samples)
constexpr uint32_t FAST_CAND_MAX = 6144;   // candidates of the threshold digit held in LDS
constexpr uint32_t FAST_K_MAX = 1344;      // K1 <= mf = n / 100 = 1310
constexpr uint32_t FAST_OWN = 4608;        // positions below bins - 65536 = 4449 can collide after `pos as u16`
constexpr uint32_t FAST_LIST_OFF = 0;        // inside buffer B: the list, then the bucket bounds, then the sort keys
constexpr uint32_t FAST_PARTIAL_OFF = 8192;  // inside buffer C (as TRIP_PARTIAL_OFF)
constexpr uint32_t FAST_RLE_ENC_MAX = 3072;  // runs k_large_decide1 sorts and sizes in LDS (index bytes < 2^16, groups < 2^16)
constexpr uint32_t FAST_RLE_DEC_MAX = 6144;  // runs of an RLE frame k_large_dparse sorts in LDS (as RLE_LDS_RUNS)

// Geometry of a frame this path serves: M = 243 x md, md = 9 P, P = 2^lg = 2 .. 32 (8192 .. 131072 samples); ceil(md / 16)
// tiles of 16 output columns; where the bucket bounds and the payload-order keys sit behind the list in buffer B (the
// list holds at most two 12-byte points per admitted bin).
struct FastGeo {
    uint32_t md, lg, tiles;
    bool ok;
};
// behind a list of the points of K bins (two 12-byte points each at most): the bucket bounds (3 x 256 words), then the
// encoder's payload-order keys
DEVI uint32_t fast_bounds_off(uint32_t K) { return (24u * K + 255u) & ~255u; }
DEVI uint32_t fast_keys_off(uint32_t K) { return fast_bounds_off(K) + 4096u; }
DEVI FastGeo fast_geo(const DevPlan &P)
{
    FastGeo g;
    g.md = P.f4_m2;
    const uint32_t p9 = P.f4_m2 / 9u;
    g.ok = P.f4_m1 == FAST_MF && P.half && p9 * 9u == P.f4_m2 && p9 >= 2 && p9 <= 32 && (p9 & (p9 - 1)) == 0 && P.mf <= 1344;
    g.lg = 31u - (uint32_t)__clz((int)(p9 | 1u));
    g.tiles = (P.f4_m2 + 15u) / 16u;
    return g;
}
// x mod (9 . 2^lg) for x < (9 . 2^lg)^2
DEVI uint32_t fast_mod_md(uint32_t x, uint32_t lg)
{
    const uint32_t t = x >> lg, qq = (t * 7282u) >> 16;  // t < 81 . 2^lg <= 2592: t / 9 exactly
    return ((t - 9u * qq) << lg) | (x & ((1u << lg) - 1u));
}


// k_large_decide1's LDS carve-up, chosen per launch from its longest frame: the arrays a 131072-sample frame needs
// (owner table of the u16 wrap, 1344 keys, 6144 candidates: 124 KB, one workgroup per CU) would leave a launch of
// 8192-sample frames -- 81 bins to admit each -- at one frame per CU.  own[] and cand[] also host the two list buffers
// of the bucket builder (two 12-byte points per bin: 3 mf u64 each).
struct FastCarve {
    uint32_t own_n, k_max, cand_max;  // u64 entries
};
inline FastCarve fast_carve(uint32_t m2_max)
{
    FastCarve c;
    if (m2_max >= FAST_MD) { c.own_n = FAST_OWN; c.k_max = FAST_K_MAX; c.cand_max = FAST_CAND_MAX; return c; }
    const uint32_t mf_bound = 5u * m2_max;  // mf = n / 100 < 2 . 243 . m2 / 100
    c.k_max = (mf_bound + 63u) & ~63u;
    c.own_n = (3u * mf_bound + 63u) & ~63u;
    c.cand_max = c.own_n > 4096u ? c.own_n : 4096u;  // (2048: a fifth of the 8192-sample frames of the mixed workload overflow it)
    return c;
}
inline uint32_t fast_d1_lds(const FastCarve &c) { return 512 + 16384 + 8 * (c.own_n + 2 * c.k_max + c.cand_max); }

DEVI void fast_emit_poly(uint8_t *out, DevResult &r, const double *xs, uint32_t n, uint32_t bitdepth, uint32_t K,
                         uint32_t step, double smin, double smax, double err, uint32_t *aux, uint32_t *wsum)
{
    // polynomial.rs:54-87 (as k_compress_large's emitter)
    const uint32_t tid = threadIdx.x;
    const uint32_t hdr = 2 + vlen(K);
    uint32_t body;
    if (bitdepth == 0 || bitdepth == 3) {
        const uint32_t vbytes = bitdepth == 0 ? 8u : 1u;
        body = K * vbytes;
        for (uint32_t k = tid; k < K; k += LT) {
            const uint32_t t = (k == K - 1) ? (n - 1) : k * step;
            put_value(out + hdr + k * vbytes, bitdepth, xs[t]);
        }
    } else {
        for (uint32_t k = tid; k < K; k += LT) {
            const uint32_t t = (k == K - 1) ? (n - 1) : k * step;
            aux[k] = value_bytes(bitdepth, xs[t]);
        }
        __syncthreads();
        body = block_excl_scan<LW>(aux, K, wsum);
        for (uint32_t k = tid; k < K; k += LT) {
            const uint32_t t = (k == K - 1) ? (n - 1) : k * step;
            put_value(out + hdr + aux[k], bitdepth, xs[t]);
        }
    }
    if (tid == 0) {
        out[0] = 0;  // PolynomialType::Polynomial
        out[1] = (uint8_t)bitdepth;
        put_varint(out + 2, K);
        put_f64(out + hdr + body, smin);
        put_f64(out + hdr + body + 8, smax);
        out[hdr + body + 16] = (uint8_t)step;
        r.err = err;
        r.len = hdr + body + 17;
        r.chosen = ATSC_POLYNOMIAL;
    }
}

// The packed-spectrum points of a list of bins, bucketed by k mod 243 and laid out for k_large_trip243 (see
// sparse_bucket): entry(i, p, x) gives bin position p (< M) and value x of list entry i, false if the entry is void.
// K <= 2 LT entries.  LDS: zl / zs two regions of 2 K SpEnt each, tab 1024 u32.  Leaves the list, the bucket bounds and
// the buckets' order by size in buffer B; returns the number of points.  Every bucket ends up in ascending (kb, kind)
// order -- the f32 sums of the tiles do not depend on the order the atomics happened to serve -- by counting: a
// point's place is the number of smaller keys in its bucket (keys are distinct), one thread per point.  (An insertion
// sort by one thread per bucket took 70 us on frames whose bins sit on few residues: a signal of period 64 puts every
// harmonic on a multiple of L / 64 = 9 * 243.)
template <class EntryFn>
DEVI uint32_t fast_bucket(uint32_t K, EntryFn entry, const float2 *tw, uint32_t M, SpEnt *zl, SpEnt *zs, uint32_t *tab,
                          uint32_t *bcw, unsigned char *Bb, uint32_t bounds_off)
{
    const uint32_t tid = threadIdx.x;
    uint32_t *beg = tab, *end = tab + 256, *cnt = tab + 512, *border = tab + 768;
    for (uint32_t e = tid; e < 256; e += LT) { beg[e] = 0; cnt[e] = 0; }
    __syncthreads();
    auto points = [&](uint32_t i, uint32_t (&kk)[2], float2 (&vv)[2]) -> uint32_t {
        uint32_t p;
        float2 x;
        if (!entry(i, p, x)) return 0;
        // bin p <= M feeds point p of the packed spectrum (p < M: bin M = L / 2, which the shorter frames can admit,
        // has no point of its own) and, as the conjugate partner, point M - p (p >= 1)
        const bool has1 = p < M, has2 = p >= 1;
        uint32_t k1 = 0, k2 = 0;
        float2 v1 = make_float2(0.0f, 0.0f), v2 = v1;
        if (has1) {
            const float2 h = make_float2(0.5f * x.x, 0.5f * x.y);
            const float2 o = cmulp(h, tw[p]);
            k1 = p;
            v1 = make_float2(h.x - o.y, -(h.y + o.x));
        }
        if (has2) {
            const float2 ee = make_float2(0.5f * x.x, -0.5f * x.y);
            const float2 d = make_float2(-0.5f * x.x, 0.5f * x.y);
            const float2 o = cmulp(d, tw[M - p]);
            k2 = (M - p) | 0x80000000u;
            v2 = make_float2(ee.x - o.y, -(ee.y + o.x));
        }
        // (slots by constant index: the arrays stay in registers)
        kk[0] = has1 ? k1 : k2;
        vv[0] = has1 ? v1 : v2;
        kk[1] = k2;
        vv[1] = v2;
        return (has1 ? 1u : 0u) + (has2 ? 1u : 0u);
    };
    uint32_t mykk[2][2];
    float2 myvv[2][2];
    uint32_t myc[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t i = tid + u * LT;
        if (i < K) {
            myc[u] = points(i, mykk[u], myvv[u]);
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                if (q < myc[u]) atomicAdd(&beg[(mykk[u][q] & 0x7fffffffu) % FAST_MF], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 243 bucket sizes by one wavefront (4 per lane)
        uint32_t c[4], s = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = beg[4 * tid + u]; s += c[u]; }
        const uint32_t incl = wave_incl_scan_u32(s);
        uint32_t run = incl - s;
#pragma unroll
        for (int u = 0; u < 4; ++u) { beg[4 * tid + u] = run; run += c[u]; }
        if (tid == 63) bcw[0] = incl;
    }
    __syncthreads();
    const uint32_t nlist = bcw[0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            if (q >= myc[u]) continue;
            const uint32_t k = mykk[u][q] & 0x7fffffffu;
            const uint32_t kb = k / FAST_MF, ka = k - kb * FAST_MF;
            const uint32_t slot = beg[ka] + atomicAdd(&cnt[ka], 1u);
            SpEnt z;
            z.key = (kb << 1) | (mykk[u][q] >> 31);
            z.re = myvv[u][q].x;
            z.im = myvv[u][q].y;
            zl[slot] = z;
        }
    __syncthreads();
    if (tid < FAST_MF) end[tid] = beg[tid] + cnt[tid];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            if (q >= myc[u]) continue;
            const uint32_t k = mykk[u][q] & 0x7fffffffu;
            const uint32_t kb = k / FAST_MF, ka = k - kb * FAST_MF;
            const uint32_t key = (kb << 1) | (mykk[u][q] >> 31);
            const uint32_t b = beg[ka], e = end[ka];
            uint32_t rank = 0;
            for (uint32_t t = b; t < e; ++t) rank += zl[t].key < key ? 1u : 0u;
            SpEnt z;
            z.key = key;
            z.re = myvv[u][q].x;
            z.im = myvv[u][q].y;
            zs[b + rank] = z;
        }
    // the buckets by descending size (ties by index): the tiles deal them over their thread groups in this order.
    // Rank by counting, four threads per bucket (a quarter of the 243 compares each; one thread per bucket walked
    // them all: 6 us of LDS round trips); the partial ranks meet in zl[], which is free once zs is built.
    __syncthreads();
    uint32_t *rk = (uint32_t *)zl;
    if (tid < 256) rk[tid] = 0;
    __syncthreads();
    if (tid < 4 * FAST_MF) {
        const uint32_t b = tid % FAST_MF, q = tid / FAST_MF;
        const uint32_t mine = cnt[b];
        const uint32_t t0 = q * 61u, t1 = min(FAST_MF, t0 + 61u);
        uint32_t rank = 0;
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t o = cnt[t];
            rank += (o > mine || (o == mine && t < b)) ? 1u : 0u;
        }
        if (rank) atomicAdd(&rk[b], rank);
    }
    __syncthreads();
    if (tid < FAST_MF) border[rk[tid]] = tid;
    __syncthreads();
    {
        uint32_t *gl = (uint32_t *)(Bb + FAST_LIST_OFF);
        const uint32_t *src = (const uint32_t *)zs;
        for (uint32_t w = tid; w < 3 * nlist; w += LT) gl[w] = src[w];
        uint32_t *gb = (uint32_t *)(Bb + bounds_off);
        for (uint32_t e = tid; e < FAST_MF; e += LT) { gb[e] = beg[e]; gb[256 + e] = end[e]; gb[512 + e] = border[e]; }
    }
    return nlist;
}

// BIG: the launch holds 131072-sample frames, whose carve-up leaves one workgroup per CU anyway: 128 VGPRs a lane, and the
// norm patterns of the counting pass stay in registers for the collection pass (one read of the 280 KB instead of two).
template <bool BIG>
__device__ __forceinline__ void large_decide1(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool, const KParams &prm,
    uint8_t *__restrict__ slots, DevResult *__restrict__ res, unsigned char *__restrict__ ws_base, uint64_t ws_stride,
    const FastCarve &cv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t fid = ids[blockIdx.x];
    const DevFrame fr = frames[fid];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = P.n, L = P.L, bins = P.bins, M = P.M;
    unsigned char *ws = ws_base + (uint64_t)blockIdx.x * ws_stride;
    const LargeWs lay = large_ws_layout(n, L, P.kcap);
    FastState *fs = (FastState *)(ws + lay.o_front);
    if (tid == 0) fs->status = 0;
    if (blockIdx.x == 0 && tid == 0) *fb_count(ws_base, ws_stride) = 0;  // the launch's list of frames left to the general kernel (k_large_decide2 fills it)
    // why a frame was left to the general kernel (read back under ATSC_DEBUG_STOP=-3 / -4 only)
#define FAST_WHY(c) do { if (tid == 0) *(uint32_t *)(ws + lay.o_front + 200) = (c); } while (0)
    FAST_WHY(0);
    // ATSC_DEBUG_STOP=-3: workgroup 0 prints the 100 MHz clock at its phase boundaries when it ends
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FSTAMP(i) do { if (prm.debug_stop <= -3) stamp[i] = wall_clock64(); } while (0)
#define FSTAMP_PRINT(tag) do { if (prm.debug_stop <= -3 && (blockIdx.x == 0 || prm.debug_stop == -4) && tid == 0) \
        printf("FSTAMP %s b%u %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n", tag, blockIdx.x, (double)(stamp[1] - stamp[0]) * 0.01, \
               (double)(stamp[2] - stamp[1]) * 0.01, (double)(stamp[3] - stamp[2]) * 0.01, (double)(stamp[4] - stamp[3]) * 0.01, \
               (double)(stamp[5] - stamp[4]) * 0.01, (double)(stamp[6] - stamp[5]) * 0.01, (double)(stamp[7] - stamp[6]) * 0.01); } while (0)
    FSTAMP(0);
    const FastGeo geo = fast_geo(P);
    const bool wrap = bins > 65536;  // `pos as u16` (fft.rs:242) can fold two bins onto one stored position
    if (!geo.ok || (wrap && bins - 65536 > cv.own_n) || P.mf > cv.k_max || 3 * P.mf > cv.own_n || 3 * P.mf > cv.cand_max) { FAST_WHY(1); return; }
    // LDS: [wsum 80][bc 64][red 256][h2 2048 u32][dcnt 2048 u32][own cv.own_n u64][above cv.k_max u64][sorted cv.k_max u64]
    // [cand cv.cand_max u64]  (the launch's carve-up, fast_carve: sized for its longest frame)
    uint32_t *wsum = (uint32_t *)smem;
    uint32_t *bc = (uint32_t *)(smem + 128);
    double *bcd = (double *)(smem + 192);
    uint32_t *h2 = (uint32_t *)(smem + 512);
    uint32_t *dcnt = (uint32_t *)(smem + 512 + 8192);  // 2048 per-digit cursors
    unsigned long long *own = (unsigned long long *)(smem + 512 + 16384);
    unsigned long long *above = own + cv.own_n;
    unsigned long long *sorted = above + cv.k_max;
    unsigned long long *cand = sorted + cv.k_max;
    const double *xs = samples + fr.sample_off;
    const float2 *tw = twpool + P.tw_off;
    uint8_t *out = slots + fr.slot_off;

    // ---- statistics (k_large_stats, or the column tiles' and polynomial chunks' records): optimizer/utils.rs:39-113 ----
    LargeStats *lst = (LargeStats *)(ws + lay.o_cnt);
    FrameStats st;
    // (everything the verdicts below read from memory is asked for here, in one round trip: the first sample, the count of
    // zero bins, the polynomial's chunk sums -- added in chunk order by one thread, polynomial.rs:209-277)
    const double x0 = xs[0];
    const uint32_t zero_bins = lst->zeros;
    if (tid == 0) {
        double s = 0.0;
        const double *part = (const double *)(ws + lay.o_part);
        for (uint32_t c = 0; c < (n + LCH - 1) / LCH; ++c) s += part[c];
        bcd[0] = s / (double)n;
    }
    if (prm.tile_stats) {
        // one record per lane, combined by wavefront reductions (nothing here depends on an order)
        double mn = __longlong_as_double(0x7ff0000000000000ll), mx = -mn;
        uint32_t fr_ = 0, ru = 0, ib = 0;
        if (tid < 64) {
            if (tid < (P.f4_m2 + FB - 1) / FB) {
                const TileStats q = ((const TileStats *)(ws + lay.o_tst))[tid];
                mn = q.mn; mx = q.mx; fr_ = q.frac;
            }
            if (tid < (n + LCH - 1) / LCH) {
                const uint2 pc = ((const uint2 *)(ws + lay.o_part + 1536))[tid];
                ru = pc.x; ib = pc.y;
            }
            mn = wave_minmax_f64<true>(mn);
            mx = wave_minmax_f64<false>(mx);
            fr_ = __ballot(fr_ != 0) ? 1u : 0u;
            ru = wave_sum_u32(ru);
            ib = wave_sum_u32(ib);
            if (tid == 0) {
                bcd[1] = mn; bcd[2] = mx;
                bc[12] = fr_; bc[13] = ru; bc[14] = ib;
                // k_compress_large<0>, behind this path, reads the combined record (`zeros` is the row pass's)
                lst->frac = fr_; lst->runs = ru; lst->ibytes = ib;
                lst->kmin = f64_key(mn); lst->kmax = f64_key(mx);
            }
        }
        __syncthreads();
        st.mn = bcd[1]; st.mx = bcd[2]; st.frac = bc[12]; st.runs = bc[13]; st.ibytes = bc[14];
    } else {
        st = frame_stats(ws, lay, false, 0);
    }
    if (!(x0 == x0)) { FAST_WHY(2); return; }  // a NaN first sample keeps the scan's start value: left to the general kernel
    const double smin = st.mn, smax = st.mx;
    if (smin == 0.0 || smax == 0.0) { FAST_WHY(3); return; }  // the first zero of either sign has to be looked up
    uint32_t bitdepth;
    {
        int64_t maxi, mini;
        bool fz;
        split_n(smax, maxi, fz);
        split_n(smin, mini, fz);
        bitdepth = st.frac ? 0u : bitdepth_of(maxi, mini);
    }
    // forced `--compressor fft` / `--compressor polynomial` (main.rs:150-162) on this path: the same first trips, no
    // competition -- the codec is accepted with whatever error its bounded ladder ends on (compressor/mod.rs:86-98)
    const bool forced_fft = prm.mode == ATSC_FFT, forced_poly = prm.mode == ATSC_POLYNOMIAL;
    if (smin == smax && (forced_fft || forced_poly)) { FAST_WHY(5); return; }  // (flat frames: the general kernel's shortcuts)
    if (smin == smax) {  // frame/mod.rs:82-88
        if (tid == 0) {
            out[0] = 30;
            out[1] = (uint8_t)bitdepth;
            const uint32_t vb = put_value(out + 2, bitdepth, smin);
            res[fid].err = 0.0;
            res[fid].len = 2 + vb;
            res[fid].chosen = ATSC_CONSTANT;
            fs->status = 2;
        }
        return;
    }
    const double me = prm.max_err;
    uint32_t best_size = 0xFFFFFFFFu;
    int best_owner = 3;
    auto can_win = [&](uint32_t size_lb, int owner) {
        return size_lb < best_size || (size_lb == best_size && owner < best_owner);
    };
    auto offer = [&](uint32_t size, int owner) {
        if (can_win(size, owner)) { best_size = size; best_owner = owner; }
    };
    // ---- RLE bound (rle.rs:142-189): run count and index bytes come with the statistics ----
    const uint32_t rle_R = st.runs, rle_ib = st.ibytes;
    const uint32_t rle_lb = 3 + rle_ib + (rle_R >= 2 ? 2u : 1u) * ((bitdepth == 0 ? 8u : 1u) + 1);

    // ---- polynomial, first trip (polynomial.rs:209-277): the chunk sums of k_large_poly1, in chunk order ----
    const uint32_t pstep = P.pstep[0], pK = P.pK[0];
    if (!(pstep >= 16 && pstep <= 256 && pK >= 2)) { FAST_WHY(4); return; }
    uint32_t vb = 0;
    if (bitdepth == 0 || bitdepth == 3) {
        vb = pK * (bitdepth == 0 ? 8u : 1u);
        __syncthreads();
    } else {
        for (uint32_t k = tid; k < pK; k += LT) vb += value_bytes(bitdepth, xs[(k == pK - 1) ? (n - 1) : k * pstep]);
        int parity = 0;
        vb = block_sum_u32<LW>(vb, (double *)(smem + 256), parity);  // (its barrier publishes bcd[0] too)
    }
    const double pcur = bcd[0];
    const bool poly_final = !(round(pcur * 10000.0) > prm.poly_q_hi);  // polynomial.rs:231
    const uint32_t poly_size = 2 + vlen(pK) + vb + 17;
    const uint32_t poly2_lb = 2 + vlen(P.pK[1]) + P.pK[1] * (bitdepth == 0 ? 8u : 1u) + 17;
    if (forced_poly) {
        // polynomial.rs:209-277: the ladder ends with its first trip, or it goes on in the general kernel
        if (!poly_final) { FAST_WHY(12); return; }
        fast_emit_poly(out, res[fid], xs, n, bitdepth, pK, pstep, smin, smax, pcur, h2, wsum);
        if (tid == 0) fs->status = 2;
        return;
    }
    if (!forced_fft && poly_final && pcur <= me) offer(poly_size, 1);

    // ---- FFT (fft.rs:288-362) ----
    const float mxf = (float)smax, mnf = (float)smin;
    if (mxf == mnf) { FAST_WHY(5); return; }
    const uint32_t Z = bins - zero_bins;
    const uint32_t K1 = min(P.mf, Z);
    if (K1 < 8) { FAST_WHY(6); return; }
    const uint32_t fft1_size = 1 + vlen(K1) + 9 * K1 + 8;  // the least the FFT candidate can store

    // ---- RLE, exactly (rle.rs:142-189), when its bound can still win and the runs fit the LDS ----
    // The run starts come as a bit map from the polynomial pieces (poly1_piece); the runs are sorted by (value bits,
    // start) = the reference's BTreeMap order and sized.  RLE reports error 0.0 and so passes whenever 0.0 <= max_error
    // (the fast path's condition).  When its payload beats the least either ladder can still store the frame is decided
    // here: no norm is selected, no transform evaluated -- a gauge's 131072-sample frame used to wait for the general
    // kernel (156 us for six of them).  Otherwise nothing changes: the exact size is not offered.
    uint32_t rle_lb_eff = rle_lb;  // the exact size once it is known: the tightest bound the later verdicts can use
    {
        const uint32_t lds64 = cv.own_n + 2 * cv.k_max + cv.cand_max;       // u64 entries from own[] on
        const uint32_t rcap = min(FAST_RLE_ENC_MAX, (8u * lds64) / 28u);     // 28 bytes of LDS per run
        if (!forced_fft && prm.tile_stats && rle_R >= 1 && rle_R <= rcap && can_win(rle_lb, 2) && rle_lb < fft1_size &&
            (poly_final || rle_lb < poly2_lb)) {
            const uint32_t R = rle_R;
            unsigned long long *kk = own;                 // value bits of run i
            uint32_t *pp = (uint32_t *)(kk + rcap);       // run index in start order (the sort's tie-break)
            uint32_t *rst = pp + rcap;                    // start of run i (start order)
            uint32_t *aux = rst + rcap;                   // heads / scans
            uint32_t *hp = aux + rcap;                    // first sorted run of group g (hp[D] = R): rcap + 1 <= rcap + 8
            uint32_t *rph = h2;                           // header bytes of group g (2048 + 2048 words: h2, dcnt)
            const unsigned long long *bm = (const unsigned long long *)(ws + lay.o_rbm);
            const uint32_t nw = n >> 6;                   // 64-sample words (n is a multiple of 64 on this path)
            // starts in order: every thread expands the words tid, tid + LT, ... -- counts first, a scan over the
            // threads' word slices would reorder them, so the words are taken in contiguous slices of wpt per thread
            const uint32_t wpt = (nw + LT - 1) / LT;
            uint32_t cnt = 0;
            for (uint32_t w = tid * wpt; w < min(nw, (tid + 1) * wpt); ++w) cnt += (uint32_t)__popcll(bm[w]);
            dcnt[tid] = cnt;
            __syncthreads();
            const uint32_t total = block_excl_scan<LW>(dcnt, LT, wsum);
            if (total != R) { FAST_WHY(15); return; }  // (cannot happen: the pieces counted the same bits)
            {
                uint32_t o = dcnt[tid];
                for (uint32_t w = tid * wpt; w < min(nw, (tid + 1) * wpt); ++w) {
                    unsigned long long m = bm[w];
                    while (m) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        rst[o++] = 64u * w + b;
                    }
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < R; i += LT) {
                kk[i] = (unsigned long long)__double_as_longlong(xs[rst[i]]);
                pp[i] = i;
            }
            __syncthreads();
            uint32_t p2 = 1;
            while (p2 < R) p2 <<= 1;
            block_sort<LW, false>((uint64_t *)kk, pp, R, p2);
            for (uint32_t i = tid; i < R; i += LT) aux[i] = (i == 0 || kk[i] != kk[i - 1]) ? 1u : 0u;
            __syncthreads();
            const uint32_t D = block_excl_scan<LW>(aux, R, wsum);
            if (D <= 4096) {
                for (uint32_t i = tid; i < R; i += LT)
                    if (i == 0 || kk[i] != kk[i - 1]) hp[aux[i]] = i;
                if (tid == 0) hp[D] = R;
                __syncthreads();
                uint32_t hb = 0;
                for (uint32_t gi = tid; gi < D; gi += LT) {
                    const uint32_t b = value_bytes(bitdepth, __longlong_as_double((long long)kk[hp[gi]])) + vlen(hp[gi + 1] - hp[gi]);
                    rph[gi] = b;
                    hb += b;
                }
                {
                    int parity = 0;
                    hb = block_sum_u32<LW>(hb, (double *)(smem + 256), parity);
                }
                const uint32_t rle_size = 2 + vlen(D) + hb + rle_ib;
                rle_lb_eff = rle_size;
                // frame/mod.rs:113-147: the smallest passing payload, the first of [FFT, Polynomial, RLE] on ties
                const bool beats_poly = can_win(rle_size, 2) && (poly_final || rle_size < poly2_lb);
                if (beats_poly && rle_size < fft1_size) {
                    // emit (rle.rs:40-67): per group its value and run count, then the starts of its runs
                    const uint32_t hdr = 2 + vlen(D);
                    // one scan for both prefixes: (group heads before i) << 16 | index varint bytes before i
                    for (uint32_t i = tid; i < R; i += LT) {
                        const bool head = (i == 0 || kk[i] != kk[i - 1]);
                        aux[i] = (head ? 0x10000u : 0u) | vlen(rst[pp[i]]);
                    }
                    __syncthreads();
                    (void)block_excl_scan<LW>(aux, R, wsum);
                    const uint32_t hbt = block_excl_scan<LW>(rph, D, wsum);
                    for (uint32_t i = tid; i < R; i += LT) {
                        const uint32_t st = rst[pp[i]];
                        const bool head = (i == 0 || kk[i] != kk[i - 1]);
                        const uint32_t pk = aux[i], ps = pk & 0xffffu;
                        const uint32_t gi = head ? (pk >> 16) : (pk >> 16) - 1;
                        const uint32_t ghb = (gi + 1 < D ? rph[gi + 1] : hbt);  // header bytes up to and incl. group gi
                        if (head) {
                            uint8_t *q = out + hdr + rph[gi] + ps;
                            q += put_value(q, bitdepth, __longlong_as_double((long long)kk[i]));
                            put_varint(q, hp[gi + 1] - hp[gi]);
                        }
                        put_varint(out + hdr + ghb + ps, st);
                    }
                    if (tid == 0) {
                        out[0] = 60;
                        out[1] = (uint8_t)bitdepth;
                        put_varint(out + 2, D);
                        res[fid].err = 0.0;
                        res[fid].len = rle_size;
                        res[fid].chosen = ATSC_RLE;
                        fs->status = 2;
                    }
                    return;
                }
            }
            __syncthreads();  // (own[] .. cand[] and h2 / dcnt are reused below)
        }
    }
    if (!can_win(fft1_size, 0)) {
        // even the first trip's payload loses to the polynomial, which passes: no transform is evaluated at all
        if (!(poly_final && pcur <= me) || can_win(rle_lb_eff, 2)) { FAST_WHY(7); return; }
        fast_emit_poly(out, res[fid], xs, n, bitdepth, pK, pstep, smin, smax, pcur, h2, wsum);
        if (tid == 0) fs->status = 2;
        return;
    }

    FSTAMP(1);  // 1: statistics, polynomial trip 1, bounds
    // ---- the K1 largest norms (fft.rs:231-257), as a set ----
    const uint32_t *nbits = (const uint32_t *)(ws + lay.o_nb);
    for (uint32_t i = tid; i < 2048; i += LT) { h2[i] = 0; dcnt[i] = 0; }
    // the counting pass adds into four copies of the counters, one per lane & 3 (a spectrum's norms crowd into a few
    // exponents: a wavefront's adds to one counter queue up in the LDS); cand[] is free until the collection pass
    // (long frames only: a short frame's pass is over before the 8192 extra counters are cleared and folded)
    const bool copies = bins >= 32768;
    uint32_t *hc = copies ? (uint32_t *)cand : h2;
    const uint32_t csh = copies ? 2u : 0u, cp = copies ? (tid & 3u) : 0u;
    if (copies)
        for (uint32_t i = tid; i < 8192; i += LT) hc[i] = 0;
    if (wrap)
        for (uint32_t i = tid; i < cv.own_n; i += LT) own[i] = 0ull;
    if (tid < 16) bc[tid] = 0;
    if (tid == 0) bc[4] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t nq = bins >> 2;  // whole uint4 groups (the norm-bit region is 256-byte aligned)
    constexpr uint32_t QPT = BIG ? 18 : 6;  // groups in flight per thread (BIG: the whole frame, 69 985 bins)
    const bool keep = BIG && nq <= QPT * LT;
    uint4 v[QPT];
    for (uint32_t q0 = 0; q0 < nq; q0 += QPT * LT) {
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t q = q0 + u * LT + tid;
            v[u] = q < nq ? ((const uint4 *)nbits)[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < QPT; ++u) {
            const uint32_t q = q0 + u * LT + tid;
            if (q < nq) {
                atomicAdd(&hc[((2047u - (v[u].x >> 20)) << csh) | cp], 1u);
                atomicAdd(&hc[((2047u - (v[u].y >> 20)) << csh) | cp], 1u);
                atomicAdd(&hc[((2047u - (v[u].z >> 20)) << csh) | cp], 1u);
                atomicAdd(&hc[((2047u - (v[u].w >> 20)) << csh) | cp], 1u);
            }
        }
    }
    for (uint32_t k = 4 * nq + tid; k < bins; k += LT) atomicAdd(&hc[((2047u - (nbits[k] >> 20)) << csh) | cp], 1u);
    __syncthreads();
    if (copies) {
        for (uint32_t i = tid; i < 2048; i += LT) {
            const uint4 c4 = ((const uint4 *)hc)[i];
            h2[i] = c4.x + c4.y + c4.z + c4.w;
        }
        __syncthreads();
    }
    FSTAMP(2);  // 2: histogram
    {
        const uint32_t c0 = h2[2 * tid], c1 = h2[2 * tid + 1];
        __syncthreads();
        (void)block_excl_scan<LW>(h2, 2048, wsum);  // h2[i] = bins with a digit above 2047 - i
        const uint32_t a0 = h2[2 * tid], a1 = h2[2 * tid + 1];
        if (a0 < K1 && a0 + c0 >= K1) { bc[4] = 2 * tid; bc[5] = a0; bc[6] = c0; }
        if (a1 < K1 && a1 + c1 >= K1) { bc[4] = 2 * tid + 1; bc[5] = a1; bc[6] = c1; }
        __syncthreads();
    }
    if (bc[4] == 0xFFFFFFFFu) { FAST_WHY(8); return; }
    const uint32_t dstar = 2047u - bc[4], n_above = bc[5], n_cand = bc[6];
    if (n_cand > cv.cand_max) { FAST_WHY(9); return; }
    {
        const uint32_t lane = tid & 63u;
        const uint64_t lt = (1ull << lane) - 1ull;
        auto visit = [&](uint32_t k, uint32_t v, bool in) {
            const uint32_t d = v >> 20;
            const bool ab = in && d > dstar, cd = in && d == dstar;
            const unsigned long long key = ((unsigned long long)(~v) << 32) | (unsigned long long)k;
            const uint64_t mc = __ballot(cd);
            // a bin above the threshold digit goes straight into its digit's stretch of the final order: h2[] (scanned)
            // holds the number of bins with a larger digit, i.e. where the stretch starts
            if (ab) above[h2[2047u - d] + atomicAdd(&dcnt[2047u - d], 1u)] = key;
            if (mc) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&bc[3], (uint32_t)__popcll(mc));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (cd) cand[base + (uint32_t)__popcll(mc & lt)] = key;
            }
        };
        for (uint32_t q0 = 0; q0 < nq; q0 += QPT * LT) {
            if (!keep) {
#pragma unroll
                for (uint32_t u = 0; u < QPT; ++u) {
                    const uint32_t q = q0 + u * LT + tid;
                    v[u] = q < nq ? ((const uint4 *)nbits)[q] : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < QPT; ++u) {
                const uint32_t q = q0 + u * LT + tid;
                const bool in = q < nq;
                // the large norms sit together (low frequencies): most wavefronts see four values below the digit
                const uint32_t top = max(max(v[u].x, v[u].y), max(v[u].z, v[u].w)) >> 20;
                if (!__ballot(in && top >= dstar)) continue;
                visit(4 * q, v[u].x, in);
                visit(4 * q + 1, v[u].y, in);
                visit(4 * q + 2, v[u].z, in);
                visit(4 * q + 3, v[u].w, in);
            }
        }
        for (uint32_t k0 = 4 * nq; k0 < bins; k0 += LT) {
            const uint32_t k = k0 + tid;
            visit(k, k < bins ? nbits[k] : 0u, k < bins);
        }
        __syncthreads();
    }
    // the payload lists the bins by descending norm, equal norms by position (fft.rs:119-130, 231-257; the large tier's
    // order): the stretches are in place, inside a stretch a key's place is the number of smaller keys in it
    for (uint32_t i = tid; i < n_above; i += LT) {
        const unsigned long long key = above[i];
        const uint32_t di = 2047u - ((~(uint32_t)(key >> 32)) >> 20);
        const uint32_t gb = h2[di], ge = gb + dcnt[di];
        uint32_t rank = 0;
        for (uint32_t j = gb; j < ge; ++j) rank += above[j] < key ? 1u : 0u;
        sorted[gb + rank] = key;
    }
    __syncthreads();
    FSTAMP(3);  // 3: digit, collection
    // the `take` smallest candidate keys (norm descending, position ascending): radix select of the take-th smallest
    // 37-bit key -- 20 norm bits below the digit, 17 position bits -- 11 bits a level
    const uint32_t take = K1 - n_above;  // 1 .. n_cand
    unsigned long long prefix = 0, resolved = 0;
    {
        uint32_t remaining = take;
        auto k37 = [](unsigned long long key) -> unsigned long long {
            return ((key >> 32) & 0xFFFFFull) << 17 | (key & 0x1FFFFull);
        };
        for (int shift = 26; shift >= -7; shift -= 11) {
            const int sh = shift < 0 ? 0 : shift;
            const uint32_t nb = shift < 0 ? 4u : 11u;  // the last level holds the 4 bits left
            const uint32_t dm = (1u << nb) - 1u;
            for (uint32_t i = tid; i < 2048; i += LT) h2[i] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < n_cand; i += LT) {
                const unsigned long long kk = k37(cand[i]);
                if ((kk & resolved) == prefix) atomicAdd(&h2[(uint32_t)(kk >> sh) & dm], 1u);
            }
            __syncthreads();
            const uint32_t c0 = h2[2 * tid], c1 = h2[2 * tid + 1];
            __syncthreads();
            (void)block_excl_scan<LW>(h2, 2048, wsum);  // h2[i] = keys with a smaller digit
            const uint32_t a0 = h2[2 * tid], a1 = h2[2 * tid + 1];
            // (bc[7]: every key of the digit is taken -- the bits below do not matter)
            if (c0 && a0 < remaining && a0 + c0 >= remaining) { bc[8] = 2 * tid; bc[9] = a0; bc[7] = (a0 + c0 == remaining) ? 1u : 0u; }
            if (c1 && a1 < remaining && a1 + c1 >= remaining) { bc[8] = 2 * tid + 1; bc[9] = a1; bc[7] = (a1 + c1 == remaining) ? 1u : 0u; }
            __syncthreads();
            prefix |= (unsigned long long)bc[8] << sh;
            resolved |= (unsigned long long)dm << sh;
            remaining -= bc[9];
            const bool whole = bc[7] != 0;
            __syncthreads();
#ifndef ATSC_NO_SEL_EXIT
            if (whole) {  // (distinct norms: the usual end, after the two levels that cover the norm's bits)
                prefix |= (1ull << sh) - 1ull;
                break;
            }
#endif
        }
        // keys are distinct: exactly `take` candidates have k37 <= prefix
    }
    FSTAMP(4);  // 4: radix select
    // ---- admission: sel[] (any order), sort keys for the payload, `pos as u16` owners, count of 3-byte positions ----
    Sel *sel = (Sel *)(ws + lay.o_sel);
    unsigned char *Bb = ws + lay.o_b;
    unsigned long long *skey = (unsigned long long *)(Bb + fast_keys_off(K1));
    const float2 *spec = (const float2 *)(ws + lay.o_a);
    uint32_t big = 0;
    unsigned long long ad_key[2];
    uint32_t ad_pos[2];
    float2 ad_z[2];
    if (tid == 0) bc[2] = 0;
    __syncthreads();
    {
        auto k37 = [](unsigned long long key) -> unsigned long long {
            return ((key >> 32) & 0xFFFFFull) << 17 | (key & 0x1FFFFull);
        };
        // the candidates that made it: behind the stretches, in order among themselves
        for (uint32_t i = tid; i < n_cand; i += LT) {
            const unsigned long long key = cand[i];
            if (k37(key) <= prefix) above[atomicAdd(&bc[2], 1u)] = key;  // (the stretches live in sorted[] now)
        }
        __syncthreads();
        if (bc[2] != take) {
#ifdef ATSC_SEL_DEBUG
            if (tid == 0) {
                uint32_t *dbg = (uint32_t *)(ws + lay.o_front + 204);
                dbg[0] = take; dbg[1] = bc[2]; dbg[2] = n_cand; dbg[3] = n_above; dbg[4] = (uint32_t)prefix; dbg[5] = (uint32_t)(prefix >> 32);
                dbg[6] = bc[7]; dbg[7] = bc[8]; dbg[8] = bc[9];
            }
#endif
            FAST_WHY(10); return; }  // (cannot happen: the select is exact)
        for (uint32_t i = tid; i < take; i += LT) {
            const unsigned long long key = above[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < take; ++j) rank += above[j] < key ? 1u : 0u;
            sorted[n_above + rank] = key;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // (K1 <= 2 LT; entry i = tid + u LT stays in this thread's registers for the bucket builder)
            const uint32_t i = tid + (uint32_t)u * LT;
            ad_key[u] = 0ull; ad_pos[u] = 0; ad_z[u] = make_float2(0.0f, 0.0f);
            if (i < K1) {
                const unsigned long long key = sorted[i];
                const uint32_t pos = (uint32_t)(key & 0xffffffffull);
                const float2 z = spec[pos];
                sel[i].pos = pos; sel[i].re = z.x; sel[i].im = z.y;
                skey[i] = key;
                ad_key[u] = key; ad_pos[u] = pos; ad_z[u] = z;
                const uint32_t p16 = pos & 0xffffu;
                big += p16 >= 251 ? 1u : 0u;
                if (wrap && p16 < cv.own_n) atomicMax(&own[p16], key);  // the later admission (the larger key) owns the position
            }
        }
    }
    {
        int parity = 0;
        big = block_sum_u32<LW>(big, (double *)(smem + 256), parity);
    }
    __syncthreads();
    FSTAMP(5);  // 5: admission
    // ---- the trip's packed-spectrum points, bucketed by k mod 243 ----
    // (the candidate list and the owners are done with once the points are formed: their LDS hosts the two list buffers)
    const uint32_t nlist = fast_bucket(
        K1,
        [&](uint32_t i, uint32_t &p, float2 &x) -> bool {  // (called with i = tid + u LT: this thread's own entries)
            const bool hi = i >= LT;
            const uint32_t e_pos = hi ? ad_pos[1] : ad_pos[0];
            const float2 e_z = hi ? ad_z[1] : ad_z[0];
            const unsigned long long e_key = hi ? ad_key[1] : ad_key[0];
            p = e_pos & 0xffffu;  // `pos as u16` (fft.rs:242): bins >= 65536 are stored and mirrored 65536 lower
            if (wrap && p < cv.own_n && own[p] != e_key) return false;
            x = (p == 0 || 2 * p == L) ? make_float2(e_z.x, 0.0f) : e_z;
            return true;
        },
        tw, M, (SpEnt *)cand, (SpEnt *)own, h2, bc + 10, Bb, fast_bounds_off(K1));  // (own[] is read by entry() before zs is first written)
    FSTAMP(6);  // 6: bucketing
    if (tid == 0) {
        FastState f;
        f.status = 1;
        f.bitdepth = bitdepth; f.K1 = K1; f.big = big; f.Z = Z;
        f.best_size = best_size; f.best_owner = best_owner;
        f.poly_final = poly_final ? 1u : 0u;
        f.forced = forced_fft ? 1u : 0u;
        f.poly_size = poly_size; f.poly_K = pK; f.poly_step = pstep; f.poly2_lb = poly2_lb; f.rle_lb = rle_lb_eff;
        f.nlist = nlist;
        f.smin = smin; f.smax = smax; f.poly_err = pcur;
        f.mxf = mxf; f.mnf = mnf;
        *fs = f;
    }
    FSTAMP(7);  // 7: list and state out
    FSTAMP_PRINT("decide1");
}

__global__ __launch_bounds__(LT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_large_decide1(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool, const KParams prm,
    uint8_t *__restrict__ slots, DevResult *__restrict__ res, unsigned char *__restrict__ ws_base, uint64_t ws_stride,
    const FastCarve cv)
{
    large_decide1<false>(samples, frames, ids, plans, twpool, prm, slots, res, ws_base, ws_stride, cv);
}

__global__ __launch_bounds__(LT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_large_decide1_big(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool, const KParams prm,
    uint8_t *__restrict__ slots, DevResult *__restrict__ res, unsigned char *__restrict__ ws_base, uint64_t ws_stride,
    const FastCarve cv)
{
    large_decide1<true>(samples, frames, ids, plans, twpool, prm, slots, res, ws_base, ws_stride, cv);
}

// One tile (16 output columns jb) of the first FFT trip of one frame: F[288 ja + jb] for every ja, compared with the
// padded samples 2 j, 2 j + 1 (j = 288 ja + jb); the tile's share of the MAPE sum goes to buffer C.
// Thread (column c, q): inputs ka = 9 a + q of its column straight from the buckets, then as k_large_cols243.
// DECODE: the same tile for k_large_dparse's list, writing the frame's decoded samples (fft.rs:426-462) instead of an
// error sum: FR = DevDFrame, io = the output buffer.
// LG: the frame's row dimension is md = 9 . 2^LG (LG = 5: 131072 samples); one instantiation per LG, launched when the
// batch holds such frames -- the workgroups of a launch leave the frames of another LG alone.  (One kernel with md at run
// time needs 170 VGPRs where the fixed form needs 161: two wavefronts per SIMD instead of three.)
// Threads of a tile's workgroup (TT): the column transforms and the evaluation keep the 192-thread geometry of
// k_large_cols243 (CT); the bucket sums in front of them -- 20 of a tile's 35 us, a chain of LDS round trips per wavefront
// -- are dealt over TT / 16 groups of 16 lanes, and the wavefronts past CT end with them.  Two forms: 512 threads for
// launches that leave the GPU part empty (32 frames of 131072 samples at e = 1 % are 360 working tiles on 256 CUs: a CU that
// gets two of the 192-thread tiles takes 61 us over them, one that gets one 35 -- 64 -> 45 us for the launch), 192 for
// launches that fill it.  Where the two cross (tools/trip_width_ab.sh, mixed classes at e = 5 %, us per launch, 192 | 512):
//   frames      32        64        80        128        256
//   encoder   52 | 45   74 | 77   75 | 85   115 | 132   196 | 246
//   decoder   52 | 46   83 | 72   98 | 83   132 | 122   223 | 222
#ifndef ATSC_TRIP_WIDE
#define ATSC_TRIP_WIDE 512
#endif
constexpr int TRIP_WIDE = ATSC_TRIP_WIDE;
constexpr uint32_t TRIP_WIDE_MAX_ENC = 900, TRIP_WIDE_MAX_DEC = 4096;  // workgroups of a launch up to which the wide form is taken
__device__ unsigned long long g_trip_log[1024][3];
__device__ unsigned long long g_trip_span[4] = {~0ull, 0ull, 0ull, 0ull};  // ATSC_DEBUG_STOP=-6: first start, last end, sum, tiles
template <bool DECODE, class FR, int LG, int TT>
__global__ __launch_bounds__(TT) void k_large_trip243(
    const double *__restrict__ samples, const FR *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const float2 *__restrict__ twpool, unsigned char *__restrict__ ws_base,
    uint64_t ws_stride, int dbg, double *__restrict__ outp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    unsigned long long tstamp[6] = {0, 0, 0, 0, 0, 0};
#define TSTAMP(i) do { if (dbg & 3) tstamp[i] = wall_clock64(); } while (0)
    TSTAMP(0);
    const FR fr = frames[ids[blockIdx.y]];
    const DevPlan &P = plans[fr.plan];
    unsigned char *ws = ws_base + (uint64_t)blockIdx.y * ws_stride;
    const LargeWs lay = large_ws_layout(P.n, P.L, P.kcap);
    const FastState *fs = (const FastState *)(ws + lay.o_front);
    if constexpr (DECODE) {
        // the frames k_large_dparse left alone, listed for k_decompress_large<0> (once: by the first launch, whatever its LG)
        if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && fs->status == 0) fb_append(ws_base, ws_stride, blockIdx.y);
    }
    if (fast_geo(P).lg != (uint32_t)LG) return;
    if constexpr (DECODE) {
        if (fs->status == 5) {  // a Constant frame: this workgroup fills piece blockIdx.x
            const uint32_t n = P.n, pieces = fast_geo(P).tiles;
            if (blockIdx.x >= pieces) return;
            const double v = fs->smin;
            double *out = frame_out(outp, fr);
            const uint32_t i0 = (uint32_t)(((uint64_t)n * blockIdx.x) / pieces);
            const uint32_t i1 = (uint32_t)(((uint64_t)n * (blockIdx.x + 1)) / pieces);
            for (uint32_t j = i0 + tid; j < i1; j += TT) out[j] = v;
            return;
        }
        if (fs->status == 4) {
            // An RLE frame k_large_dparse prepared: this workgroup writes piece blockIdx.x of the frame.  A wavefront takes
            // 64 runs at a time, one per lane (start, next start, value), and writes them one after the other, 64 samples a
            // store; samples in front of the first run are 0.0 (rle.rs:204-236).
            const uint32_t n = P.n, E = fs->K1;
            const uint32_t pieces = fast_geo(P).tiles;
            if (blockIdx.x >= pieces) return;
            const uint32_t *rstart = (const uint32_t *)(ws + lay.o_hp);
            const double *rval = (const double *)(ws + lay.o_rec);
            const uint32_t *ptab = (const uint32_t *)(ws + lay.o_part);
            double *out = frame_out(outp, fr);
            const uint32_t i0 = (uint32_t)(((uint64_t)n * blockIdx.x) / pieces);
            const uint32_t i1 = (uint32_t)(((uint64_t)n * (blockIdx.x + 1)) / pieces);
            const uint32_t c0 = ptab[blockIdx.x], c1 = ptab[blockIdx.x + 1];  // runs that start at or before i0 / i1
            const uint32_t lane = tid & 63u, wv = tid >> 6;
            if (c0 == 0) {
                const uint32_t z1 = min(i1, rstart[0]);
                for (uint32_t j = i0 + tid; j < z1; j += TT) out[j] = 0.0;
            }
            const uint32_t a = c0 ? c0 - 1 : 0;
            for (uint32_t base = a + 64u * wv; base < c1; base += 64u * (TT / 64)) {
                const uint32_t r = base + lane;
                uint32_t st = 0, en = 0;
                double v = 0.0;
                if (r < c1) {
                    st = rstart[r];
                    en = (r + 1 < E) ? rstart[r + 1] : n;
                    v = rval[r];
                }
                const uint32_t cc = min(64u, c1 - base);
                for (uint32_t u = 0; u < cc; ++u) {
                    const uint32_t s0 = max((uint32_t)__builtin_amdgcn_readlane((int)st, (int)u), i0);
                    const uint32_t e0 = min((uint32_t)__builtin_amdgcn_readlane((int)en, (int)u), i1);
                    const double vv = lane_f64(v, (int)u);
                    for (uint32_t j = s0 + lane; j < e0; j += 64) out[j] = vv;
                }
            }
            return;
        }
        if (fs->status == 3) {
            // A polynomial frame k_large_dparse prepared: this workgroup evaluates piece blockIdx.x of the frame's tiles
            // (polynomial.rs:342-373 with the encoder's tables -- tangents per segment, Hermite basis per in-segment
            // offset, exact r / step -- bit for bit the general decoder's values).
            const uint32_t n = P.n;
            const uint32_t K = fs->poly_K, step = fs->poly_step;
            const double mn = fs->smin, mx = fs->smax;
            const double *vals = (const double *)(ws + lay.o_tab);
            double *out = frame_out(outp, fr);
            double4 *hbt = (double4 *)smem;                 // step <= 255 entries
            double2 *mms = (double2 *)(smem + 8192);        // segments of this piece
            const uint32_t pieces = fast_geo(P).tiles;  // (the grid is as wide as the launch's widest frame)
            if (blockIdx.x >= pieces) return;
            const uint32_t i0 = (uint32_t)(((uint64_t)n * blockIdx.x) / pieces);
            const uint32_t i1 = (uint32_t)(((uint64_t)n * (blockIdx.x + 1)) / pieces);
            const uint32_t magic = (uint32_t)(0x100000000ull / step) + 1u;
            const uint32_t gapL = (n - 1) - (K - 2) * step;
            const double stepd = (double)step, gapLd = (double)gapL;
            const double ry = 1.0 / stepd, ryL = 1.0 / gapLd;
            uint32_t sgA = __umulhi(i0, magic), sgB = __umulhi(i1 - 1, magic);
            if (sgA > K - 2) sgA = K - 2;
            if (sgB > K - 2) sgB = K - 2;
            if (sgB - sgA + 1 > 1536) return;  // (step >= 16 and pieces of at most 7282 samples: at most 457 segments)
            for (uint32_t sg = sgA + tid; sg <= sgB; sg += TT) {
                double2 t = make_double2(0.0, 0.0);
                if (sg >= 1 && sg + 2 < K) {
                    const uint32_t t0i = sg * step;
                    const uint32_t t1i = (sg + 1 == K - 1) ? (n - 1) : (sg + 1) * step;
                    const uint32_t tmi = (sg - 1) * step;
                    const uint32_t tpi = (sg + 2 == K - 1) ? (n - 1) : (sg + 2) * step;
                    const double t0 = (double)t0i, t1 = (double)t1i;
                    const double v0 = vals[sg], v1 = vals[sg + 1], vm = vals[sg - 1], vp = vals[sg + 2];
                    t.x = (v1 - vm) / (t1 - (double)tmi) * (t1 - t0);
                    t.y = (vp - v0) / ((double)tpi - t0) * (t1 - t0);
                }
                mms[sg - sgA] = t;
            }
            for (uint32_t r = tid; r < step; r += TT) {
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
            for (uint32_t i = i0 + tid; i < i1; i += TT) {
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
                        const double2 t = mms[sg - sgA];
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
    }
    if (fs->status != 1) return;
    const FastGeo geo = fast_geo(P);
    if (blockIdx.x >= geo.tiles) return;  // (the grid is as wide as the launch's widest frame)
    constexpr uint32_t lg = LG, MD = 9u << LG;
    const uint32_t n = P.n, L = P.L, pre = P.pre;
    const float2 *tw = twpool + P.tw_off;
    const double *xs = frame_samples(samples, fr);
    // LDS: [T 243 x 17 points][w1 243][wd 288][beg 256][end 256][list]
    float2 *T = (float2 *)smem;
    float2 *w1 = T + 243 * CSI;
    float2 *wd = w1 + 243;
    uint32_t *beg = (uint32_t *)(wd + FAST_MD), *end = beg + 256, *border = end + 256;
    const unsigned char *Bb = ws + lay.o_b;
    const SpEnt *gl = (const SpEnt *)(Bb + FAST_LIST_OFF);  // the list stays in memory (L2: the frame's 18 tiles share it)
    const uint32_t nlist = fs->nlist;
    {
        const uint32_t *gb = (const uint32_t *)(Bb + fast_bounds_off(fs->K1));
        for (uint32_t e = tid; e < 256; e += TT) { beg[e] = gb[e]; end[e] = gb[256 + e]; border[e] = gb[512 + e]; }
        for (uint32_t e = tid; e < 243; e += TT) w1[e] = tw[e * (L / FAST_MF)];
        for (uint32_t e = tid; e < MD; e += TT) wd[e] = tw[e * (L / MD)];
    }
    const uint32_t c = tid & 15u, q = tid >> 4;
    const uint32_t jb_raw = blockIdx.x * 16 + c;
    const bool col_live = jb_raw < MD;           // (md = 18, 36, 72: the last tile is partly empty)
    const uint32_t jb = col_live ? jb_raw : MD - 1;  // dead columns compute a live column's values and drop them
    const bool live = q < 9;
    __syncthreads();
    TSTAMP(1);
    // Inputs of the column transforms: T[ka][c] = G[ka][jb] times W_M^{jb ka} (jb ka < M: no wrap).  The twelve groups
    // of 16 lanes (one DPP row each) take the buckets in the order of their size (largest first, round-robin): bins
    // that sit on few residues mod 243 would otherwise leave one group summing while the others wait.  A group reads
    // its bucket 16 entries at a time, one per lane, straight from memory (the next batch is requested before the
    // current one is used) and hands entry u to its 16 columns by a row broadcast: the list never sits in LDS, which
    // is what lets four of these workgroups share a CU.  The sum over a bucket runs in list order whichever group it
    // falls to (slots past the bucket's end add zeros).
    {
        const uint32_t grp = tid >> 4;  // 0 .. TT / 16 - 1

        auto fetch = [&](uint32_t e, uint32_t e1) -> SpEnt {
            SpEnt z;
            z.key = 0; z.re = 0.0f; z.im = 0.0f;
            if (e + c < e1) z = gl[e + c];
            return z;
        };
        uint32_t i = grp;
        uint32_t ka = border[i], e = beg[ka], e1 = end[ka];
        SpEnt nxt = fetch(e, e1);
        float2 twn = tw[jb * ka * P.sc];
        float2 acc = make_float2(0.0f, 0.0f);
        while (i < FAST_MF) {
            const SpEnt cur = nxt;
            const float2 twc = twn;
            const uint32_t kc = ka;
            const uint32_t cnt = min(16u, e1 - e);
            const bool last = e + 16 >= e1;
            // where the next batch comes from
            uint32_t ni = i;
            if (last) {
                ni = i + TT / 16;
                if (ni < FAST_MF) { ka = border[ni]; e = beg[ka]; e1 = end[ka]; twn = tw[jb * ka * P.sc]; }
            } else {
                e += 16;
            }
            if (ni < FAST_MF) nxt = fetch(e, e1);
            // (four slots at a time, run while any row of the wavefront still has an entry among them -- a uniform
            // branch; a slot past a bucket's end holds a zero entry.  A branch per slot kept the compiler from
            // requesting a slot's twiddle before the slot in front of it was summed: sixteen LDS round trips a batch
            // one after the other, 25 of a tile's 44 us.)
#define ATSC_BKT(U)                                                                                          \
            {                                                                                                     \
                const uint32_t ku = dpp_u32<0x150 + U, 0xf>(cur.key);                                             \
                const float ru = __uint_as_float(dpp_u32<0x150 + U, 0xf>(__float_as_uint(cur.re)));                \
                const float iu = __uint_as_float(dpp_u32<0x150 + U, 0xf>(__float_as_uint(cur.im)));                \
                const float2 t = cmulc(make_float2(ru, iu), wd[fast_mod_md(jb * (ku >> 1), lg)]);                          \
                acc.x += t.x;                                                                                     \
                acc.y += t.y;                                                                                     \
            }
            if (__ballot(0 < cnt)) { ATSC_BKT(0) ATSC_BKT(1) ATSC_BKT(2) ATSC_BKT(3) }
            if (__ballot(4 < cnt)) { ATSC_BKT(4) ATSC_BKT(5) ATSC_BKT(6) ATSC_BKT(7) }
            if (__ballot(8 < cnt)) { ATSC_BKT(8) ATSC_BKT(9) ATSC_BKT(10) ATSC_BKT(11) }
            if (__ballot(12 < cnt)) { ATSC_BKT(12) ATSC_BKT(13) ATSC_BKT(14) ATSC_BKT(15) }
#undef ATSC_BKT
            if (last) {
                T[kc * CSI + c] = cmulc(acc, twc);
                acc = make_float2(0.0f, 0.0f);
            }
            i = ni;
        }
    }
    __syncthreads();
    if (tid >= CT) return;  // the wavefronts that only sum buckets are done (a wavefront that has ended does not hold a barrier up)
    TSTAMP(2);
    float2 a[27];
#pragma unroll
    for (int aa = 0; aa < 27; ++aa) a[aa] = live ? T[(9u * aa + q) * CSI + c] : make_float2(0.0f, 0.0f);
    __syncthreads();  // every thread holds its inputs: T becomes the exchange buffer
    dft27f(a);
    if (live) {
#pragma unroll
        for (int ka = 1; ka < 27; ++ka) a[ka] = cmulc(a[ka], w1[q * ka]);
#pragma unroll
        for (int ka = 0; ka < 27; ++ka) T[(q * 27 + ka) * CSI + c] = a[ka];
    }
    __syncthreads();
    TSTAMP(3);
    if constexpr (DECODE) {
        // fft.rs:455-461: skip the padding, / L in f32, 5 decimals, clamp to the stored f32 range
        if (live) {
            double *out = frame_out(outp, fr);
            const double mxd = (double)fs->mxf, mnd = (double)fs->mnf;
            const float Lf = (float)L;
            const bool pair_ok = (((uintptr_t)out & 15u) == 0);
            auto val = [&](float re) -> double {
                const float v = re / Lf;
                double o = round((double)v * 100000.0) / 100000.0;
                if (o > mxd) o = mxd;
                if (o < mnd) o = mnd;
                return o;
            };
#pragma unroll 1
            for (uint32_t m = 0; m < 3; ++m) {
                const uint32_t ka = q + 9u * m;
                float2 b[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) b[j] = T[(j * 27 + ka) * CSI + c];
                dft9f(b);
#pragma unroll
                for (int kq = 0; kq < 9; ++kq) {  // idft_L = 2 idft_M: even sample -> re, odd sample -> -im
                    const uint32_t ja = ka + 27u * kq;
                    const int32_t j2 = (int32_t)(2 * (MD * ja + jb)) - (int32_t)pre;  // even (pre and n are)
                    if (col_live && j2 >= 0 && j2 < (int32_t)n) {
                        const double o0 = val(2.0f * b[kq].x), o1 = val(-2.0f * b[kq].y);
                        if (pair_ok) *(double2 *)(out + j2) = make_double2(o0, o1);
                        else { out[j2] = o0; out[j2 + 1] = o1; }
                    }
                }
            }
        }
        TSTAMP(4);
        return;
    }
    // Evaluation: ja = (q + 9 m) + 27 kq, samples 2 j and 2 j + 1, j = 288 ja + jb, against the padded signal (pre and n
    // are even for this class: a pair never straddles a frame edge).  The samples of group m + 1 are requested while
    // group m is transformed and evaluated.
    double s = 0.0;
    if (live && col_live) {
        const double mxd = (double)fs->mxf, mnd = (double)fs->mnf;
        const float Lf = (float)L;
        auto term = [&](float re, double gg) {  // fft.rs:341-345, utils/error.rs:104-116
            const double v = (double)(re / Lf);
            double o = div1e5(round(v * 100000.0));
            if (o > mxd) o = mxd;
            if (o < mnd) o = mnd;
            s += fabs(o - gg) * recip_abs(gg);
        };
        auto gload = [&](uint32_t m, double2 (&g)[9]) {
#pragma unroll
            for (int kq = 0; kq < 9; ++kq) {
                const uint32_t ja = (q + 9u * m) + 27u * kq;
                const int32_t j2 = (int32_t)(2 * (MD * ja + jb)) - (int32_t)pre;
                const int32_t jc = j2 < 0 ? 0 : (j2 >= (int32_t)n ? (int32_t)n - 2 : j2);
                double2 v = *(const double2 *)(xs + jc);
                if (j2 < 0) v.y = v.x;
                if (j2 >= (int32_t)n) v.x = v.y;
                g[kq] = v;
            }
        };
        double2 gn[9];
        gload(0, gn);
#pragma unroll 1
        for (uint32_t m = 0; m < 3; ++m) {
            double2 g[9];
#pragma unroll
            for (int kq = 0; kq < 9; ++kq) g[kq] = gn[kq];
            if (m + 1 < 3) gload(m + 1, gn);
            const uint32_t ka = q + 9u * m;
            float2 b[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) b[j] = T[(j * 27 + ka) * CSI + c];
            dft9f(b);
#pragma unroll
            for (int kq = 0; kq < 9; ++kq) {  // idft_L = 2 idft_M: even sample -> re, odd sample -> -im
                term(2.0f * b[kq].x, g[kq].x);
                term(-2.0f * b[kq].y, g[kq].y);
            }
        }
    }
    // block sum in a fixed order: wavefront DPP trees, then the three wavefronts in turn
    double *red = (double *)T;
    __syncthreads();
    const double wsum_ = wave_sum_f64(s);
    if ((tid & 63) == 0) red[tid >> 6] = wsum_;
    __syncthreads();
    if (tid == 0) ((double *)(ws + lay.o_c + FAST_PARTIAL_OFF))[blockIdx.x] = (red[0] + red[1]) + red[2];
    TSTAMP(4);
    if (dbg == 2 && tid == 0) {  // span of the launch's working tiles, printed by k_large_decide2
        atomicMin(&g_trip_span[0], tstamp[0]);
        atomicMax(&g_trip_span[1], tstamp[4]);
        atomicAdd(&g_trip_span[2], tstamp[4] - tstamp[0]);
        const unsigned long long ix = atomicAdd(&g_trip_span[3], 1ull);
        if (ix < 1024) { g_trip_log[ix][0] = tstamp[0]; g_trip_log[ix][1] = tstamp[4]; g_trip_log[ix][2] = blockIdx.x | (blockIdx.y << 8) | ((unsigned long long)__smid() << 32); }
    }
    if (dbg == 1 && tid == 0 && blockIdx.x == 3 && blockIdx.y < 5)
        printf("TSTAMP trip243 frame %u: setup %.1f  bucket sums %.1f  column transform %.1f  evaluation %.1f us (nlist %u)\n", blockIdx.y,
               (double)(tstamp[1] - tstamp[0]) * 0.01, (double)(tstamp[2] - tstamp[1]) * 0.01,
               (double)(tstamp[3] - tstamp[2]) * 0.01, (double)(tstamp[4] - tstamp[3]) * 0.01, nlist);
}

// The decision after the first FFT trip: frame/mod.rs:113-147 with every ladder either ended or pruned; anything
// else leaves the frame to k_compress_large<0>.
__global__ __launch_bounds__(LT) void k_large_decide2(
    const double *__restrict__ samples, const DevFrame *__restrict__ frames, const uint32_t *__restrict__ ids,
    const DevPlan *__restrict__ plans, const KParams prm, uint8_t *__restrict__ slots, DevResult *__restrict__ res,
    unsigned char *__restrict__ ws_base, uint64_t ws_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    if (prm.debug_stop == -6 && blockIdx.x == 0 && tid == 0) {
        printf("TSPAN trip243: %llu tiles, first start to last end %.1f us, mean tile %.1f us, decide2 starts %.1f us after the last end\n",
               g_trip_span[3], (double)(g_trip_span[1] - g_trip_span[0]) * 0.01,
               (double)g_trip_span[2] * 0.01 / (double)max(1ull, g_trip_span[3]), (double)(wall_clock64() - g_trip_span[1]) * 0.01);
        for (unsigned long long q = 0; q < min(1024ull, g_trip_span[3]); ++q)
            printf("TLOG %llu %llu %.1f %.1f %llx\n", g_trip_log[q][2] & 255, (g_trip_log[q][2] >> 8) & 0xffffff, (double)(g_trip_log[q][0] - g_trip_span[0]) * 0.01,
                   (double)(g_trip_log[q][1] - g_trip_span[0]) * 0.01, g_trip_log[q][2] >> 32);
        g_trip_span[0] = ~0ull; g_trip_span[1] = 0; g_trip_span[2] = 0; g_trip_span[3] = 0;
    }
    const uint32_t fid = ids[blockIdx.x];
    const DevFrame fr = frames[fid];
    const DevPlan &P = plans[fr.plan];
    const uint32_t n = P.n, L = P.L;
    unsigned char *ws = ws_base + (uint64_t)blockIdx.x * ws_stride;
    const LargeWs lay = large_ws_layout(n, L, P.kcap);
    FastState *fs = (FastState *)(ws + lay.o_front);
    if (tid == 0) {  // the tiles' sums, in tile order (asked for together with the state: one round trip)
        double s = 0.0;
        const double *part = (const double *)(ws + lay.o_c + FAST_PARTIAL_OFF);
        const uint32_t tiles = min(fast_geo(P).tiles, 32u);  // (18 at most for the frames this path serves)
        for (uint32_t t = 0; t < tiles; ++t) s += part[t];
        *(double *)(smem + 256) = s;
    }
    if (fs->status != 1) {  // decided by k_large_decide1 (2), or left by it to the general kernel (0): listed here
        if (tid == 0 && fs->status == 0) fb_append(ws_base, ws_stride, blockIdx.x);
        return;
    }
    const FastState f = *fs;
    uint32_t *wsum = (uint32_t *)smem;
    uint32_t *aux = (uint32_t *)(smem + 512);                       // 2048 u32
    const double *xs = samples + fr.sample_off;
    uint8_t *out = slots + fr.slot_off;
    const double me = prm.max_err;
    __syncthreads();
    const double cur = *(const double *)(smem + 256) / (double)L;  // mean over the padded samples (fft.rs:345)
    uint32_t best_size = f.best_size;
    int best_owner = f.best_owner;
    auto can_win = [&](uint32_t size_lb, int owner) {
        return size_lb < best_size || (size_lb == best_size && owner < best_owner);
    };
    const uint32_t K1 = f.K1;
    const uint32_t fft_size = 1 + vlen(K1) + 9 * K1 + 2 * f.big + 8;
    const bool fft_ends = !(prm.max_err_m < sat_i32(cur * 1000.0));  // fft.rs:334
    if (f.forced) {
        // forced FFT: the first trip ends the ladder (the payload is emitted below, whatever the error) or the general
        // kernel goes on with it
        if (!fft_ends) { if (tid == 0) { fs->status = 0; fb_append(ws_base, ws_stride, blockIdx.x); } FAST_WHY(11); return; }
        best_size = fft_size; best_owner = 0;
    } else if (fft_ends) {
        if (cur <= me && can_win(fft_size, 0)) { best_size = fft_size; best_owner = 0; }
    } else {
        // the ladder goes on with K2 bins: pruned only if that payload cannot beat a candidate that passes
        const uint32_t K2 = min(P.mf + P.dk1, f.Z);
        if (K2 <= K1 || can_win(1 + vlen(K2) + 9 * K2 + 8, 0)) { if (tid == 0) { fs->status = 0; fb_append(ws_base, ws_stride, blockIdx.x); } FAST_WHY(11); return; }
    }
    if (!f.forced && !f.poly_final && can_win(f.poly2_lb, 1)) { if (tid == 0) { fs->status = 0; fb_append(ws_base, ws_stride, blockIdx.x); } FAST_WHY(12); return; }
    if (!f.forced && (can_win(f.rle_lb, 2) || best_owner == 3)) { if (tid == 0) { fs->status = 0; fb_append(ws_base, ws_stride, blockIdx.x); } FAST_WHY(best_owner == 3 ? 14 : 13); return; }
    if (best_owner == 1) {
        fast_emit_poly(out, res[fid], xs, n, f.bitdepth, f.poly_K, f.poly_step, f.smin, f.smax, f.poly_err, aux, wsum);
    } else {
        // FFT payload (fft.rs:119-130): the bins by descending norm, equal norms by position (the large tier's order)
        // (k_large_decide1 left sel[] in the payload's order)
        const Sel *sel = (const Sel *)(ws + lay.o_sel);
        const uint32_t hdr = 1 + vlen(K1);
        for (uint32_t i = tid; i < K1; i += LT) aux[i] = vlen(sel[i].pos & 0xffffu) + 8;
        __syncthreads();
        const uint32_t body = block_excl_scan<LW>(aux, K1, wsum);
        for (uint32_t i = tid; i < K1; i += LT) {
            const Sel e = sel[i];
            uint8_t *p = out + hdr + aux[i];
            p += put_varint(p, e.pos & 0xffffu);
            put_f32(p, e.re);
            put_f32(p + 4, e.im);
        }
        if (tid == 0) {
            out[0] = 15;
            put_varint(out + 1, K1);
            put_f32(out + hdr + body, f.mxf);
            put_f32(out + hdr + body + 4, f.mnf);
            res[fid].err = cur;
            res[fid].len = hdr + body + 8;
            res[fid].chosen = ATSC_FFT;
        }
    }
    if (tid == 0) fs->status = 2;
}

// Decoder: an FFT frame of 131072 samples (fft.rs:426-462) up to the point where k_large_trip243<true> takes over: the
// payload's entries (rds_fft_entries: positions mirrored into [0, L/2], purely real bins' imaginary parts cleared), "later
// entries overwrite earlier ones" (fft.rs:401-422: an entry is void when a later one names its position), the bucketed
// list.  Anything else -- another codec, more entries than the encoder's first trip stores, a malformed payload (the
// general decoder reports it) -- is left to k_decompress_large<0> (FastState::status stays 0).
constexpr uint32_t FAST_DP_LDS = 512 + STG_BYTES + 4096 + 8 * FAST_K_MAX + 2 * 12 * (2 * FAST_K_MAX);
__global__ __launch_bounds__(LT) void k_large_dparse(
    const DevDFrame *__restrict__ frames, const uint32_t *__restrict__ ids, const DevPlan *__restrict__ plans,
    const float2 *__restrict__ twpool, const uint8_t *__restrict__ body, unsigned char *__restrict__ ws_base,
    uint64_t ws_stride, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    unsigned long long dst_[6] = {0, 0, 0, 0, 0, 0};
#define DSTAMP(i) do { if (dbg) dst_[i] = wall_clock64(); } while (0)
    DSTAMP(0);
    const DevDFrame fr = frames[ids[blockIdx.x]];
    const DevPlan &P = plans[fr.plan];
    const uint32_t L = P.L, M = P.M;
    unsigned char *ws = ws_base + (uint64_t)blockIdx.x * ws_stride;
    const LargeWs lay = large_ws_layout(fr.n, L, P.kcap);
    FastState *fs = (FastState *)(ws + lay.o_front);
    if (tid == 0) fs->status = 0;
    if (blockIdx.x == 0 && tid == 0) *fb_count(ws_base, ws_stride) = 0;  // (filled by the first k_large_trip243<true> launch)
    const FastGeo geo = fast_geo(P);
    if ((fr.tag != ATSC_FFT && fr.tag != ATSC_POLYNOMIAL && fr.tag != ATSC_RLE && fr.tag != ATSC_CONSTANT) || !geo.ok || (P.pre & 1u) ||
        (fr.n & 1u)) return;
    // LDS: [hdr 512][payload window STG_BYTES][tab 1024 u32][pos FAST_K_MAX u32 + dead FAST_K_MAX u32][zl][zs]
    uint32_t *bc = (uint32_t *)smem;
    float *bcf = (float *)(smem + 64);
    uint32_t *tab = (uint32_t *)(smem + 512 + STG_BYTES);
    uint32_t *lpos = tab + 1024;
    uint32_t *ldead = lpos + FAST_K_MAX;
    SpEnt *zl = (SpEnt *)(ldead + FAST_K_MAX);
    SpEnt *zs = zl + 2 * FAST_K_MAX;
    Sel *ent = (Sel *)(ws + lay.o_sel);
    const float2 *tw = twpool + P.tw_off;
    // The whole payload into LDS first, by every thread (aligned dwords; the bytes in front of the payload belong to the
    // record's header, the last partial dword is read byte by byte): the lock-step parser below then never refills its
    // window -- a refill is 64 lanes copying 16 KB four bytes at a time, ~100 us for a payload of this size.
    if (fr.payload_len + 4 > STG_BYTES) return;
    const uint8_t *pay = body + fr.payload_off;
    const uint32_t mis = (uint32_t)((uintptr_t)pay & 3u);
    {
        const uint32_t *src = (const uint32_t *)(pay - mis);
        const uint32_t nd = (mis + fr.payload_len) >> 2;  // whole dwords
        uint32_t *dst = (uint32_t *)(smem + 512);
        for (uint32_t w = tid; w < nd; w += LT) dst[w] = src[w];
        for (uint32_t b = 4 * nd + tid; b < mis + fr.payload_len; b += LT) (smem + 512)[b] = (pay - mis)[b];
    }
    __syncthreads();
    DSTAMP(1);
    // Header: id byte, varint count.  Then the entries -- 9 bytes (position < 251) or 11 (marker 251 + u16) each, so an
    // entry's start depends on every entry before it.  One wavefront walking them in lock step (rds_fft_entries)
    // took 88 us for 1310 entries; here next(o) = o + 9 or 11 is tabulated for EVERY byte offset o, squared six times
    // (next^64), a single thread hops from group to group of 64 entries, and the sixteen wavefronts parse the
    // groups side by side.
    uint8_t *win = smem + 512 + mis;
    if (fr.tag == ATSC_CONSTANT) {
        // constant.rs:141-144: one value for the whole frame; the tile grid writes it (status 5), a piece per workgroup,
        // instead of one workgroup of the general decoder writing a megabyte alone
        if (tid < 64) {
            RdS r{pay, fr.payload_len, 0, false, win, 0, fr.payload_len, STG_BYTES};
            (void)rds_u8(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            double v = 0.0;
            bool bad = bd > 3;
            if (!bad) v = rds_value(r, bd);
            bad = bad || r.bad;
            if (tid == 0 && !bad) { fs->smin = v; fs->status = 5; }
        }
        return;
    }
    if (fr.tag == ATSC_RLE) {
        // RLE frame (rle.rs:204-236): groups of (value, count, count run starts).  One wavefront walks the group headers
        // (below), every thread decodes starts, and the keys (start << 32 | group) are sorted in LDS (runs that share a
        // start: sorted by (start, group), all but the last of them are empty -- the reference's outcome); then (start,
        // value) per run in sample order and, per piece of the tile grid, how many runs start at or before the piece's
        // first sample go to the workspace.  k_large_trip243<true> expands the runs, one piece per workgroup.
        unsigned long long *lk = (unsigned long long *)zl;  // zl + zs: 64512 bytes
        double *gvals = (double *)(ws + lay.o_tab);
        // Group table (tab .. ldead: 14336 bytes): per group {offset of its first start, 1-byte starts, 3-byte starts,
        // runs before the group}.  The encoder pushes a group's starts in scan order (rle.rs:158-169), so their varint
        // widths only grow -- 1 byte below 251, then 3 (marker 251), then 5 (marker 252) -- and the walker finds a
        // group's end from two marker scans, 64 bytes / entries a step, instead of decoding one varint after the other
        // (a lone wavefront pays ~700 cycles of LDS latency per dependent field: 327 us for a gauge's 964 runs).  The
        // starts are then decoded by every thread at once, and each checks its marker: a stream that is not built that
        // way is left to the general decoder.
        constexpr uint32_t GMAX = 896;
        uint4 *gtab = (uint4 *)tab;
        if (tid < 64) {
            const uint32_t lane = tid;
            RdS r{pay, fr.payload_len, 0, false, win, 0, fr.payload_len, STG_BYTES};
            (void)rds_u8(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t groups = rds_varint(r);
            bool bad = r.bad || bd > 3 || groups == 0 || groups > GMAX;
            uint32_t e = 0;
            for (uint32_t gi = 0; gi < (uint32_t)groups && !bad; ++gi) {
                // The group's first 64 bytes, one per lane, in ONE read: value, count and -- for a group of up to ~15
                // starts, i.e. most groups of a gauge -- both marker scans come out of that register by readlane / ballot.
                // (Field by field through rds_*: five dependent LDS round trips a group, 0.6 us; 41 groups: 24.5 us.)
                const uint32_t g0 = r.pos;
                uint32_t c = 0, a = 0, b = 0, o = 0;
                double v = 0.0;
                bool fastok = false;
                {
                    const uint32_t B = (g0 + lane < r.len) ? (uint32_t)win[g0 + lane] : 0xFFu;  // (255: no marker)
                    auto rb = [&](uint32_t k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)B, (int)k); };
                    auto le = [&](uint32_t k, uint32_t nb) -> uint64_t {
                        uint64_t x = 0;
                        for (uint32_t q = 0; q < nb; ++q) x |= (uint64_t)rb(k + q) << (8 * q);
                        return x;
                    };
                    auto vint = [&](uint32_t k, uint64_t &val) -> uint32_t {  // width; 0: not a varint
                        const uint32_t t = rb(k);
                        if (t < 251) { val = t; return 1; }
                        if (t == 251) { val = le(k + 1, 2); return 3; }
                        if (t == 252) { val = le(k + 1, 4); return 5; }
                        if (t == 253) { val = le(k + 1, 8); return 9; }
                        return 0;
                    };
                    uint64_t raw = 0, c64 = 0;
                    uint32_t w1;
                    if (bd == 3) { raw = rb(0); w1 = 1; }
                    else if (bd == 0) { raw = le(0, 8); w1 = 8; }
                    else w1 = vint(0, raw);
                    const uint32_t w2 = w1 ? vint(w1, c64) : 0u;
                    const uint32_t hh = w1 + w2;  // 2 .. 18
                    if (w1 && w2 && g0 + hh <= r.len && c64 >= 1 && c64 <= FAST_RLE_DEC_MAX - e) {
                        c = (uint32_t)c64;
                        v = bd == 3 ? (double)raw : bd == 2 ? (double)(int16_t)unzig(raw) : bd == 1 ? (double)(int32_t)unzig(raw)
                                                                                              : __longlong_as_double((long long)raw);
                        o = g0 + hh;
                        // (lanes past the window count as "does not conform": the shifts fill with zeros)
                        const unsigned long long m1 = __ballot(B < 251) >> hh;
                        a = min((uint32_t)__builtin_ctzll(~m1), c);
                        if (a == c) {
                            fastok = true;
                        } else if (hh + a < 64) {
                            const uint32_t p3 = hh + a;
                            const unsigned long long m3 = __ballot(B == 251) >> p3;
                            const uint32_t k3 = (uint32_t)__builtin_ctzll(~m3 & 0x9249249249249249ull) / 3u;
                            if (k3 >= c - a) { b = c - a; fastok = true; }
                            else if (p3 + 3 * k3 < 64) { b = k3; fastok = true; }  // (the marker that ends the scan is in the window)
                        }
                    }
                }
                if (!fastok) {
                    v = rds_value(r, bd);
                    const uint64_t c64 = rds_varint(r);
                    if (r.bad || c64 == 0 || c64 > FAST_RLE_DEC_MAX - e) { bad = true; break; }
                    c = (uint32_t)c64; o = r.pos;
                    // leading 1-byte starts, then 3-byte ones; what is left must be 5-byte starts
                    a = 0; b = 0;
                    for (;;) {
                        const uint32_t k = a + lane;
                        const bool one = k < c && o + k < r.len && win[o + k] < 251;
                        const unsigned long long m = __ballot(!one);
                        if (m) { a += (uint32_t)__builtin_ctzll(m); break; }
                        a += 64;
                    }
                    a = min(a, c);
                    for (;;) {
                        const uint32_t k = b + lane;
                        const bool three = a + k < c && o + a + 3 * k + 2 < r.len && win[o + a + 3 * k] == 251;
                        const unsigned long long m = __ballot(!three);
                        if (m) { b += (uint32_t)__builtin_ctzll(m); break; }
                        b += 64;
                    }
                    b = min(b, c - a);
                }
                if (lane == 0) gvals[gi] = v;
                const uint32_t bytes = a + 3 * b + 5 * (c - a - b);
                if (o + bytes > r.len) { bad = true; break; }
                if (lane == 0) gtab[gi] = make_uint4(o, a, b, e);
                r.pos = o + bytes;
                e += c;
            }
            if (r.pos != r.len) bad = true;  // (anything unusual is the general decoder's to judge)
            if (tid == 0) { bc[0] = bad ? 0u : 1u; bc[1] = e; bc[2] = (uint32_t)groups; bc[3] = 0; }
        }
        __syncthreads();
        DSTAMP(2);
        if (!bc[0] || bc[1] == 0) return;
        {
            const uint32_t E0 = bc[1], D = bc[2];
            for (uint32_t j = tid; j < E0; j += LT) {
                uint32_t lo = 0, hi = D;  // the last group that starts at or before run j
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (gtab[mid].w <= j) lo = mid;
                    else hi = mid;
                }
                const uint4 g = gtab[lo];
                const uint32_t k = j - g.w;
                uint32_t idx, okk;
                if (k < g.y) {
                    idx = win[g.x + k];
                    okk = idx < 251 ? 1u : 0u;
                } else if (k < g.y + g.z) {
                    const uint32_t q = g.x + g.y + 3 * (k - g.y);
                    idx = (uint32_t)win[q + 1] | ((uint32_t)win[q + 2] << 8);
                    okk = win[q] == 251 ? 1u : 0u;
                } else {
                    const uint32_t q = g.x + g.y + 3 * g.z + 5 * (k - g.y - g.z);
                    idx = (uint32_t)win[q + 1] | ((uint32_t)win[q + 2] << 8) | ((uint32_t)win[q + 3] << 16) | ((uint32_t)win[q + 4] << 24);
                    okk = win[q] == 252 ? 1u : 0u;
                }
                if (!okk || idx >= fr.n) bc[3] = 1;
                lk[j] = ((unsigned long long)idx << 32) | lo;
            }
        }
        __syncthreads();
        DSTAMP(3);
        if (bc[3]) return;
        const uint32_t E = bc[1];
        uint32_t *rstart = (uint32_t *)(ws + lay.o_hp);
        double *rval = (double *)(ws + lay.o_rec);
        uint32_t *ptab = (uint32_t *)(ws + lay.o_part);
        // A run's place in sample order = the number of runs that start before it.  Starts are distinct unless the stream
        // holds empty runs, so: one bit per sample where a run starts (16 KB, where the payload and the group table were --
        // both are done with), the bits counted per 64-sample word and scanned, and every run reads its place off the
        // counts -- 3 us where sorting the 950 keys of a gauge through 55 barriers took 12.  Two runs on one start: the
        // sort below.
        uint32_t *bm = (uint32_t *)(smem + 512);  // n / 32 words (n <= 131072); [smem + 512, zl) is 31232 bytes
        uint32_t *pre = bm + 4096;                // n / 64 counts, then their exclusive scan
        const uint32_t nw64 = (fr.n + 63) >> 6;
        for (uint32_t w = tid; w < 2 * nw64; w += LT) bm[w] = 0;
        if (tid == 0) bc[4] = 0;
        __syncthreads();
        for (uint32_t j = tid; j < E; j += LT) {
            const uint32_t idx = (uint32_t)(lk[j] >> 32), bit = 1u << (idx & 31u);
            if (atomicOr(&bm[idx >> 5], bit) & bit) bc[4] = 1;
        }
        __syncthreads();
        if (!bc[4]) {
            for (uint32_t w = tid; w < nw64; w += LT) pre[w] = __popc(bm[2 * w]) + __popc(bm[2 * w + 1]);
            __syncthreads();
            (void)block_excl_scan<LW>(pre, nw64, bc + 64);
            auto upto = [&](uint32_t i) -> uint32_t {  // starts at samples < i
                const uint32_t w = i >> 6, bi = i & 63u;
                const unsigned long long word = (unsigned long long)bm[2 * w] | ((unsigned long long)bm[2 * w + 1] << 32);
                return pre[w] + (uint32_t)__popcll(word & ((1ull << bi) - 1ull));
            };
            for (uint32_t j = tid; j < E; j += LT) {
                const unsigned long long k = lk[j];
                const uint32_t idx = (uint32_t)(k >> 32), at = upto(idx);
                rstart[at] = idx;
                rval[at] = gvals[(uint32_t)(k & 0xffffffffull)];
            }
            if (tid <= geo.tiles) {
                uint32_t cnt = E;
                if (tid < geo.tiles) {
                    const uint32_t i0 = (uint32_t)(((uint64_t)fr.n * tid) / geo.tiles);
                    cnt = upto(i0) + ((bm[i0 >> 5] >> (i0 & 31u)) & 1u);  // runs with start <= i0
                }
                ptab[tid] = cnt;
            }
        } else {
            uint32_t p2 = 1;
            while (p2 < E) p2 <<= 1;
            block_sort<LW, true>((uint64_t *)lk, nullptr, E, p2);
            for (uint32_t i = tid; i < E; i += LT) {
                const unsigned long long k = lk[i];
                rstart[i] = (uint32_t)(k >> 32);
                rval[i] = gvals[(uint32_t)(k & 0xffffffffull)];
            }
            if (tid <= geo.tiles) {
                uint32_t cnt = E;
                if (tid < geo.tiles) {
                    const uint32_t i0 = (uint32_t)(((uint64_t)fr.n * tid) / geo.tiles);
                    uint32_t lo = 0, hi = E;  // runs with start <= i0
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint32_t)(lk[mid] >> 32) <= i0) lo = mid + 1;
                        else hi = mid;
                    }
                    cnt = lo;
                }
                ptab[tid] = cnt;
            }
        }
        DSTAMP(4);
        if (tid == 0) { fs->K1 = E; fs->status = 4; }
        DSTAMP(5);
        if (dbg && tid == 0)
            printf("DSTAMP dparse RLE: staging %.1f  group walk %.1f  starts %.1f  ranks %.1f  tables %.1f us (%u runs, %u groups)\n",
                   (double)(dst_[1] - dst_[0]) * 0.01, (double)(dst_[2] - dst_[1]) * 0.01, (double)(dst_[3] - dst_[2]) * 0.01,
                   (double)(dst_[4] - dst_[3]) * 0.01, (double)(dst_[5] - dst_[4]) * 0.01, E, bc[2]);
        return;
    }
    if (fr.tag == ATSC_POLYNOMIAL) {
        // Catmull-Rom frame (polynomial.rs:395-404): the points into the workspace as doubles, the rest is
        // k_large_trip243<true>'s (one piece of the frame per workgroup).  Fixed-width points (F64 / U8) are read by
        // every thread; varint points (I16 / I32) are found like the FFT entries below: next(o) = o + width tabulated
        // for every byte offset, squared six times, groups of 64 values parsed side by side.
        double *vals = (double *)(ws + lay.o_tab);
        double *bcd = (double *)(smem + 128);
        if (tid < 64) {
            RdS r{pay, fr.payload_len, 0, false, win, 0, fr.payload_len, STG_BYTES};
            const uint32_t id = (uint32_t)rds_varint(r);
            const uint32_t bd = (uint32_t)rds_varint(r);
            const uint64_t c64 = rds_varint(r);
            if (tid == 0) {
                bc[0] = (!r.bad && id == 0 && bd <= 3 && c64 >= 2 && c64 <= FAST_K_MAX && r.pos + 17 <= r.len) ? 1u : 0u;
                bc[1] = (uint32_t)c64; bc[2] = r.pos; bc[3] = 0; bc[5] = bd;
            }
        }
        __syncthreads();
        if (!bc[0]) return;
        const uint32_t K = bc[1], hdr = bc[2], bd = bc[5];
        const uint32_t E = fr.payload_len - hdr - 17;  // bytes of the points
        if (bd == 0 || bd == 3) {
            if (E != K * (bd == 0 ? 8u : 1u)) return;
            for (uint32_t k = tid; k < K; k += LT) {
                if (bd == 3) {
                    vals[k] = (double)win[hdr + k];
                } else {
                    uint64_t v = 0;
                    for (int b = 0; b < 8; ++b) v |= (uint64_t)win[hdr + 8 * k + b] << (8 * b);
                    vals[k] = __longlong_as_double((long long)v);
                }
            }
        } else {
            if (E < K || E > 9 * K) return;
            uint16_t *Ja = (uint16_t *)zl, *Jb = Ja + STG_BYTES;
            for (uint32_t o = tid; o <= E; o += LT) {
                uint32_t nx = E;
                if (o < E) {
                    const uint32_t m = win[hdr + o];
                    nx = o + (m < 251 ? 1u : m == 251 ? 3u : m == 252 ? 5u : 9u);
                }
                Ja[o] = (uint16_t)min(nx, E);
            }
            __syncthreads();
            for (int k = 0; k < 6; ++k) {
                for (uint32_t o = tid; o <= E; o += LT) Jb[o] = Ja[Ja[o]];
                __syncthreads();
                uint16_t *t = Ja; Ja = Jb; Jb = t;
            }
            uint32_t *cs = tab;
            const uint32_t ngrp = (K + 63) / 64;
            if (tid == 0) {
                uint32_t o = 0;
                for (uint32_t k = 0; k < ngrp; ++k) { cs[k] = o; o = Ja[o]; }
                cs[ngrp] = E;
            }
            __syncthreads();
            for (uint32_t k = tid >> 6; k < ngrp; k += LW) {
                RdS r{pay, hdr + E, hdr + cs[k], false, win, 0, hdr + E, STG_BYTES};
                const uint32_t grp = min(64u, K - 64 * k);
                rds_varints(r, grp, [&](uint32_t i, uint64_t v) {
                    vals[64 * k + i] = (bd == 2) ? (double)(int16_t)unzig(v) : (double)(int32_t)unzig(v);
                });
                if ((r.bad || r.pos != hdr + cs[k + 1]) && (tid & 63) == 0) atomicOr(&bc[3], 1u);
            }
            __syncthreads();
            if (bc[3]) return;
        }
        if (tid == 0) {
            uint64_t a = 0, b2 = 0;
            for (uint32_t q = 0; q < 8; ++q) {
                a |= (uint64_t)win[hdr + E + q] << (8 * q);
                b2 |= (uint64_t)win[hdr + E + 8 + q] << (8 * q);
            }
            bcd[0] = __longlong_as_double((long long)a);
            bcd[1] = __longlong_as_double((long long)b2);
            bc[6] = win[hdr + E + 16];
        }
        __syncthreads();
        const double mn = bcd[0], mx = bcd[1];
        const uint32_t step = bc[6];
        if (mx == mn || step < 16) return;  // a fill, or a step the piece kernel has no tables for: the general decoder
        {
            const uint32_t cntk = (fr.n + step - 1) / step;
            const uint32_t Kp = cntk + (((cntk - 1) * step != fr.n - 1) ? 1u : 0u);
            if (Kp != K) return;  // (the general decoder reports it)
        }
        if (tid == 0) {
            fs->smin = mn; fs->smax = mx; fs->poly_K = K; fs->poly_step = step;
            fs->status = 3;
        }
        return;
    }
    if (tid < 64) {
        RdS r{pay, fr.payload_len, 0, false, win, 0, fr.payload_len, STG_BYTES};
        (void)rds_u8(r);
        const uint64_t cnt64 = rds_varint(r);
        if (tid == 0) {
            bc[0] = (!r.bad && cnt64 >= 1 && cnt64 <= FAST_K_MAX && cnt64 <= P.kcap && r.pos + 8 <= r.len) ? 1u : 0u;
            bc[1] = (uint32_t)cnt64;
            bc[2] = r.pos;
            bc[3] = 0;  // raised by a group that does not parse
        }
    }
    __syncthreads();
    if (!bc[0]) return;
    const uint32_t cnt = bc[1], hdr = bc[2];
    const uint32_t E = fr.payload_len - hdr - 8;  // bytes of the entries
    if (E < 9 * cnt || E > 11 * cnt || ((11 * cnt - E) & 1u)) return;  // (left to the general decoder, which reports it)
    {
        // Where the entries start.  An entry is 9 or 11 bytes, so the chain of starts enters every block of BB bytes
        // within the block's first 11: thread (block, c) walks the block from its byte c -- 40 dependent byte reads at
        // most -- and leaves the starts it met, their number and the byte of the next block it came out at; one thread
        // then follows the real chain from block to block, and every entry is parsed by a thread of its own.  (Before:
        // next(o) tabulated for every byte offset and squared six times, a hop per 64 entries, the groups of 64 parsed
        // by a wavefront each in lock step -- 20 us for 1310 entries.)
        constexpr uint32_t BB = 352, BW = 40, NBMAX = (11 * FAST_K_MAX + BB - 1) / BB;  // 42 blocks of <= 40 entries
        static_assert(NBMAX * 11 * BW * 2 + NBMAX * 11 * 2 <= 2 * 12 * 2 * FAST_K_MAX, "start tables fit the list buffers");
        static_assert((NBMAX + 1) * 8 <= 4096, "chain tables fit tab");
        uint16_t *offs = (uint16_t *)zl;               // [block][c][BW]
        uint16_t *bxc = offs + NBMAX * 11 * BW;        // [block][c]: count | exit byte << 8 (15: the chain ends inside)
        uint32_t *centry = tab, *ebase = tab + NBMAX + 1;
        const uint32_t nblk = (E + BB - 1) / BB;
        for (uint32_t t = tid; t < nblk * 11; t += LT) {
            const uint32_t b = t / 11, c = t % 11, bs = b * BB, be = min(bs + BB, E);
            uint32_t o = bs + c, j = 0;
            uint16_t *row = offs + t * BW;
            while (o < be) {  // (j < BW: 352 / 9 < 40)
                row[j++] = (uint16_t)o;
                o += win[hdr + o] < 251 ? 9u : 11u;
            }
            // out of the last block: exactly at the end of the entries (0), or past it (14: an entry cut short)
            const uint32_t ex = be == E ? (o == E ? 0u : 14u) : o - be;
            bxc[t] = (uint16_t)(j | (ex << 8));
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t c = 0, base = 0;
            for (uint32_t b = 0; b < nblk; ++b) {
                centry[b] = c;
                ebase[b] = base;
                const uint32_t w = bxc[b * 11 + c];
                base += w & 0xffu;
                c = w >> 8;
            }
            ebase[nblk] = base;
            if (c != 0 || base != cnt) bc[3] = 1;  // (left to the general decoder, which reports it)
        }
        __syncthreads();
        if (bc[3]) return;
        for (uint32_t t = tid; t < nblk * BW; t += LT) {
            const uint32_t b = t / BW, j = t % BW, row = b * 11 + centry[b];
            if (j >= (bxc[row] & 0xffu)) continue;
            const uint32_t o = hdr + offs[row * BW + j], first = win[o];
            uint32_t pos = first, q = o + 1;
            if (first == 251) {
                pos = (uint32_t)win[o + 1] | ((uint32_t)win[o + 2] << 8);
                q = o + 3;
            }
            uint32_t wre = 0, wim = 0;
            for (uint32_t k = 0; k < 4; ++k) {
                wre |= (uint32_t)win[q + k] << (8 * k);
                wim |= (uint32_t)win[q + 4 + k] << (8 * k);
            }
            float re = __uint_as_float(wre), im = __uint_as_float(wim);
            if (first > 251 || pos >= L) {  // (a u16 field: 4- and 8-byte varints cannot occur; rds_fft_entries)
                bc[3] = 1;
                continue;
            }
            if (pos > L / 2) { pos = L - pos; im = -im; }
            if (pos == 0 || 2 * pos == L) im = 0.0f;
            Sel e;
            e.pos = pos; e.re = re; e.im = im;
            ent[ebase[b] + j] = e;
        }
        __syncthreads();
        if (bc[3]) return;
        if (tid == 0) {
            uint32_t a = 0, b2 = 0;
            for (uint32_t q = 0; q < 4; ++q) {
                a |= (uint32_t)win[hdr + E + q] << (8 * q);
                b2 |= (uint32_t)win[hdr + E + 4 + q] << (8 * q);
            }
            bcf[0] = __uint_as_float(a);
            bcf[1] = __uint_as_float(b2);
        }
        __syncthreads();
    }
    const float mxf = bcf[0], mnf = bcf[1];
    if (mxf == mnf) return;  // a fill: left to the general decoder
    DSTAMP(2);
    // later entries overwrite earlier ones: entry i is void when a later one names its position.  A conforming stream
    // names a position twice only through the `pos as u16` wrap, so a bitmap of the positions (8 KB of the idle list
    // buffers) first tells whether any position repeats at all; only then every entry looks at the entries behind it.
    uint32_t *bitmap = (uint32_t *)zs;  // 65536 bits
    for (uint32_t i = tid; i < 2048; i += LT) bitmap[i] = 0;
    for (uint32_t i = tid; i < FAST_K_MAX; i += LT) {
        lpos[i] = i < cnt ? ent[i].pos : 0xFFFFFFFFu - i;
        ldead[i] = 0;
    }
    // (a frame with such repeats used to have every entry walk all the entries behind it: 25 us for 1310 of them, one
    // frame in four at e = 1 %.  The repeats are a handful: the positions found set go to a list, the entries that name a
    // listed position agree on the last of them by an LDS maximum, and the others are void.)
    constexpr uint32_t DUPMAX = 64;
    uint32_t *dupp = tab, *dupl = tab + DUPMAX;  // listed positions, last entry naming each (tab is idle until the bucketing)
    if (tid == 0) bc[4] = 0;
    if (tid < DUPMAX) dupl[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += LT) {
        const uint32_t p = lpos[i] & 0xffffu, bit = 1u << (p & 31u);
        if (atomicOr(&bitmap[p >> 5], bit) & bit) {  // (positions are < 65536: stored as u16, and 65536 < L / 2 = 69984)
            const uint32_t q = atomicAdd(&bc[4], 1u);
            if (q < DUPMAX) dupp[q] = lpos[i];
        }
    }
    __syncthreads();
    const uint32_t ndup = bc[4];
    if (ndup && ndup <= DUPMAX) {
        uint32_t slot[(FAST_K_MAX + LT - 1) / LT];
        for (uint32_t i = tid, u = 0; i < cnt; i += LT, ++u) {
            const uint32_t mine = lpos[i];
            uint32_t q = 0;
            while (q < ndup && dupp[q] != mine) ++q;
            slot[u] = q;
            if (q < ndup) atomicMax(&dupl[q], i);
        }
        __syncthreads();
        for (uint32_t i = tid, u = 0; i < cnt; i += LT, ++u)
            if (slot[u] < ndup) ldead[i] = i < dupl[slot[u]] ? 1u : 0u;
    } else if (ndup) {
        for (uint32_t i = tid; i < cnt; i += LT) {
            const uint32_t mine = lpos[i];
            uint32_t dead = 0;
            for (uint32_t j = (i + 1) & ~3u; j < ((cnt + 3u) & ~3u); j += 4) {  // (padding slots hold positions nobody has)
                const uint4 p4 = *(const uint4 *)(lpos + j);
                dead |= (j > i && p4.x == mine) | (j + 1 > i && p4.y == mine) | (j + 2 > i && p4.z == mine) |
                        (j + 3 > i && p4.w == mine);
            }
            ldead[i] = dead;
        }
    }
    __syncthreads();
    DSTAMP(3);
    const uint32_t nlist = fast_bucket(
        cnt,
        [&](uint32_t i, uint32_t &p, float2 &x) -> bool {
            if (ldead[i]) return false;
            const Sel e = ent[i];
            p = e.pos;
            if (p > M) return false;  // (cannot be: the entries come mirrored into [0, L / 2])
            x = make_float2(e.re, e.im);
            return true;
        },
        tw, M, zl, zs, tab, bc + 8, ws + lay.o_b, fast_bounds_off(cnt));
    if (tid == 0) {
        fs->nlist = nlist;
        fs->mxf = mxf;
        fs->mnf = mnf;
        fs->K1 = cnt;
        fs->status = 1;
    }
    DSTAMP(4);
    if (dbg && tid == 0 && blockIdx.x == 0)
        printf("DSTAMP dparse: staging %.1f  parse %.1f  void entries %.1f  bucketing %.1f us (%u entries)\n",
               (double)(dst_[1] - dst_[0]) * 0.01, (double)(dst_[2] - dst_[1]) * 0.01, (double)(dst_[3] - dst_[2]) * 0.01,
               (double)(dst_[4] - dst_[3]) * 0.01, cnt);
}

constexpr uint32_t FAST_D1_LDS = 512 + 16384 + 8 * (FAST_OWN + 2 * FAST_K_MAX + FAST_CAND_MAX);
constexpr uint32_t FAST_TILE_LDS = 8 * (243 * CSI + 243 + FAST_MD) + 4 * 768;
constexpr uint32_t FAST_D2_LDS = 512 + 8192;
