// atsc_large_cols.h -- included by atsc_large.hip (inside namespace atsc, after PreFrame / large_ws_layout).
//
// Column pass of the large tier's two-pass forward transform for plans whose split has M1 = 243 = 27 * 9 (every
// frame length the reference chunker emits above 4096 samples has M = 2^a 3^7; 131072 samples: M = 243 x 288).
// Same inputs, outputs and layout as k_large_pre1<DevFrame, false>, other geometry:
//   * a workgroup of 192 threads (144 of them busy) per tile of 16 columns instead of 1024 threads with four
//     points each and five workgroup barriers: thread (column c, q) loads the 27 points n1 = 9 a + q of its
//     column -- all 27 loads in flight at once -- and transforms them in registers (radix 27 = 9 x 3), the
//     results cross the tile once through LDS, and the same thread then runs three radix-9 transforms over q;
//   * 35 KB of LDS per workgroup, i.e. four workgroups per CU in different phases (one loading while another
//     transforms), where the 64 KB tile pair of the Stockham form left two in lock step.
// fft.rs:315-323 (rustfft forward transform; complex f32 arithmetic is not bit-reproducible across transform
// algorithms: parity is the tolerance bar of tests/parity.py).

DEVI void dft3f(float2 &a0, float2 &a1, float2 &a2)  // forward, in place: outputs k = 0, 1, 2
{
    const float2 t1 = make_float2(a1.x + a2.x, a1.y + a2.y);
    const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
    const float2 d = make_float2(a1.x - a2.x, a1.y - a2.y);
    const float h = 0.8660254037844386f;
    const float2 t3 = make_float2(h * d.y, -h * d.x);
    a0 = make_float2(a0.x + t1.x, a0.y + t1.y);
    a1 = make_float2(t2.x + t3.x, t2.y + t3.y);
    a2 = make_float2(t2.x - t3.x, t2.y - t3.y);
}
// x[j], j = 0..8 -> X[k] = sum_j x[j] W9^{jk}, natural order, in place
DEVI void dft9f(float2 (&a)[9])
{
    // j = 3 j1 + j2, k = k1 + 3 k2:  W9^{jk} = W3^{j1 k1} W9^{j2 k1} W3^{j2 k2}
    dft3f(a[0], a[3], a[6]);
    dft3f(a[1], a[4], a[7]);
    dft3f(a[2], a[5], a[8]);
    const float2 w1 = make_float2(0.766044443118978f, 0.642787609686539f);   // (cos, sin) 40 deg
    const float2 w2 = make_float2(0.17364817766693f, 0.984807753012208f);    // 80 deg
    const float2 w4 = make_float2(-0.939692620785908f, 0.342020143325669f);  // 160 deg
    a[4] = cmulc(a[4], w1);
    a[7] = cmulc(a[7], w2);
    a[5] = cmulc(a[5], w2);
    a[8] = cmulc(a[8], w4);
    dft3f(a[0], a[1], a[2]);  // k1 = 0: k = 0, 3, 6
    dft3f(a[3], a[4], a[5]);  // k1 = 1: k = 1, 4, 7
    dft3f(a[6], a[7], a[8]);  // k1 = 2: k = 2, 5, 8
    // a[3 k1 + k2] holds X[k1 + 3 k2]: transpose the 3 x 3
    float2 t;
    t = a[1]; a[1] = a[3]; a[3] = t;
    t = a[2]; a[2] = a[6]; a[6] = t;
    t = a[5]; a[5] = a[7]; a[7] = t;
}
// x[j], j = 0..26 -> X[k] = sum_j x[j] W27^{jk}, natural order, in place
DEVI void dft27f(float2 (&a)[27])
{
    // j = 3 j1 + j2 (j1 < 9), k = k1 + 9 k2 (k1 < 9):  W27^{jk} = W9^{j1 k1} W27^{j2 k1} W3^{j2 k2}
    constexpr float W27[17][2] = {
        {1.0f, 0.0f},
        {0.97304487057982381f, 0.23061587074244017f},
        {0.89363264032341228f, 0.44879918020046217f},
        {0.76604444311897801f, 0.64278760968653925f},
        {0.59715859170278618f, 0.80212319275504373f},
        {0.3960797660391569f, 0.918216106880274f},
        {0.17364817766693041f, 0.98480775301220802f},
        {-0.058144828910475774f, 0.99830815827126818f},
        {-0.28680323271109021f, 0.9579895123154889f},
        {-0.5f, 0.86602540378443871f},
        {-0.68624163786873349f, 0.72737364157304885f},
        {-0.83548781141293627f, 0.54950897807080623f},
        {-0.93969262078590832f, 0.34202014332566888f},
        {-0.99323835774194302f, 0.11609291412522993f},
        {-0.99323835774194302f, -0.11609291412523012f},
        {-0.93969262078590854f, -0.34202014332566821f},
        {-0.83548781141293649f, -0.54950897807080601f},
    };
    float2 b[3][9];
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2) {
#pragma unroll
        for (int j1 = 0; j1 < 9; ++j1) b[j2][j1] = a[3 * j1 + j2];
        dft9f(b[j2]);
    }
#pragma unroll
    for (int k1 = 1; k1 < 9; ++k1) {
        b[1][k1] = cmulc(b[1][k1], make_float2(W27[k1][0], W27[k1][1]));
        b[2][k1] = cmulc(b[2][k1], make_float2(W27[2 * k1][0], W27[2 * k1][1]));
    }
#pragma unroll
    for (int k1 = 0; k1 < 9; ++k1) {
        dft3f(b[0][k1], b[1][k1], b[2][k1]);
        a[k1] = b[0][k1];
        a[k1 + 9] = b[1][k1];
        a[k1 + 18] = b[2][k1];
    }
}

constexpr int CT = 192;          // threads of a column-pass workgroup (12 x 16; q = 0..8 work)
constexpr uint32_t CSI = 17;     // tile row stride in points (16 columns + 1)

// STATS: minimum, maximum and the fractional flag of the frame ride along (optimizer/utils.rs:39-89): the pass reads
// every sample exactly once, as (x[j], x[j + 1]) pairs, so k_large_stats' separate walk over the samples is not
// needed (the run starts, which need every sample's predecessor, are counted by k_large_poly1's contiguous walk).
// Needs the paired-load form (even padding, 16-byte aligned frame): the caller falls back to k_large_stats otherwise.
template <bool STATS>
__global__ __launch_bounds__(CT, 3) void k_large_cols243(const double *__restrict__ samples,
                                                      const DevFrame *__restrict__ frames,
                                                      const uint32_t *__restrict__ ids,
                                                      const DevPlan *__restrict__ plans,
                                                      const float2 *__restrict__ twpool,
                                                      unsigned char *__restrict__ ws_base, uint64_t ws_stride)
{
    __shared__ float2 w1[243];
    __shared__ float2 T[243 * CSI];
    __shared__ double sred[8];
    __shared__ uint32_t ured[12];
    const DevPlan *P;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, P);
    const uint32_t c0 = blockIdx.x * FB;
    if (c0 >= f.M2 || f.M1 != 243) return;
    const LargeWs lay = large_ws_layout(f.n, f.L, P->kcap);
    float2 *Y = (float2 *)(f.ws + lay.o_b);
    if (blockIdx.x == 0 && threadIdx.x == 0) *(uint32_t *)(f.ws + lay.o_cnt) = 0;
    const float2 *tw = twpool + P->tw_off;
    const uint32_t tid = threadIdx.x, c = tid & 15u, q = tid >> 4;
    const uint32_t n2 = c0 + c;
    const bool live = q < 9 && n2 < f.M2;
    for (uint32_t e = tid; e < 243; e += CT) w1[e] = tw[e * (f.M2 * f.sc)];
    float2 a[27];
    double smn = __longlong_as_double(0x7ff0000000000000ll), smx = -smn;
    uint32_t sfrac = 0;
    const bool pairs = f.half && ((f.pre & 1u) == 0) && ((f.n & 1u) == 0) && (((uintptr_t)f.xs & 15u) == 0);
    {
        // fft.rs:184-204 (edge-replicated padding), then `as f32`; even L: point i = (g[2i], g[2i+1])
        if (live && pairs) {
            double2 v[27];
#pragma unroll
            for (int aa = 0; aa < 27; ++aa) {
                const uint32_t i = f.M2 * (9u * aa + q) + n2;
                const int32_t j2 = (int32_t)(2 * i) - (int32_t)f.pre;  // even; a pair never straddles a frame edge
                const int32_t j = j2 < 0 ? 0 : (j2 >= (int32_t)f.n ? (int32_t)f.n - 2 : j2);
                v[aa] = *(const double2 *)(f.xs + j);
            }
#pragma unroll
            for (int aa = 0; aa < 27; ++aa) {
                const uint32_t i = f.M2 * (9u * aa + q) + n2;
                const int32_t j2 = (int32_t)(2 * i) - (int32_t)f.pre;
                if (STATS) {  // (the padding repeats x[0] / x[n-1]: harmless for min / max / fractional)
                    const double x0 = v[aa].x, x1 = v[aa].y;
                    sfrac |= (frac_nonzero(x0) || frac_nonzero(x1)) ? 1u : 0u;
                    if (x0 > smx) smx = x0;
                    if (x0 < smn) smn = x0;
                    if (x1 > smx) smx = x1;
                    if (x1 < smn) smn = x1;
                }
                if (j2 < 0) v[aa].y = v[aa].x;              // x[0], x[0]
                if (j2 >= (int32_t)f.n) v[aa].x = v[aa].y;  // x[n-1], x[n-1]
                a[aa] = make_float2((float)v[aa].x, (float)v[aa].y);
            }
        } else {
            auto g = [&](uint32_t j) -> float {
                int32_t i = (int32_t)j - (int32_t)f.pre;
                i = i < 0 ? 0 : (i >= (int32_t)f.n ? (int32_t)f.n - 1 : i);
                return (float)f.xs[i];
            };
#pragma unroll
            for (int aa = 0; aa < 27; ++aa) {
                const uint32_t i = f.M2 * (9u * aa + q) + n2;
                a[aa] = make_float2(0.0f, 0.0f);
                if (live) a[aa] = f.half ? make_float2(g(2 * i), g(2 * i + 1)) : make_float2(g(i), 0.0f);
            }
        }
    }
    if (STATS) {
        // the tile's record (three wavefronts; lanes that hold no real sample carry the neutral values)
        const double wmn = wave_minmax_f64<true>(smn), wmx = wave_minmax_f64<false>(smx);
        const uint32_t wfr = __ballot(sfrac != 0) ? 1u : 0u;
        if ((tid & 63) == 0) {
            sred[tid >> 6] = wmn; sred[4 + (tid >> 6)] = wmx;
            ured[tid >> 6] = wfr;
        }
    }
    // n1 = 9 a + q, k1 = ka + 27 kq:  W243^{n1 k1} = W27^{a ka} . W243^{q ka} . W9^{q kq}
    dft27f(a);
    __syncthreads();  // w1, sred / ured
    if (STATS && tid == 0 && blockIdx.x < TST_MAX) {
        TileStats t;
        t.mn = sred[0]; t.mx = sred[4];
        for (int w = 1; w < CT / 64; ++w) {
            if (sred[w] < t.mn) t.mn = sred[w];
            if (sred[4 + w] > t.mx) t.mx = sred[4 + w];
        }
        t.frac = (ured[0] | ured[1] | ured[2]) ? 1u : 0u;
        t.runs = t.ibytes = 0;  // (counted by k_large_poly1)
        t.pad = pairs ? 1u : 0u;
        ((TileStats *)(f.ws + lay.o_tst))[blockIdx.x] = t;
    }
    if (q < 9) {
#pragma unroll
        for (int ka = 1; ka < 27; ++ka) a[ka] = cmulc(a[ka], w1[q * ka]);
#pragma unroll
        for (int ka = 0; ka < 27; ++ka) T[(q * 27 + ka) * CSI + c] = a[ka];
    }
    __syncthreads();
    if (!live) return;
    // this thread: ka = q + 9 m (m = 0, 1, 2), every kq: the 27 bins k1 = ka + 27 kq of its column
    float2 tws[27];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int kq = 0; kq < 9; ++kq) {
            const uint32_t k1 = (q + 9u * m) + 27u * kq;
            tws[9 * m + kq] = tw[n2 * k1 * f.sc];  // n2 k1 < M: no wrap
        }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const uint32_t ka = q + 9u * m;
        float2 b[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) b[j] = T[(j * 27 + ka) * CSI + c];
        dft9f(b);
#pragma unroll
        for (int kq = 0; kq < 9; ++kq) {
            const uint32_t k1 = ka + 27u * kq;
            Y[k1 * f.M2 + n2] = cmulc(b[kq], tws[9 * m + kq]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Row pass with the untangle step, the norms and the zero count fused in (what k_large_pre23 does) for plans with
// M1 = 243 and M2 = 9 P, P = 2 .. 32 a power of two (131072 samples: P = 32).  A workgroup of 256 threads takes 8
// rows with their 8 mirror rows (row 0, its own mirror, goes to workgroup 0):
//   step 1  item (row r, v < P): the 9 points n2 = P u + v of the row from memory (consecutive v: runs of 8 P bytes),
//           radix 9 in registers, times W_M2^{v ku}, into LDS as B[r][ku][v] (rows of P + 1 points: conflict-free)
//   step 2  item (r, ku < 9): the P points over v, radix P in registers, back into LDS as Z[r][k2 = ku + 9 kv]
//   step 3  untangle bin k = k1 + 243 k2 with its partner in the mirrored row (arithmetic as in k_large_pre23,
//           operation for operation), norm bits, count of zero bins
// Three workgroup barriers per tile instead of seven, 40 KB of LDS (four workgroups per CU instead of two).
// --------------------------------------------------------------------------------------------
template <int N>
DEVI void dft_pow2f(float2 (&a)[N])  // forward, natural order in and out, N = 1 .. 32 a power of two
{
    if constexpr (N == 2) {
        const float2 x = a[0], y = a[1];
        a[0] = make_float2(x.x + y.x, x.y + y.y);
        a[1] = make_float2(x.x - y.x, x.y - y.y);
    } else if constexpr (N > 2) {
        constexpr float W32[16][2] = {
            {1.0f, 0.0f},
            {0.98078528040323043f, 0.19509032201612825f},
            {0.92387953251128674f, 0.38268343236508978f},
            {0.83146961230254524f, 0.55557023301960218f},
            {0.70710678118654757f, 0.70710678118654746f},
            {0.55557023301960229f, 0.83146961230254524f},
            {0.38268343236508984f, 0.92387953251128674f},
            {0.19509032201612833f, 0.98078528040323043f},
            {0.0f, 1.0f},
            {-0.19509032201612819f, 0.98078528040323043f},
            {-0.38268343236508973f, 0.92387953251128674f},
            {-0.55557023301960196f, 0.83146961230254546f},
            {-0.70710678118654746f, 0.70710678118654757f},
            {-0.83146961230254535f, 0.55557023301960218f},
            {-0.92387953251128674f, 0.38268343236508989f},
            {-0.98078528040323043f, 0.19509032201612861f},
        };
        float2 e[N / 2], o[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
        dft_pow2f<N / 2>(e);
        dft_pow2f<N / 2>(o);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            float2 t;
            if (k == 0) t = o[0];
            else if (4 * k == N) t = make_float2(o[k].y, -o[k].x);  // times -i
            else t = cmulc(o[k], make_float2(W32[k * (32 / N)][0], W32[k * (32 / N)][1]));
            a[k] = make_float2(e[k].x + t.x, e[k].y + t.y);
            a[k + N / 2] = make_float2(e[k].x - t.x, e[k].y - t.y);
        }
    }
}

constexpr int RT = 256;  // threads of a row-pass workgroup

// A frame whose every sample is one non-zero value (and whose first sample is a number): k_large_decide1 emits Constant
// from the column tiles' statistics alone (frame/mod.rs:82-88) and nothing ever reads such a frame's rows, norms or
// polynomial pieces -- a fifth of configs[3]'s series, and the row launch is the large tier's longest.  (A zero
// extreme or a NaN first sample send the frame to the general kernel, which does read them: no skip.)
DEVI bool frame_is_constant(const double *xs, const unsigned char *ws, const LargeWs &lay, uint32_t m2)
{
    const uint32_t nt = (m2 + FB - 1) / FB;
    const TileStats *q = (const TileStats *)(ws + lay.o_tst);
    const double x0 = xs[0];
    double mn = q[0].mn, mx = q[0].mx;
    for (uint32_t t = 1; t < nt; ++t) {
        mn = fmin(mn, q[t].mn);
        mx = fmax(mx, q[t].mx);
    }
    return mn == mx && mn != 0.0 && x0 == x0;
}

// First trip of the polynomial ladder (polynomial.rs:209-277: points = max(3, n / 100), the plan's pstep[0] / pK[0]) for
// one piece of 1024 samples, by a 256-thread workgroup -- the arithmetic of k_large_poly1, sample for sample; the
// piece's share of the MAPE sum and its run starts (rle.rs:142-189) go to the workspace as plain stores.  Runs as extra
// workgroups of the row pass's launch (the two are independent: one less launch boundary, and the latencies of one
// hide behind the other's).  Clamp range from the column tiles' records (k_large_cols243<true>).
constexpr uint32_t PCH = 4096;  // = LCH: the pieces are k_large_poly1's chunks, their sums land where its sums would
DEVI void poly1_piece(const double *xs, const DevPlan &P, unsigned char *ws, const LargeWs &lay, uint32_t piece,
                      unsigned char *lds)
{
    double4 *hbt = (double4 *)lds;                    // 256 x 32 B
    double2 *mms = (double2 *)(lds + 8192);           // 512 segments at most (step >= 8)
    double *red = (double *)(lds + 8192 + 16 * 512);
    const uint32_t tid = threadIdx.x;
    const uint32_t n = P.n, c0 = piece * PCH;
    if (c0 >= n) return;
    const uint32_t c1 = min(c0 + PCH, n);
    const uint32_t step = P.pstep[0], K = P.pK[0];
    if (step < 16 || step > 256 || K < 2) return;  // (the fast path's frames have step = 100)
    // clamp range: the column tiles' records, one per lane of the first wavefront, then broadcast
    if (tid < 64) {
        const uint32_t nt = (P.f4_m2 + FB - 1) / FB;
        double mn = __longlong_as_double(0x7ff0000000000000ll), mx = -mn;
        if (tid < nt) {
            const TileStats q = ((const TileStats *)(ws + lay.o_tst))[tid];
            mn = q.mn; mx = q.mx;
        }
        mn = wave_minmax_f64<true>(mn);
        mx = wave_minmax_f64<false>(mx);
        if (tid == 0) { red[40] = mn; red[41] = mx; }
    }
    const uint32_t magic = P.pmagic[0];
    const uint32_t gapL = (n - 1) - (K - 2) * step;
    const double stepd = (double)step, gapLd = (double)gapL;
    const double ry = 1.0 / stepd, ryL = 1.0 / gapLd;
    uint32_t sgA = __umulhi(c0, magic), sgB = __umulhi(c1 - 1, magic);
    if (sgA > K - 2) sgA = K - 2;
    if (sgB > K - 2) sgB = K - 2;
    for (uint32_t sg = sgA + tid; sg <= sgB; sg += RT) {
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
    for (uint32_t r = tid; r < step; r += RT) {
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
    const double smin = red[40], smax = red[41];
    double s = 0.0;
    uint32_t runs = 0, ib = 0;
    constexpr uint32_t SPB4 = 4;  // samples of a thread in flight
    for (uint32_t b0 = c0; b0 < c1; b0 += SPB4 * RT) {
        double g[SPB4], v0[SPB4], v1[SPB4], pv[SPB4];
#pragma unroll
        for (uint32_t u = 0; u < SPB4; ++u) {
            const uint32_t i = b0 + u * RT + tid;
            g[u] = v0[u] = v1[u] = pv[u] = 0.0;
            if (i < c1) {
                uint32_t sg = __umulhi(i, magic);
                if (sg > K - 2) sg = K - 2;
                const uint32_t t0i = sg * step;
                g[u] = xs[i];
                v0[u] = xs[t0i];
                v1[u] = xs[(sg == K - 2) ? (n - 1) : t0i + step];
                if (i) pv[u] = xs[i - 1];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < SPB4; ++u) {
            const uint32_t i = b0 + u * RT + tid;
            // where the runs start, one bit per sample (rle.rs:142-189; read by k_large_decide1 when RLE can win): a
            // wavefront's 64 samples are consecutive and start at a multiple of 64 -- its ballot is one word of the map
            const bool rstart = i < c1 && (i == 0 || g[u] != pv[u]);
            const unsigned long long rmask = __ballot(rstart);
            if ((tid & 63u) == 0 && i < c1) ((unsigned long long *)(ws + lay.o_rbm))[i >> 6] = rmask;
            if (i >= c1) continue;
            if (rstart) { ++runs; ib += vlen(i); }
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
    }
    int parity = 0;
    s = block_sum_f64<RT / 64>(s, red, parity);
    runs = block_sum_u32<RT / 64>(runs, red, parity);
    ib = block_sum_u32<RT / 64>(ib, red, parity);
    if (tid == 0) {
        ((double *)(ws + lay.o_part))[piece] = s;  // where k_large_poly1 leaves its chunk sums
        ((uint2 *)(ws + lay.o_part + 1536))[piece] = make_uint2(runs, ib);
    }
}

template <int P>
__global__ __launch_bounds__(RT) void k_large_rows9p(const double *__restrict__ samples,
                                                      const DevFrame *__restrict__ frames,
                                                      const uint32_t *__restrict__ ids,
                                                      const DevPlan *__restrict__ plans,
                                                      const float2 *__restrict__ twpool,
                                                      unsigned char *__restrict__ ws_base, uint64_t ws_stride,
                                                      int sparse_inv, uint32_t row_tiles)
{
    constexpr uint32_t M2 = 9 * P, BS = P + 1, ZS = M2 + 1;
    constexpr uint32_t NPT0 = (FB * 9 * BS > FB * ZS) ? FB * 9 * BS : FB * ZS;
    constexpr uint32_t NPT = NPT0 > 2112 ? NPT0 : 2112;  // (the polynomial pieces borrow the tile buffer: 16.5 KB)
    __shared__ float2 w2[M2];
    __shared__ __attribute__((aligned(16))) float2 T[NPT];
    static_assert(NPT * 8 >= 8192 + 16 * 512 + 512, "the polynomial pieces borrow the tile buffer (basis, tangents, 48 doubles)");
    const DevPlan *Pl;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, Pl);
    if (blockIdx.x >= row_tiles) {  // workgroups behind the row tiles: pieces of the first polynomial trip
        if (f.M2 != 9 * P) return;  // (another launch's frame)
        if ((sparse_inv & 2) && frame_is_constant(f.xs, f.ws, large_ws_layout(f.n, f.L, Pl->kcap), f.M2)) return;
        poly1_piece(f.xs, *Pl, f.ws, large_ws_layout(f.n, f.L, Pl->kcap), blockIdx.x - row_tiles, (unsigned char *)T);
        return;
    }
    const uint32_t M1 = f.M1, M = f.M;
    if (f.M2 != M2) return;
    if ((sparse_inv & 2) && frame_is_constant(f.xs, f.ws, large_ws_layout(f.n, f.L, Pl->kcap), f.M2)) return;
    const uint32_t half_pairs = (M1 - 1) / 2;
    const LargeWs lay = large_ws_layout(f.n, f.L, Pl->kcap);
    const float2 *Y = (const float2 *)(f.ws + lay.o_b);
    float2 *spec = (float2 *)(f.ws + lay.o_a);
    uint32_t *nbits = (uint32_t *)(f.ws + lay.o_nb);
    float2 *Xs = (float2 *)(f.ws + lay.o_x);
    const float2 *tw = twpool + Pl->tw_off;
    const uint32_t tid = threadIdx.x;
    for (uint32_t e = tid; e < M2; e += RT) w2[e] = tw[e * (M1 * f.sc)];
    // Short rows (P <= 8): a workgroup takes G tiles one after the other -- a tile of 16 rows is 288 points at P = 2, and a
    // workgroup per tile spends its life on the launch, the table and three barriers (the other workgroups of such a
    // group leave at once)
    constexpr uint32_t G = (P <= 4) ? 4u : (P == 8 ? 2u : 1u);
    if (blockIdx.x % G) return;
    uint32_t zeros = 0;
    __syncthreads();  // w2
    auto do_tile = [&](const uint32_t bx) -> bool {
        uint32_t a0 = 0, cnt = 0, nrow;
        if (bx == 0) {
            nrow = (M1 % 2 == 0 && M1 >= 2) ? 2u : 1u;
        } else {
            const uint32_t first = (bx - 1) * FBH;
            if (first >= half_pairs) return false;
            a0 = 1 + first;
            cnt = min(FBH, half_pairs - first);
            nrow = 2 * cnt;
        }
        auto row_of = [&](uint32_t r) -> uint32_t {
            if (bx == 0) return r == 0 ? 0u : M1 / 2;
            return r < cnt ? a0 + r : M1 - a0 - cnt + 1 + (r - cnt);
        };
        // ---- step 1 ----
        constexpr int IT1 = (FB * P + RT - 1) / RT;  // items per thread
        float2 b[IT1][9];
    #pragma unroll
        for (int s = 0; s < IT1; ++s) {
            const uint32_t it = tid + s * RT, r = it / P, v = it % P;
    #pragma unroll
            for (int u = 0; u < 9; ++u) {
                b[s][u] = make_float2(0.0f, 0.0f);
                if (it < FB * P && r < nrow) b[s][u] = Y[row_of(r) * M2 + P * u + v];
            }
        }
        __syncthreads();  // w2
    #pragma unroll
        for (int s = 0; s < IT1; ++s) {
            const uint32_t it = tid + s * RT, r = it / P, v = it % P;
            if (it < FB * P && r < nrow) {
                dft9f(b[s]);
    #pragma unroll
                for (int ku = 0; ku < 9; ++ku) {
                    const float2 z = ku ? cmulc(b[s][ku], w2[v * ku]) : b[s][0];
                    T[(r * 9 + ku) * BS + v] = z;
                }
            }
        }
        __syncthreads();
        // ---- step 2: item (r, ku), r = tid & 15 ----
        const uint32_t r = tid & (FB - 1), ku2 = tid >> 4;
        const bool live = r < nrow;
        {
            float2 c[P];
            const bool act = live && ku2 < 9;
            if (act) {
    #pragma unroll
                for (int v = 0; v < P; ++v) c[v] = T[(r * 9 + ku2) * BS + v];
                dft_pow2f<P>(c);
            }
            __syncthreads();
            if (act) {
    #pragma unroll
                for (int kv = 0; kv < P; ++kv) T[r * ZS + ku2 + 9 * kv] = c[kv];
            }
        }
        // twiddles of this thread's bins (issued before the barrier: a dependent global load per bin otherwise)
        constexpr uint32_t PU2 = (M2 + (RT / FB) - 1) / (RT / FB);
        const uint32_t kc0 = tid >> 4;
        const uint32_t k1 = live ? row_of(r) : 0u;
        float2 twk[PU2];
    #pragma unroll
        for (uint32_t u = 0; u < PU2; ++u) {
            const uint32_t k2 = kc0 + u * (RT / FB);
            twk[u] = make_float2(1.0f, 0.0f);
            if (live && k2 < M2 && f.half) twk[u] = tw[k1 + M1 * k2];
        }
        __syncthreads();
        // ---- step 3 ----
        const bool dense = !((sparse_inv & 1) && Pl->sp_mf);
        auto finish = [&](uint32_t k, float2 z) {
            spec[k] = z;
            nbits[k] = __float_as_uint((float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y));
            zeros += (z.x != 0.0f || z.y != 0.0f) ? 0u : 1u;
            if (dense) Xs[k] = make_float2(0.0f, 0.0f);
        };
        const bool row0 = bx == 0 && r == 0;
        const float2 *Rrow = T + r * ZS;
        const float2 *Rpart = T + ((bx == 0) ? r : (nrow - 1 - r)) * ZS;
        auto untangle = [&](float2 zk, float2 zm, float2 wk) -> float2 {  // see fft_untangle (atsc_kernels.hip)
            const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);
            const float2 bb = make_float2(zk.x - zm.x, zk.y + zm.y);
            const float2 t = cmulc(make_float2(bb.y, -bb.x), wk);
            return make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
        };
    #pragma unroll
        for (uint32_t u = 0; u < PU2; ++u) {
            const uint32_t k2 = kc0 + u * (RT / FB);
            if (!live || k2 >= M2) continue;
            const uint32_t k = k1 + M1 * k2;
            const float2 zk = Rrow[k2];
            if (!f.half) {
                if (k < f.bins) finish(k, zk);
                else spec[k] = zk;
                continue;
            }
            const float2 zm = Rpart[row0 ? (k2 == 0 ? 0u : M2 - k2) : (M2 - 1 - k2)];
            finish(k, untangle(zk, zm, twk[u]));
            if (k == 0) finish(M, untangle(zk, zk, tw[M]));
        }
        return true;
    };
    if constexpr (G == 1) {
        (void)do_tile(blockIdx.x);
    } else {
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t bx = blockIdx.x + g;
            if (bx >= row_tiles) break;
            if (g) __syncthreads();  // the tile buffer is free again
            if (!do_tile(bx)) break;
        }
    }
    if (__ballot(zeros != 0)) {
        zeros = wave_sum_u32(zeros);
        if ((tid & 63) == 0) atomicAdd((uint32_t *)(f.ws + lay.o_cnt), zeros);
    }
}

// The same row pass for the two shortest row dimensions (M2 = 18, 36: frames of 8192 and 16384 samples), one THREAD per
// row: a frame's 243 rows fit one workgroup, a row's 9 P points fit a thread's registers, and what k_large_rows9p<P> does
// through three barriers and two trips through LDS per 16-row tile -- with most of its 256 threads idle on rows this
// short -- is a straight run here: load the row, radix 9 over u, twiddle, radix P over v (the same butterflies and the
// same twiddles, operation for operation), one exchange through LDS for the untangle step's partner row (thread r
// needs row M1 - r mirrored), stores with consecutive threads on consecutive bins.  1280 frames of 8192 samples:
// 93 -> ~15 us.  Workgroups behind the first (blockIdx.x >= 1) are the first polynomial trip's pieces, as in
// k_large_rows9p.
template <int P>
__global__ __launch_bounds__(RT) void k_large_rows_thread(const double *__restrict__ samples,
                                                           const DevFrame *__restrict__ frames,
                                                           const uint32_t *__restrict__ ids,
                                                           const DevPlan *__restrict__ plans,
                                                           const float2 *__restrict__ twpool,
                                                           unsigned char *__restrict__ ws_base, uint64_t ws_stride,
                                                           int sparse_inv)
{
    constexpr uint32_t M2 = 9 * P, TS = 244;  // TS: row stride of the exchange buffer (one column of all rows, padded)
    constexpr uint32_t NPT0 = M2 * TS;
    constexpr uint32_t NPT = NPT0 > 2112 ? NPT0 : 2112;  // (the polynomial pieces borrow the buffer: 16.5 KB)
    __shared__ float2 w2[M2];
    __shared__ __attribute__((aligned(16))) float2 T[NPT];
    const DevPlan *Pl;
    const PreFrame f = pre_frame(samples, frames, ids, plans, ws_base, ws_stride, Pl);
    if (f.M2 != M2 || f.M1 != 243 || !f.half) return;
    const LargeWs lay = large_ws_layout(f.n, f.L, Pl->kcap);
    if ((sparse_inv & 2) && frame_is_constant(f.xs, f.ws, lay, f.M2)) return;
    if (blockIdx.x >= 1) {
        poly1_piece(f.xs, *Pl, f.ws, lay, blockIdx.x - 1, (unsigned char *)T);
        return;
    }
    constexpr uint32_t M1 = 243;
    const uint32_t M = f.M;
    const float2 *Y = (const float2 *)(f.ws + lay.o_b);
    float2 *spec = (float2 *)(f.ws + lay.o_a);
    uint32_t *nbits = (uint32_t *)(f.ws + lay.o_nb);
    float2 *Xs = (float2 *)(f.ws + lay.o_x);
    const float2 *tw = twpool + Pl->tw_off;
    const uint32_t r = threadIdx.x;
    const bool live = r < M1;
    for (uint32_t e = r; e < M2; e += RT) w2[e] = tw[e * (M1 * f.sc)];
    float2 z[M2];  // in: point n2 = P u + v at [P u + v]; after the transform: bin k2 = ku + 9 kv at [ku + 9 kv]
#pragma unroll
    for (uint32_t j = 0; j < M2; ++j) z[j] = live ? Y[r * M2 + j] : make_float2(0.0f, 0.0f);
    __syncthreads();  // w2
    {
        float2 c[9][P];  // [ku][v]
#pragma unroll
        for (int v = 0; v < P; ++v) {
            float2 b[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) b[u] = z[P * u + v];
            dft9f(b);
#pragma unroll
            for (int ku = 0; ku < 9; ++ku) c[ku][v] = ku ? cmulc(b[ku], w2[v * ku]) : b[0];
        }
#pragma unroll
        for (int ku = 0; ku < 9; ++ku) {
            dft_pow2f<P>(c[ku]);
#pragma unroll
            for (int kv = 0; kv < P; ++kv) z[ku + 9 * kv] = c[ku][kv];
        }
    }
    if (live) {
#pragma unroll
        for (uint32_t k2 = 0; k2 < M2; ++k2) T[k2 * TS + r] = z[k2];
    }
    __syncthreads();
    if (!live) return;
    const bool dense = !((sparse_inv & 1) && Pl->sp_mf);
    uint32_t zeros = 0;
    auto finish = [&](uint32_t k, float2 v) {
        spec[k] = v;
        nbits[k] = __float_as_uint((float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y));
        zeros += (v.x != 0.0f || v.y != 0.0f) ? 0u : 1u;
        if (dense) Xs[k] = make_float2(0.0f, 0.0f);
    };
    auto untangle = [&](float2 zk, float2 zm, float2 wk) -> float2 {  // as k_large_rows9p
        const float2 a = make_float2(zk.x + zm.x, zk.y - zm.y);
        const float2 bb = make_float2(zk.x - zm.x, zk.y + zm.y);
        const float2 t = cmulc(make_float2(bb.y, -bb.x), wk);
        return make_float2(0.5f * a.x + 0.5f * t.x, 0.5f * a.y + 0.5f * t.y);
    };
    // bin k = r + 243 k2 pairs with M - k = (243 - r) + 243 (M2 - 1 - k2)  (r >= 1);  row 0: M - k = 243 (M2 - k2)
    const uint32_t pr = r ? M1 - r : 0u;
#pragma unroll
    for (uint32_t k2 = 0; k2 < M2; ++k2) {
        const uint32_t pk2 = r ? (M2 - 1 - k2) : (k2 ? M2 - k2 : 0u);
        const float2 zm = T[pk2 * TS + pr];
        const uint32_t k = r + M1 * k2;
        finish(k, untangle(z[k2], zm, tw[k]));  // (the untangle twiddle W_L^k: consecutive threads, consecutive entries)
        if (k == 0) finish(M, untangle(z[0], z[0], tw[M]));
    }
    if (__ballot(zeros != 0)) {
        zeros = wave_sum_u32(zeros);
        if ((r & 63) == 0) atomicAdd((uint32_t *)(f.ws + lay.o_cnt), zeros);
    }
}
