/*
 * atsc_oracle.c -- TEST INFRASTRUCTURE ONLY (see atsc_oracle.h).
 *
 * Plain-C restatement of the instaclustr/atsc per-frame compressor path.
 * Every function cites the reference file:line it follows (paths relative
 * to /root/reference/).  Build with -ffp-contract=off: Rust never contracts
 * a*b+c into an fma, and the byte-exact known-answer tests depend on that.
 *
 * Rust semantics restated here on purpose:
 *   - `as` float->int casts saturate and map NaN to 0
 *   - f64::round is half-away-from-zero (C round())
 *   - usize->u16 / usize->u8 `as` casts wrap
 *   - BinaryHeap::from(vec) + pop() order (std 1.81) for the FFT top-K
 */
#include "atsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small utilities                                                          */
/* ------------------------------------------------------------------------ */

void orc_free(void *p) { free(p); }

static int buf_reserve(orc_buf *b, size_t extra)
{
    if (b->len + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap * 2 : 64;
    while (nc < b->len + extra) nc *= 2;
    uint8_t *np = (uint8_t *)realloc(b->ptr, nc);
    if (!np) return -1;
    b->ptr = np;
    b->cap = nc;
    return 0;
}
static void buf_init(orc_buf *b) { b->ptr = NULL; b->len = 0; b->cap = 0; }
static void buf_u8(orc_buf *b, uint8_t v)
{
    if (buf_reserve(b, 1)) return;
    b->ptr[b->len++] = v;
}
static void buf_bytes(orc_buf *b, const void *p, size_t n)
{
    if (buf_reserve(b, n)) return;
    memcpy(b->ptr + b->len, p, n);
    b->len += n;
}
/* bincode 2.0.0-rc.3 config::standard() varint (pinned by
 * data.rs:152, polynomial.rs:451,466) */
static void buf_varint(orc_buf *b, uint64_t v)
{
    if (v < 251) {
        buf_u8(b, (uint8_t)v);
    } else if (v < (1ull << 16)) {
        buf_u8(b, 251);
        uint16_t x = (uint16_t)v;
        buf_bytes(b, &x, 2);
    } else if (v < (1ull << 32)) {
        buf_u8(b, 252);
        uint32_t x = (uint32_t)v;
        buf_bytes(b, &x, 4);
    } else {
        buf_u8(b, 253);
        buf_bytes(b, &v, 8);
    }
}
static void buf_zigzag(orc_buf *b, int64_t v)
{
    uint64_t u = v >= 0 ? ((uint64_t)v << 1) : ((~(uint64_t)v << 1) | 1ull);
    buf_varint(b, u);
}
static void buf_f32(orc_buf *b, float v) { buf_bytes(b, &v, 4); }
static void buf_f64(orc_buf *b, double v) { buf_bytes(b, &v, 8); }

typedef struct {
    const uint8_t *p;
    size_t len;
    size_t pos;
    int err;
} rd;
static uint8_t rd_u8(rd *r)
{
    if (r->pos + 1 > r->len) { r->err = 1; return 0; }
    return r->p[r->pos++];
}
static void rd_bytes(rd *r, void *out, size_t n)
{
    if (r->pos + n > r->len) { r->err = 1; memset(out, 0, n); return; }
    memcpy(out, r->p + r->pos, n);
    r->pos += n;
}
static uint64_t rd_varint(rd *r)
{
    uint8_t t = rd_u8(r);
    if (t < 251) return t;
    if (t == 251) { uint16_t x; rd_bytes(r, &x, 2); return x; }
    if (t == 252) { uint32_t x; rd_bytes(r, &x, 4); return x; }
    if (t == 253) { uint64_t x; rd_bytes(r, &x, 8); return x; }
    r->err = 1;
    return 0;
}
static int64_t rd_zigzag(rd *r)
{
    uint64_t u = rd_varint(r);
    return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
}
static float rd_f32(rd *r) { float v; rd_bytes(r, &v, 4); return v; }
static double rd_f64(rd *r) { double v; rd_bytes(r, &v, 8); return v; }

/* Rust `as` casts from f64 (saturating, NaN -> 0) */
static int64_t sat_i64(double x)
{
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
static int32_t sat_i32(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
static int16_t sat_i16(double x)
{
    if (x != x) return 0;
    if (x >= 32767.0) return INT16_MAX;
    if (x <= -32768.0) return INT16_MIN;
    return (int16_t)x;
}
static uint8_t sat_u8(double x)
{
    if (x != x) return 0;
    if (x >= 255.0) return 255;
    if (x <= 0.0) return 0;
    return (uint8_t)x;
}

/* ------------------------------------------------------------------------ */
/* utils/mod.rs, utils/error.rs                                             */
/* ------------------------------------------------------------------------ */

/* utils/mod.rs:41-49 */
int orc_is_decomposable(size_t n)
{
    if (n == 0) return 0; /* the reference would spin forever on 0; never called with it */
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    return n == 1;
}
/* utils/mod.rs:32-38 */
size_t orc_next_size(size_t n)
{
    n += 1;
    while (!orc_is_decomposable(n)) n += 1;
    return n;
}
/* utils/mod.rs:24-29 */
size_t orc_prev_power_of_two(size_t n)
{
    uint64_t v = (uint64_t)n | 1;
    int hi = 63 - __builtin_clzll(v);
    return ((size_t)1 << hi) & n;
}
static double pow10i(uint32_t d)
{
    int32_t y = 1;
    for (uint32_t i = 0; i < d; i++) y *= 10;
    return (double)y;
}
/* utils/mod.rs:61-64 */
double orc_round_f64(double x, uint32_t d)
{
    double y = pow10i(d);
    return round(x * y) / y;
}
/* utils/mod.rs:66-74 : min is checked first */
double orc_round_and_limit_f64(double x, double mn, double mx, uint32_t d)
{
    double y = pow10i(d);
    double out = round(x * y) / y;
    if (out < mn) return mn;
    if (out > mx) return mx;
    return out;
}
/* utils/error.rs:104-116 : sequential f64 sum of |(gen-orig)/orig| / n */
double orc_error_mape(const double *orig, const double *gen, size_t n)
{
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += fabs((gen[i] - orig[i]) / orig[i]);
    return s / (double)n;
}

/* ------------------------------------------------------------------------ */
/* optimizer/utils.rs : DataStats                                           */
/* ------------------------------------------------------------------------ */

/* optimizer/utils.rs:115-160 */
void orc_split_n(double x, int64_t *ip, double *frac)
{
    const double FRACT_SCALE = 1.0 / (65536.0 * 65536.0 * 65536.0 * 65536.0);
    const uint32_t STORED = 52;
    const uint64_t MASK = (1ull << STORED) - 1;
    const uint64_t MSB = 1ull << STORED;
    const int32_t BIAS = 1023;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int is_negative = ((int64_t)bits) < 0;
    int32_t exponent = (int32_t)((uint32_t)(bits >> STORED) & 0x7ffu);
    uint64_t mant_u = (bits & MASK) | MSB;
    int64_t mantissa = is_negative ? -(int64_t)mant_u : (int64_t)mant_u;
    int32_t shl = exponent + (64 - 53 - BIAS + 1);
    if (shl <= 0) {
        int32_t shr = -shl;
        if (shr < 64) {
            *ip = 0;
            *frac = (double)(((uint64_t)mantissa) >> shr) * FRACT_SCALE;
        } else {
            *ip = 0;
            *frac = 0.0;
        }
    } else if (shl < 64) {
        *ip = mantissa >> (64 - shl); /* arithmetic shift, as Rust i64 >> */
        *frac = (double)(((uint64_t)mantissa) << shl) * FRACT_SCALE;
    } else if (shl < 128) {
        /* Rust: mantissa << (shl - 64) on i64 (wrapping in release) */
        *ip = (int64_t)(((uint64_t)mantissa) << (shl - 64));
        *frac = 0.0;
    } else {
        *ip = 0;
        *frac = 0.0;
    }
}

/* optimizer/utils.rs:91-113 */
static int bitdepth_of(int64_t max_int, int64_t min_int)
{
    int bd = max_int <= 255 ? 8 : max_int <= 32767 ? 16 : max_int <= 2147483647LL ? 32 : 64;
    int bs = (min_int >= 0 && min_int <= 255) ? 8
             : min_int >= -32768              ? 16
             : min_int >= -2147483648LL       ? 32
                                              : 64;
    int m = bd > bs ? bd : bs;
    return m == 8 ? ORC_BD_U8 : m == 16 ? ORC_BD_I16 : m == 32 ? ORC_BD_I32 : ORC_BD_F64;
}

/* optimizer/utils.rs:39-89 */
void orc_stats_new(const double *x, size_t n, orc_stats *s)
{
    double mn = x[0], mx = x[0], mean = 0.0;
    size_t mnl = 0, mxl = 0;
    int fractional = 0;
    for (size_t i = 0; i < n; i++) {
        double v = x[i];
        mean += v;
        int64_t ip;
        double fr;
        orc_split_n(v, &ip, &fr);
        if (fr != 0.0) fractional = 1;
        if (v > mx) { mx = v; mxl = i; }
        if (v < mn) { mn = v; mnl = i; }
    }
    mean /= (double)n;
    int64_t max_int, min_int;
    double fr;
    orc_split_n(mx, &max_int, &fr);
    orc_split_n(mn, &min_int, &fr);
    s->bitdepth = fractional ? ORC_BD_F64 : bitdepth_of(max_int, min_int);
    s->max = mx;
    s->max_loc = mxl;
    s->min = mn;
    s->min_loc = mnl;
    s->mean = mean;
    s->fractional = fractional;
}

/* ------------------------------------------------------------------------ */
/* optimizer/mod.rs : cleaning + chunking                                   */
/* ------------------------------------------------------------------------ */

/* optimizer/mod.rs:64-71 */
size_t orc_clean_data(const double *x, size_t n, double *out)
{
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (!(isnan(x[i]) || isinf(x[i]))) out[k++] = x[i];
    return k;
}
/* optimizer/mod.rs:78-98 */
size_t orc_chunk_sizes(size_t len, size_t *out, size_t cap)
{
    size_t k = 0;
    while (len > 0) {
        size_t sz;
        if (len >= 131072) sz = 131072;
        else if (len <= 512) sz = len;
        else sz = orc_prev_power_of_two(len);
        if (k < cap) out[k] = sz;
        k++;
        len -= sz;
    }
    return k;
}

/* ------------------------------------------------------------------------ */
/* constant.rs                                                              */
/* ------------------------------------------------------------------------ */

/* constant.rs:37-64 (Encode), :135-139 (constant_compressor stores stats.min) */
static void constant_encode(const orc_stats *st, orc_buf *b)
{
    buf_u8(b, 30);
    buf_varint(b, (uint64_t)st->bitdepth);
    switch (st->bitdepth) {
    case ORC_BD_U8: buf_u8(b, sat_u8(st->min)); break;
    case ORC_BD_I16: buf_zigzag(b, sat_i16(st->min)); break;
    case ORC_BD_I32: buf_zigzag(b, sat_i32(st->min)); break;
    default: buf_f64(b, st->min); break;
    }
}
int orc_constant(const double *x, size_t n, orc_buf *out)
{
    orc_stats st;
    buf_init(out);
    orc_stats_new(x, n, &st);
    constant_encode(&st, out);
    return 0;
}
/* constant.rs:66-102 (Decode), :141-144 */
static int constant_to_data(const uint8_t *d, size_t len, size_t n, double *out)
{
    rd r = {d, len, 0, 0};
    (void)rd_u8(&r);
    uint64_t bd = rd_varint(&r);
    double c;
    switch (bd) {
    case ORC_BD_U8: c = (double)rd_u8(&r); break;
    case ORC_BD_I16: c = (double)(int16_t)rd_zigzag(&r); break;
    case ORC_BD_I32: c = (double)(int32_t)rd_zigzag(&r); break;
    case ORC_BD_F64: c = rd_f64(&r); break;
    default: return -2;
    }
    if (r.err) return -2;
    for (size_t i = 0; i < n; i++) out[i] = c;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* noop.rs                                                                  */
/* ------------------------------------------------------------------------ */

/* noop.rs:37-43 (round() as i64), :23-27 derive(Encode): id u8, Vec<i64> */
int orc_noop(const double *x, size_t n, orc_buf *out)
{
    buf_init(out);
    buf_u8(out, 250);
    buf_varint(out, n);
    for (size_t i = 0; i < n; i++) buf_zigzag(out, sat_i64(round(x[i])));
    return 0;
}
/* noop.rs:79-83 : returns the stored vector, whatever sample_number says */
static int noop_to_data(const uint8_t *d, size_t len, double **out, size_t *out_n)
{
    rd r = {d, len, 0, 0};
    (void)rd_u8(&r);
    uint64_t cnt = rd_varint(&r);
    if (r.err || cnt > len) return -2;
    double *o = (double *)malloc((cnt ? cnt : 1) * sizeof(double));
    for (uint64_t i = 0; i < cnt; i++) o[i] = (double)rd_zigzag(&r);
    if (r.err) { free(o); return -2; }
    *out = o;
    *out_n = cnt;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* rle.rs                                                                   */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint64_t key;
    uint64_t idx;
} rle_run;
static int rle_run_cmp(const void *a, const void *b)
{
    const rle_run *x = (const rle_run *)a, *y = (const rle_run *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static void rle_value_encode(orc_buf *b, int bitdepth, double v)
{
    switch (bitdepth) {
    case ORC_BD_U8: buf_u8(b, sat_u8(v)); break;
    case ORC_BD_I16: buf_zigzag(b, sat_i16(v)); break;
    case ORC_BD_I32: buf_zigzag(b, sat_i32(v)); break;
    default: buf_f64(b, v); break;
    }
}
/* rle.rs:142-189 (IndexRLE::new: BTreeMap<u64 bits, Vec<run start>>), :40-67 Encode */
static void rle_encode(const double *x, size_t n, int bitdepth, orc_buf *out)
{
    rle_run *runs = (rle_run *)malloc((n ? n : 1) * sizeof(rle_run));
    size_t nr = 0, cur = 0;
    for (size_t i = 0; i < n; i++) {
        double v = x[i];
        if (i + 1 >= n || x[i + 1] != v) {
            uint64_t bits;
            memcpy(&bits, &v, 8);
            runs[nr].key = bits;
            runs[nr].idx = cur;
            nr++;
            cur = i + 1;
        }
    }
    /* BTreeMap iteration = ascending u64 key; per key the starts were pushed in
     * increasing index order */
    qsort(runs, nr, sizeof(rle_run), rle_run_cmp);
    size_t groups = 0;
    for (size_t i = 0; i < nr; i++)
        if (i == 0 || runs[i].key != runs[i - 1].key) groups++;
    buf_u8(out, 60);
    buf_varint(out, (uint64_t)bitdepth);
    buf_varint(out, groups);
    size_t i = 0;
    while (i < nr) {
        size_t j = i;
        while (j < nr && runs[j].key == runs[i].key) j++;
        double v;
        memcpy(&v, &runs[i].key, 8);
        rle_value_encode(out, bitdepth, v);
        buf_varint(out, j - i);
        for (size_t k = i; k < j; k++) buf_varint(out, runs[k].idx);
        i = j;
    }
    free(runs);
}
int orc_rle(const double *x, size_t n, orc_buf *out)
{
    orc_stats st;
    buf_init(out);
    orc_stats_new(x, n, &st);
    rle_encode(x, n, st.bitdepth, out);
    return 0;
}
typedef struct {
    uint64_t idx;
    double v;
} rle_flat;
static int rle_flat_cmp(const void *a, const void *b)
{
    const rle_flat *x = (const rle_flat *)a, *y = (const rle_flat *)b;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
/* rle.rs:69-101 Decode, :204-236 to_data */
static int rle_to_data(const uint8_t *d, size_t len, size_t n, double *out)
{
    rd r = {d, len, 0, 0};
    (void)rd_u8(&r);
    uint64_t bd = rd_varint(&r);
    uint64_t groups = rd_varint(&r);
    if (r.err || bd > 3 || groups > len) return -2;
    size_t cap = 16, nf = 0;
    rle_flat *fl = (rle_flat *)malloc(cap * sizeof(rle_flat));
    for (uint64_t g = 0; g < groups && !r.err; g++) {
        double v;
        switch (bd) {
        case ORC_BD_U8: v = (double)rd_u8(&r); break;
        case ORC_BD_I16: v = (double)(int16_t)rd_zigzag(&r); break;
        case ORC_BD_I32: v = (double)(int32_t)rd_zigzag(&r); break;
        default: v = rd_f64(&r); break;
        }
        uint64_t cnt = rd_varint(&r);
        if (cnt > len) { r.err = 1; break; }
        for (uint64_t k = 0; k < cnt; k++) {
            if (nf == cap) { cap *= 2; fl = (rle_flat *)realloc(fl, cap * sizeof(rle_flat)); }
            fl[nf].idx = rd_varint(&r);
            fl[nf].v = v;
            nf++;
        }
    }
    if (r.err) { free(fl); return -2; }
    qsort(fl, nf, sizeof(rle_flat), rle_flat_cmp);
    for (size_t i = 0; i < n; i++) out[i] = 0.0;
    for (size_t i = 0; i < nf; i++) {
        uint64_t s = fl[i].idx;
        uint64_t e = (i + 1 < nf) ? fl[i + 1].idx : n;
        if (e > n) e = n; /* iter_mut().take(end) cannot run past the frame */
        for (uint64_t k = s; k < e; k++) out[k] = fl[i].v;
    }
    free(fl);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* polynomial.rs (Catmull-Rom "Polynomial" + IDW)                            */
/* ------------------------------------------------------------------------ */

typedef struct {
    int idw;          /* PolynomialType: 0 Polynomial, 1 Idw (polynomial.rs:29-34) */
    double *points;   /* data_points */
    size_t npoints;
    double min, max;
    uint8_t point_step;
    int bitdepth;
    double error;
    int has_error;
} poly_t;

/* polynomial.rs:279-305 */
static void poly_compress_hinted(poly_t *p, const double *x, size_t n, size_t points)
{
    if (p->max == p->min) return;
    size_t step = n / points;
    if (step < 1) step = 1;
    size_t cnt = (n + step - 1) / step; /* (0..n).step_by(step) */
    size_t last_pos = (cnt - 1) * step;
    int push_last = (last_pos != n - 1);
    free(p->points);
    p->points = (double *)malloc((cnt + 1) * sizeof(double));
    for (size_t i = 0; i < cnt; i++) p->points[i] = x[i * step];
    if (push_last) p->points[cnt++] = x[n - 1];
    p->npoints = cnt;
    p->point_step = (uint8_t)step; /* `step as u8` wraps */
}

/* polynomial.rs:329-340; returns count, fills pos (cap >= frame_size+1) */
static size_t poly_positions(const poly_t *p, size_t n, size_t *pos)
{
    size_t k = 0;
    size_t step = p->point_step;
    for (size_t v = 0; v < n; v += step) pos[k++] = v;
    if (k == 0 || pos[k - 1] != n - 1) pos[k++] = n - 1;
    return k;
}

/* splines 4.3.1 Interpolate::cubic_hermite for f64 [3P] */
static double cubic_hermite(double t, double x0, double xv, double a0, double av, double b0,
                            double bv, double y0, double yv)
{
    double t2 = t * t;
    double t3 = t2 * t;
    double two_t3 = t3 * 2.0;
    double two_t2 = t2 * 2.0;
    double three_t2 = t2 * 3.0;
    double m0 = (bv - xv) / (b0 - x0) * (b0 - a0);
    double m1 = (yv - av) / (y0 - a0) * (b0 - a0);
    return av * (two_t3 - three_t2 + 1.0) + m0 * (t3 - two_t2 + t) + bv * (three_t2 - two_t3) +
           m1 * (t3 - t2);
}

/* splines 4.3.1 Spline::sample + clamped_sample for strictly increasing keys [3P].
 * interp[i]: 0 Linear, 1 CatmullRom.  returns 1 and *out when Some(_). */
static int spline_clamped_sample(const double *t, const double *v, const uint8_t *interp, size_t K,
                                 double x, double *out)
{
    if (K == 0) return 0;
    int have = 0;
    if (K >= 2) {
        /* search_lower_cp: binary_search_by(key.t.partial_cmp(&x)) */
        size_t lo = 0, hi = K;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (t[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        size_t i = 0;
        int ok = 0;
        if (lo < K && t[lo] == x) {
            if (lo != K - 1) { i = lo; ok = 1; }
        } else if (lo < K && lo > 0) {
            i = lo - 1;
            ok = 1;
        }
        if (ok) {
            double nt = (x - t[i]) / (t[i + 1] - t[i]);
            if (interp[i] == 0) {
                *out = v[i] * (1.0 - nt) + v[i + 1] * nt; /* Interpolate::lerp */
                have = 1;
            } else if (!(i == 0 || i >= K - 2)) {
                *out = cubic_hermite(nt, t[i - 1], v[i - 1], t[i], v[i], t[i + 1], v[i + 1],
                                     t[i + 2], v[i + 2]);
                have = 1;
            }
        }
    }
    if (have) return 1;
    if (x <= t[0]) { *out = v[0]; return 1; }
    if (x >= t[K - 1]) { *out = v[K - 1]; return 1; }
    return 0;
}

/* polynomial.rs:342-373 */
static void poly_polynomial_to_data(const poly_t *p, size_t n, double *out)
{
    size_t *pos = (size_t *)malloc((n + 2) * sizeof(size_t));
    size_t np = poly_positions(p, n, pos);
    size_t K = np < p->npoints ? np : p->npoints; /* zip() */
    double *t = (double *)malloc((K ? K : 1) * sizeof(double));
    uint8_t *interp = (uint8_t *)malloc(K ? K : 1);
    for (size_t k = 0; k < K; k++) {
        t[k] = (double)pos[k];
        interp[k] = (k > 0 && np - k > 2) ? 1 : 0;
    }
    double prev = p->min;
    for (size_t xi = 0; xi < n; xi++) {
        double sv;
        if (!spline_clamped_sample(t, p->points, interp, K, (double)xi, &sv)) sv = prev;
        prev = sv;
        out[xi] = orc_round_and_limit_f64(sv, p->min, p->max, 5);
    }
    free(pos);
    free(t);
    free(interp);
}

/* polynomial.rs:375-393 with inverse_distance_weight 0.1.1 [3P] */
static void poly_idw_to_data(const poly_t *p, size_t n, double *out)
{
    size_t *pos = (size_t *)malloc((n + 2) * sizeof(size_t));
    size_t np = poly_positions(p, n, pos);
    size_t K = np < p->npoints ? np : p->npoints;
    for (size_t xi = 0; xi < n; xi++) {
        double x = (double)xi, num = 0.0, den = 0.0, val = 0.0;
        int hit = 0;
        for (size_t k = 0; k < K; k++) {
            double d = fabs((double)pos[k] - x);
            if (d == 0.0) { val = p->points[k]; hit = 1; break; }
            double w = 1.0 / (d * d);
            num += w * p->points[k];
            den += w;
        }
        if (!hit) val = num / den;
        out[xi] = orc_round_and_limit_f64(val, p->min, p->max, 5);
    }
    free(pos);
}

/* polynomial.rs:395-404 */
static void poly_to_data(const poly_t *p, size_t n, double *out)
{
    if (p->max == p->min) {
        for (size_t i = 0; i < n; i++) out[i] = p->max;
        return;
    }
    if (p->idw) poly_idw_to_data(p, n, out);
    else poly_polynomial_to_data(p, n, out);
}

/* polynomial.rs:209-277 */
static int poly_compress_bounded(poly_t *p, const double *x, size_t n, double max_err)
{
    int iterations = 0;
    if (p->max == p->min) return 0;
    size_t baseline = (3 >= n / 100) ? 3 : n / 100;
    double current_err = max_err + 1.0;
    size_t jump = 0;
    double target = orc_round_f64(max_err, 3);
    double *outd = (double *)malloc(n * sizeof(double));
    while (target < orc_round_f64(current_err, 4)) {
        iterations++;
        poly_compress_hinted(p, x, n, baseline + jump);
        if (p->idw) poly_idw_to_data(p, n, outd);
        else poly_polynomial_to_data(p, n, outd);
        current_err = orc_error_mape(x, outd, n);
        if (iterations >= 1 && iterations <= 17) {
            size_t j = n / 10;
            jump += j > 1 ? j : 1;
        } else if (iterations >= 18 && iterations <= 22) {
            size_t j = n / 100;
            jump += j > 1 ? j : 1;
        } else if (target > orc_round_f64(current_err, 4)) {
            break;
        } else {
            poly_compress_hinted(p, x, n, n);
            current_err = 0.0;
            break;
        }
        if (p->npoints == n) {
            current_err = 0.0;
            break;
        }
    }
    free(outd);
    p->error = current_err;
    p->has_error = 1;
    return iterations;
}

/* polynomial.rs:54-87 */
static void poly_encode(const poly_t *p, orc_buf *b)
{
    buf_varint(b, (uint64_t)p->idw);
    buf_varint(b, (uint64_t)p->bitdepth);
    buf_varint(b, p->npoints);
    for (size_t i = 0; i < p->npoints; i++) {
        double f = p->points[i];
        switch (p->bitdepth) {
        case ORC_BD_U8: buf_u8(b, sat_u8(f)); break;
        case ORC_BD_I16: buf_zigzag(b, sat_i16(f)); break;
        case ORC_BD_I32: buf_zigzag(b, sat_i32(f)); break;
        default: buf_f64(b, f); break;
        }
    }
    buf_f64(b, p->min);
    buf_f64(b, p->max);
    buf_u8(b, p->point_step);
}

/* polynomial.rs:89-131 */
static int poly_decode(const uint8_t *d, size_t len, poly_t *p)
{
    rd r = {d, len, 0, 0};
    uint64_t id = rd_varint(&r);
    uint64_t bd = rd_varint(&r);
    uint64_t cnt = rd_varint(&r);
    if (r.err || id > 1 || bd > 3 || cnt > len) return -2;
    p->idw = (int)id;
    p->bitdepth = (int)bd;
    p->points = (double *)malloc((cnt ? cnt : 1) * sizeof(double));
    p->npoints = cnt;
    for (uint64_t i = 0; i < cnt; i++) {
        switch (bd) {
        case ORC_BD_U8: p->points[i] = (double)rd_u8(&r); break;
        case ORC_BD_I16: p->points[i] = (double)(int16_t)rd_zigzag(&r); break;
        case ORC_BD_I32: p->points[i] = (double)(int32_t)rd_zigzag(&r); break;
        default: p->points[i] = rd_f64(&r); break;
        }
    }
    p->min = rd_f64(&r);
    p->max = rd_f64(&r);
    p->point_step = rd_u8(&r);
    p->has_error = 0;
    p->error = 0.0;
    if (r.err) { free(p->points); p->points = NULL; return -2; }
    return 0;
}

static void poly_new(poly_t *p, const orc_stats *st, int idw)
{
    p->idw = idw;
    p->points = NULL;
    p->npoints = 0;
    p->min = st->min;
    p->max = st->max;
    p->point_step = 1;
    p->bitdepth = st->bitdepth;
    p->error = 0.0;
    p->has_error = 0;
}

/* polynomial.rs:407-413 (compress :307-314) */
int orc_polynomial(const double *x, size_t n, int idw, orc_buf *out)
{
    orc_stats st;
    poly_t p;
    buf_init(out);
    orc_stats_new(x, n, &st);
    poly_new(&p, &st, idw);
    size_t pts = (3 >= n / 100) ? 3 : n / 100;
    poly_compress_hinted(&p, x, n, pts);
    poly_encode(&p, out);
    free(p.points);
    return 0;
}
/* polynomial.rs:415-425 */
int orc_polynomial_allowed_error(const double *x, size_t n, double max_err, int idw, orc_buf *out,
                                 double *err, int *iterations)
{
    orc_stats st;
    poly_t p;
    buf_init(out);
    orc_stats_new(x, n, &st);
    poly_new(&p, &st, idw);
    int it = poly_compress_bounded(&p, x, n, max_err);
    poly_encode(&p, out);
    if (err) *err = p.has_error ? p.error : 0.0;
    if (iterations) *iterations = it;
    free(p.points);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* fft.rs                                                                   */
/* ------------------------------------------------------------------------ */

typedef struct {
    float re, im;
} c32;

/* rustfft 6.2.0 FftPlanner<f32>::plan_fft_{forward,inverse}(n).process() [3P]:
 * in-place unnormalised DFT, sign -1 forward / +1 inverse, any n.  Restated
 * as a recursive decimation-in-time mixed-radix transform over the prime
 * factors of n (radix-p butterflies are direct sums), complex f32 arithmetic,
 * twiddles computed in f64 and rounded to f32 as rustfft's
 * twiddles::compute_twiddle does.  Last-ulp results are not rustfft's (its
 * butterfly order depends on the CPU's SIMD level); the KATs at
 * fft.rs:551-579 pin this to f32 rounding. */
static size_t smallest_factor(size_t n)
{
    if (n % 2 == 0) return 2;
    for (size_t p = 3; p * p <= n; p += 2)
        if (n % p == 0) return p;
    return n;
}
static void fft_rec(const c32 *in, size_t istride, c32 *out, size_t n, const c32 *tw, size_t N,
                    c32 *scratch)
{
    if (n == 1) { out[0] = in[0]; return; }
    size_t p = smallest_factor(n);
    size_t m = n / p;
    /* p sub-transforms of length m over the decimated inputs */
    for (size_t r = 0; r < p; r++)
        fft_rec(in + r * istride, istride * p, out + r * m, m, tw, N, scratch);
    /* combine: X[k + m q] = sum_r W_n^{r(k + m q)} Y_r[k] */
    size_t tws = N / n;
    for (size_t k = 0; k < m; k++) {
        for (size_t r = 0; r < p; r++) {
            c32 y = out[r * m + k];
            c32 w = tw[(r * k * tws) % N];
            scratch[r].re = y.re * w.re - y.im * w.im;
            scratch[r].im = y.re * w.im + y.im * w.re;
        }
        for (size_t q = 0; q < p; q++) {
            float sr = scratch[0].re, si = scratch[0].im;
            for (size_t r = 1; r < p; r++) {
                c32 w = tw[((r * q) % p) * (N / p)];
                sr += scratch[r].re * w.re - scratch[r].im * w.im;
                si += scratch[r].re * w.im + scratch[r].im * w.re;
            }
            scratch[p + q].re = sr;
            scratch[p + q].im = si;
        }
        for (size_t q = 0; q < p; q++) out[k + m * q] = scratch[p + q];
    }
}
static void fft_process(c32 *buf, size_t n, int inverse)
{
    if (n <= 1) return;
    c32 *tw = (c32 *)malloc(n * sizeof(c32));
    double constant = (inverse ? 2.0 : -2.0) * 3.14159265358979323846 / (double)n;
    for (size_t i = 0; i < n; i++) {
        double a = constant * (double)i;
        tw[i].re = (float)cos(a);
        tw[i].im = (float)sin(a);
    }
    c32 *tmp = (c32 *)malloc(n * sizeof(c32));
    c32 *scratch = (c32 *)malloc(2 * n * sizeof(c32));
    fft_rec(buf, 1, tmp, n, tw, n, scratch);
    memcpy(buf, tmp, n * sizeof(c32));
    free(tw);
    free(tmp);
    free(scratch);
}

typedef struct {
    uint16_t pos;
    float re, im;
} fpoint;

typedef struct {
    fpoint *freqs;
    size_t nfreqs;
    float max_value, min_value;
    double error;
    int has_error;
} fft_t;

/* fft.rs:88-106 Ord for FrequencyPoint: by f32 norm = hypot(re, im) */
static int fp_cmp(const fpoint *a, const fpoint *b)
{
    float n1 = hypotf(a->re, a->im), n2 = hypotf(b->re, b->im);
    if (n1 == n2) return 0;
    if (n1 > n2) return 1;
    return -1;
}
#define FP_LE(a, b) (fp_cmp((a), (b)) <= 0)
#define FP_GE(a, b) (fp_cmp((a), (b)) >= 0)
#define FP_LT(a, b) (fp_cmp((a), (b)) < 0)

/* std::collections::BinaryHeap (Rust 1.81, rust-toolchain.toml:2) restated so
 * that equal-norm bins pop in the reference's order. */
static void heap_sift_down_range(fpoint *d, size_t pos, size_t end)
{
    fpoint elt = d[pos];
    size_t child = 2 * pos + 1;
    while (child <= (end >= 2 ? end - 2 : 0) && end >= 2) {
        child += FP_LE(&d[child], &d[child + 1]) ? 1 : 0;
        if (FP_GE(&elt, &d[child])) { d[pos] = elt; return; }
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1 && FP_LT(&elt, &d[child])) {
        d[pos] = d[child];
        pos = child;
    }
    d[pos] = elt;
}
static void heap_rebuild(fpoint *d, size_t len)
{
    size_t n = len / 2;
    while (n > 0) {
        n--;
        heap_sift_down_range(d, n, len);
    }
}
static void heap_sift_down_to_bottom(fpoint *d, size_t end)
{
    size_t pos = 0, start = 0;
    fpoint elt = d[pos];
    size_t child = 1;
    while (end >= 2 && child <= end - 2) {
        child += FP_LE(&d[child], &d[child + 1]) ? 1 : 0;
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1) {
        d[pos] = d[child];
        pos = child;
    }
    /* sift_up(start, pos) */
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (FP_LE(&elt, &d[parent])) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
}
static int heap_pop(fpoint *d, size_t *len, fpoint *out)
{
    if (*len == 0) return 0;
    fpoint item = d[--(*len)];
    if (*len > 0) {
        fpoint top = d[0];
        d[0] = item;
        item = top;
        heap_sift_down_to_bottom(d, *len);
    }
    *out = item;
    return 1;
}

/* Test support: the pop order of BinaryHeap::from(bins in position order) for bins with the given f32
 * norms (fft.rs:245-255 as fft_trim uses it) -- what the GPU's heap replay is held against.  Each bin is
 * the point (norm, 0), whose hypotf is the norm itself; n <= 65536 (pos is a u16, fft.rs:36). */
int orc_heap_order(const float *norms, size_t n, size_t k, uint32_t *out)
{
    if (!norms || !out || k > n || n > 65536) return -1;
    fpoint *d = (fpoint *)malloc((n ? n : 1) * sizeof(fpoint));
    if (!d) return -2;
    for (size_t i = 0; i < n; i++) { d[i].pos = (uint16_t)i; d[i].re = norms[i]; d[i].im = 0.0f; }
    size_t len = n;
    heap_rebuild(d, len);
    for (size_t i = 0; i < k; i++) {
        fpoint it;
        if (!heap_pop(d, &len, &it)) break;
        out[i] = it.pos;
    }
    free(d);
    return 0;
}

/* fft.rs:231-257 */
static void fft_trim(fft_t *f, const c32 *half, size_t nbins, size_t max_freq)
{
    free(f->freqs);
    f->freqs = (fpoint *)malloc((max_freq ? max_freq : 1) * sizeof(fpoint));
    f->nfreqs = 0;
    if (max_freq == 1) {
        f->freqs[0].pos = 0;
        f->freqs[0].re = half[0].re;
        f->freqs[0].im = half[0].im;
        f->nfreqs = 1;
        return;
    }
    fpoint *heap = (fpoint *)malloc((nbins ? nbins : 1) * sizeof(fpoint));
    for (size_t i = 0; i < nbins; i++) {
        heap[i].pos = (uint16_t)i; /* `pos as u16` wraps */
        heap[i].re = half[i].re;
        heap[i].im = half[i].im;
    }
    size_t hl = nbins;
    heap_rebuild(heap, hl);
    for (size_t i = 0; i < max_freq; i++) {
        fpoint item;
        if (heap_pop(heap, &hl, &item)) {
            if (item.im == 0.0f && item.re == 0.0f) break;
            f->freqs[f->nfreqs++] = item;
        }
    }
    free(heap);
}

/* fft.rs:173-180 */
static float f64_to_f32(double x) { return (float)x; }

/* fft.rs:162-171 */
static void fft_new(fft_t *f, double mn, double mx)
{
    f->freqs = NULL;
    f->nfreqs = 0;
    f->max_value = f64_to_f32(mx);
    f->min_value = f64_to_f32(mn);
    f->error = 0.0;
    f->has_error = 0;
}

/* fft.rs:208-218 : max is checked first */
static double fft_round(const fft_t *f, float x, uint32_t decimals)
{
    double y = pow10i(decimals);
    double out = round((double)x * y) / y;
    if (out > (double)f->max_value) return (double)f->max_value;
    if (out < (double)f->min_value) return (double)f->min_value;
    return out;
}

/* fft.rs:184-204 */
size_t orc_gibbs_sizing(const double *x, size_t n, double *out)
{
    size_t ns = orc_next_size(n);
    size_t added = ns - n;
    size_t pre = added / 2, suf = added - pre;
    size_t k = 0;
    if (n == 0) return 0;
    for (size_t i = 0; i < pre; i++) out[k++] = x[0];
    for (size_t i = 0; i < n; i++) out[k++] = x[i];
    for (size_t i = 0; i < suf; i++) out[k++] = x[n - 1];
    return k;
}

/* fft.rs:401-422 */
static c32 *fft_mirrored(const fft_t *f, size_t len)
{
    c32 *d = (c32 *)calloc(len ? len : 1, sizeof(c32));
    for (size_t i = 0; i < f->nfreqs; i++) {
        size_t pos = f->freqs[i].pos;
        if (pos >= len) continue; /* the reference would panic on a corrupt stream */
        d[pos].re = f->freqs[i].re;
        d[pos].im = f->freqs[i].im;
        if (pos == 0) continue;
        d[len - pos].re = f->freqs[i].re;
        d[len - pos].im = f->freqs[i].im * -1.0f;
    }
    return d;
}

/* fft.rs:262-282 (compress_hinted) and :366-388 (compress): no Gibbs padding here */
static void fft_compress_hinted(fft_t *f, const double *x, size_t n, size_t max_freq)
{
    if (f->max_value == f->min_value) return;
    c32 *buf = (c32 *)malloc(n * sizeof(c32));
    for (size_t i = 0; i < n; i++) { buf[i].re = f64_to_f32(x[i]); buf[i].im = 0.0f; }
    fft_process(buf, n, 0);
    fft_trim(f, buf, n / 2 + 1, max_freq);
    free(buf);
}

/* fft.rs:288-362 */
static int fft_compress_bounded(fft_t *f, const double *x, size_t n, double max_err)
{
    int iterations = 0;
    if (f->max_value == f->min_value) return 0;
    size_t max_freq = (3 >= n / 100) ? 3 : n / 100;
    double *g;
    size_t len;
    if (n >= 128) {
        g = (double *)malloc(orc_next_size(n) * sizeof(double));
        len = orc_gibbs_sizing(x, n, g);
    } else {
        g = (double *)malloc(n * sizeof(double));
        memcpy(g, x, n * sizeof(double));
        len = n;
    }
    float len_f32 = (float)len;
    c32 *buf = (c32 *)malloc(len * sizeof(c32));
    for (size_t i = 0; i < len; i++) { buf[i].re = f64_to_f32(g[i]); buf[i].im = 0.0f; }
    fft_process(buf, len, 0);
    size_t nbins = len / 2 + 1;
    double current_err = max_err + 1.0;
    size_t jump = 0;
    double *outd = (double *)malloc(len * sizeof(double));
    while (sat_i32(max_err * 1000.0) < sat_i32(current_err * 1000.0)) {
        iterations++;
        fft_trim(f, buf, nbins, max_freq + jump);
        c32 *idata = fft_mirrored(f, len);
        fft_process(idata, len, 1);
        for (size_t i = 0; i < len; i++) outd[i] = fft_round(f, idata[i].re / len_f32, 5);
        free(idata);
        current_err = orc_error_mape(g, outd, len);
        if (iterations >= 1 && iterations <= 17) {
            size_t j = max_freq / 2;
            jump += j > 1 ? j : 1;
        } else if (iterations >= 18 && iterations <= 22) {
            size_t j = max_freq / 10;
            jump += j > 1 ? j : 1;
        } else {
            break;
        }
    }
    f->error = current_err;
    f->has_error = 1;
    free(outd);
    free(buf);
    free(g);
    return iterations;
}

/* fft.rs:119-130 */
static void fft_encode(const fft_t *f, orc_buf *b)
{
    buf_u8(b, 15);
    buf_varint(b, f->nfreqs);
    for (size_t i = 0; i < f->nfreqs; i++) {
        buf_varint(b, f->freqs[i].pos);
        buf_f32(b, f->freqs[i].re);
        buf_f32(b, f->freqs[i].im);
    }
    buf_f32(b, f->max_value);
    buf_f32(b, f->min_value);
}
/* fft.rs:132-144 */
static int fft_decode(const uint8_t *d, size_t len, fft_t *f)
{
    rd r = {d, len, 0, 0};
    (void)rd_u8(&r);
    uint64_t cnt = rd_varint(&r);
    if (r.err || cnt > len) return -2;
    f->freqs = (fpoint *)malloc((cnt ? cnt : 1) * sizeof(fpoint));
    f->nfreqs = cnt;
    for (uint64_t i = 0; i < cnt; i++) {
        f->freqs[i].pos = (uint16_t)rd_varint(&r);
        f->freqs[i].re = rd_f32(&r);
        f->freqs[i].im = rd_f32(&r);
    }
    f->max_value = rd_f32(&r);
    f->min_value = rd_f32(&r);
    f->has_error = 0;
    f->error = 0.0;
    if (r.err) { free(f->freqs); f->freqs = NULL; return -2; }
    return 0;
}
/* fft.rs:426-462 */
static void fft_to_data(const fft_t *f, size_t n, double *out)
{
    if (f->max_value == f->min_value) {
        for (size_t i = 0; i < n; i++) out[i] = (double)f->max_value;
        return;
    }
    size_t pre = 0, suf = 0;
    if (n >= 128) {
        size_t added = orc_next_size(n) - n;
        pre = added / 2;
        suf = added - pre;
    }
    size_t gl = n + pre + suf;
    c32 *d = fft_mirrored(f, gl);
    fft_process(d, gl, 1);
    float len = (float)gl;
    for (size_t i = 0; i < n; i++) out[i] = fft_round(f, d[pre + i].re / len, 5);
    free(d);
}

static void minmax_scan(const double *x, size_t n, double *mn, double *mx)
{
    /* fft.rs:468-477 (strict comparisons, start from data[0]) */
    double a = x[0], b = x[0];
    for (size_t i = 0; i < n; i++) {
        if (x[i] > b) b = x[i];
        if (x[i] < a) a = x[i];
    }
    *mn = a;
    *mx = b;
}
/* fft.rs:466-484 */
int orc_fft(const double *x, size_t n, orc_buf *out)
{
    double mn, mx;
    fft_t f;
    buf_init(out);
    minmax_scan(x, n, &mn, &mx);
    fft_new(&f, mn, mx);
    size_t max_freq = (3 >= n / 100) ? 3 : n / 100;
    fft_compress_hinted(&f, x, n, max_freq);
    fft_encode(&f, out);
    free(f.freqs);
    return 0;
}
/* fft.rs:526-544 */
int orc_fft_set(const double *x, size_t n, size_t freqs, orc_buf *out)
{
    double mn, mx;
    fft_t f;
    buf_init(out);
    minmax_scan(x, n, &mn, &mx);
    fft_new(&f, mn, mx);
    fft_compress_hinted(&f, x, n, freqs);
    fft_encode(&f, out);
    free(f.freqs);
    return 0;
}
/* fft.rs:494-512 and :516-524 (stats.min/max equal the scan above) */
int orc_fft_allowed_error(const double *x, size_t n, double max_err, orc_buf *out, double *err,
                          int *iterations)
{
    double mn, mx;
    fft_t f;
    buf_init(out);
    minmax_scan(x, n, &mn, &mx);
    fft_new(&f, mn, mx);
    int it = fft_compress_bounded(&f, x, n, max_err);
    fft_encode(&f, out);
    if (err) *err = f.has_error ? f.error : 0.0;
    if (iterations) *iterations = it;
    free(f.freqs);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* compressor/mod.rs dispatch                                               */
/* ------------------------------------------------------------------------ */

int orc_compress(int compressor, const double *x, size_t n, int bounded, double max_err,
                 orc_buf *out, double *err)
{
    double e = 0.0;
    int rc = 0;
    if (n == 0) return -1;
    switch (compressor) {
    case ORC_NOOP: rc = orc_noop(x, n, out); break;
    case ORC_CONSTANT: rc = orc_constant(x, n, out); break;
    case ORC_RLE: rc = orc_rle(x, n, out); break;
    case ORC_FFT:
        rc = bounded ? orc_fft_allowed_error(x, n, max_err, out, &e, NULL) : orc_fft(x, n, out);
        break;
    case ORC_POLYNOMIAL:
        rc = bounded ? orc_polynomial_allowed_error(x, n, max_err, 0, out, &e, NULL)
                     : orc_polynomial(x, n, 0, out);
        break;
    case ORC_IDW:
        rc = bounded ? orc_polynomial_allowed_error(x, n, max_err, 1, out, &e, NULL)
                     : orc_polynomial(x, n, 1, out);
        break;
    default: return -3; /* Compressor::Auto => todo!() (mod.rs:72,90,105) */
    }
    if (err) *err = e;
    return rc;
}

/* compressor/mod.rs:109-119 */
int orc_decompress(int compressor, const uint8_t *data, size_t len, size_t samples, double **out,
                   size_t *out_n)
{
    if (compressor == ORC_NOOP) return noop_to_data(data, len, out, out_n);
    double *o = (double *)malloc((samples ? samples : 1) * sizeof(double));
    int rc = 0;
    switch (compressor) {
    case ORC_CONSTANT: rc = constant_to_data(data, len, samples, o); break;
    case ORC_RLE: rc = rle_to_data(data, len, samples, o); break;
    case ORC_FFT: {
        fft_t f;
        rc = fft_decode(data, len, &f);
        if (!rc) { fft_to_data(&f, samples, o); free(f.freqs); }
        break;
    }
    case ORC_POLYNOMIAL:
    case ORC_IDW: {
        poly_t p;
        rc = poly_decode(data, len, &p);
        if (!rc) {
            if (p.point_step == 0 && p.max != p.min) rc = -2; /* step_by(0) panics */
            else poly_to_data(&p, samples, o);
            free(p.points);
        }
        break;
    }
    default: rc = -3;
    }
    if (rc) { free(o); return rc; }
    *out = o;
    *out_n = samples;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* frame/mod.rs : compress_best                                             */
/* ------------------------------------------------------------------------ */

static const int32_t COMPRESSION_SPEED[7] = {INT32_MAX, 4096, 2048, 1024, 512, 256, 128};

/* frame/mod.rs:71-149 */
int orc_compress_best(const double *x, size_t n, float max_error, int level, orc_buf *out,
                      int *chosen, double *err)
{
    static const int list[3] = {ORC_FFT, ORC_POLYNOMIAL, ORC_RLE};
    if (n == 0 || level < 0 || level > 6) return -1;
    size_t data_sample = (size_t)COMPRESSION_SPEED[level];
    double me = (double)max_error;
    orc_stats st;
    orc_stats_new(x, n, &st);
    buf_init(out);
    if (st.min == st.max) {
        *chosen = ORC_CONSTANT;
        if (err) *err = 0.0;
        return orc_compress(ORC_CONSTANT, x, n, 1, me, out, NULL);
    }
    if (n >= data_sample) {
        int best = -1;
        size_t best_len = 0;
        for (int c = 0; c < 3; c++) {
            orc_buf b;
            double e;
            orc_compress(list[c], x, data_sample, 1, me, &b, &e);
            if (e <= me && (best < 0 || b.len < best_len)) { best = list[c]; best_len = b.len; }
            free(b.ptr);
        }
        if (best < 0) return -4; /* .unwrap() on None */
        *chosen = best;
        return orc_compress(best, x, n, 1, me, out, err);
    }
    orc_buf res[3];
    double errs[3];
    int all_fail = 1;
    for (int c = 0; c < 3; c++) {
        orc_compress(list[c], x, n, 1, me, &res[c], &errs[c]);
        if (errs[c] <= me) all_fail = 0;
    }
    int best = -1;
    for (int c = 0; c < 3; c++) {
        if (!all_fail && !(errs[c] <= me)) continue;
        if (best < 0 || res[c].len < res[best].len) best = c;
    }
    *out = res[best];
    *chosen = list[best];
    if (err) *err = errs[best];
    for (int c = 0; c < 3; c++)
        if (c != best) free(res[c].ptr);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* data.rs + header.rs + main.rs                                            */
/* ------------------------------------------------------------------------ */

int orc_stream_compress(const double *x, const uint64_t *chunk_off, size_t n_chunks,
                        int compressor, int bounded, float max_error, int level, orc_buf *out,
                        uint8_t *chosen, double *errs)
{
    buf_init(out);
    /* header.rs:60-67 ; frame_count is a u8 that wraps (header.rs:52-54, release build) */
    buf_bytes(out, "BRRO", 4);
    uint32_t ver = 1;
    buf_bytes(out, &ver, 4);
    buf_u8(out, (uint8_t)(n_chunks & 0xff));
    /* data.rs:79-85 : bincode Vec<CompressorFrame> */
    buf_varint(out, n_chunks);
    for (size_t c = 0; c < n_chunks; c++) {
        const double *cx = x + chunk_off[c];
        size_t cn = (size_t)(chunk_off[c + 1] - chunk_off[c]);
        orc_buf pl;
        int tag = compressor;
        double e = 0.0;
        int rc;
        if (bounded && compressor == ORC_AUTO)
            rc = orc_compress_best(cx, cn, max_error, level, &pl, &tag, &e); /* data.rs:70 */
        else
            rc = orc_compress(compressor, cx, cn, bounded, (double)max_error, &pl, &e);
        if (rc) { free(out->ptr); buf_init(out); return rc; }
        /* frame/mod.rs:25-33 derive(Encode) order; frame_size = 41 (frame/mod.rs:50-56) */
        buf_varint(out, 41);
        buf_varint(out, cn);
        buf_varint(out, (uint64_t)tag);
        buf_varint(out, pl.len);
        buf_bytes(out, pl.ptr, pl.len);
        free(pl.ptr);
        if (chosen) chosen[c] = (uint8_t)tag;
        if (errs) errs[c] = e;
    }
    return 0;
}

/* main.rs:130-165 */
int orc_compress_data(const double *x, size_t n, int compressor, uint8_t cli_error, int level,
                      orc_buf *out)
{
    double *clean = (double *)malloc((n ? n : 1) * sizeof(double));
    size_t cn = orc_clean_data(x, n, clean);
    size_t nch = orc_chunk_sizes(cn, NULL, 0);
    size_t *sizes = (size_t *)malloc((nch ? nch : 1) * sizeof(size_t));
    orc_chunk_sizes(cn, sizes, nch);
    uint64_t *off = (uint64_t *)malloc((nch + 1) * sizeof(uint64_t));
    off[0] = 0;
    for (size_t i = 0; i < nch; i++) off[i + 1] = off[i] + sizes[i];
    int lossy = (compressor == ORC_FFT || compressor == ORC_POLYNOMIAL || compressor == ORC_IDW ||
                 compressor == ORC_AUTO);
    float max_error = (float)cli_error / 100.0f; /* main.rs:157 */
    int rc = orc_stream_compress(clean, off, nch, compressor, lossy, max_error, level, out, NULL,
                                 NULL);
    free(clean);
    free(sizes);
    free(off);
    return rc;
}

/* main.rs:168-172 ; data.rs:89-109 ; header.rs:69-84 */
int orc_decompress_data(const uint8_t *bro, size_t len, double **out, size_t *out_n)
{
    if (len < 9) return -2;
    if (memcmp(bro, "BRRO", 4) != 0) return -5; /* panic!("Magic bytes are not correct!") */
    uint32_t ver;
    memcpy(&ver, bro + 4, 4);
    if (ver > 1) return -6; /* "is higher than compressor version" */
    rd r = {bro + 9, len - 9, 0, 0};
    uint64_t nf = rd_varint(&r);
    size_t cap = 1024, tot = 0;
    double *o = (double *)malloc(cap * sizeof(double));
    for (uint64_t f = 0; f < nf; f++) {
        (void)rd_varint(&r); /* frame_size */
        uint64_t samples = rd_varint(&r);
        uint64_t tag = rd_varint(&r);
        uint64_t dl = rd_varint(&r);
        if (r.err || r.pos + dl > r.len) { free(o); return -2; }
        double *fo;
        size_t fn;
        int rc = orc_decompress((int)tag, r.p + r.pos, dl, samples, &fo, &fn);
        if (rc) { free(o); return rc; }
        r.pos += dl;
        if (tot + fn > cap) {
            while (tot + fn > cap) cap *= 2;
            o = (double *)realloc(o, cap * sizeof(double));
        }
        memcpy(o + tot, fo, fn * sizeof(double));
        tot += fn;
        free(fo);
    }
    *out = o;
    *out_n = tot;
    return 0;
}
