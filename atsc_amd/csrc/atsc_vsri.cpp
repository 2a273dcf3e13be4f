// Host side of the csv-compressor front end: the VSRI timestamp index
// (vsri/src/lib.rs), the `timestamp,value` sample files (csv-compressor/src/csv.rs) and the
// Metric glue between them (csv-compressor/src/metric.rs).  No kernels: integer bookkeeping and
// text I/O around the GPU compressor, behind the same C ABI.
//
// Arithmetic is Rust's i32 in a release build: wrapping add / sub / mul, division truncating toward
// zero; where the reference panics (division by zero on a one-point segment, i32::MIN / -1, unwrap
// of a missing value) the entry points return an error code instead.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/atsc_hip.h"
#include "atsc_internal.h"

namespace atsc_text {
bool parse_rust_f64(const std::string &s, double &v);                     // atsc_stream.cpp
void split_csv(const std::string &line, std::vector<std::string> &f);     // atsc_stream.cpp
int read_lines(const char *path, std::vector<std::string> &lines);        // atsc_stream.cpp
}  // namespace atsc_text

struct atsc_vsri {
    int32_t min_ts = 0, max_ts = 0;
    struct Seg { int32_t v[4]; };  // [sample rate m, x0, y0, number of samples]  (lib.rs:105)
    std::vector<Seg> seg;
};

namespace {

inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
// i32 `/`: panics on a zero divisor and on MIN / -1
inline bool wdiv(int32_t a, int32_t b, int32_t &q)
{
    if (b == 0 || (a == INT32_MIN && b == -1)) return false;
    q = a / b;
    return true;
}

using Seg = atsc_vsri::Seg;

Seg current_segment(const atsc_vsri *v)  // lib.rs:293-298
{
    if (v->seg.empty()) return Seg{{0, 0, 0, 0}};
    return v->seg.back();
}
int32_t calculate_b(const Seg &s) { return wsub(s.v[2], wmul(s.v[0], s.v[1])); }  // lib.rs:286-290
int32_t sample_count(const atsc_vsri *v)  // lib.rs:355-358
{
    const Seg l = current_segment(v);
    return wadd(l.v[3], l.v[1]);
}
int32_t seg_end(const Seg &s) { return wadd(s.v[2], wmul(s.v[0], wsub(s.v[3], 1))); }

// lib.rs:301-317.  1 = Some, 0 = None, ATSC_E_INVALID = the reference's division panic
int get_sample(const atsc_vsri *v, int32_t y, int32_t *out)
{
    for (const Seg &s : v->seg) {
        if (y >= s.v[2] && y <= seg_end(s)) {
            int32_t x;
            if (!wdiv(wsub(y, calculate_b(s)), s.v[0], x)) return ATSC_E_INVALID;
            *out = x;
            return 1;
        }
    }
    return 0;
}
int get_next_sample(const atsc_vsri *v, int32_t y, int32_t *out)  // lib.rs:154-169
{
    if (y < v->min_ts) { *out = 0; return 1; }
    if (y >= v->max_ts) return 0;
    for (size_t i = v->seg.size(); i-- > 0;) {
        if (y <= v->seg[i].v[2]) { *out = v->seg[i].v[1]; return 1; }
    }
    return 0;
}
int get_previous_sample(const atsc_vsri *v, int32_t y, int32_t *out)  // lib.rs:175-193
{
    if (y < v->min_ts) return 0;
    if (y >= v->max_ts) { *out = sample_count(v); return 1; }
    for (const Seg &s : v->seg) {
        if (y < s.v[2]) { *out = wsub(s.v[1], 1); return 1; }
    }
    return 0;
}

bool parse_i32(const std::string &s0, int32_t &out)  // `line.trim().parse::<i32>()`
{
    size_t b = 0, e = s0.size();
    while (b < e && isspace((unsigned char)s0[b])) ++b;
    while (e > b && isspace((unsigned char)s0[e - 1])) --e;
    if (b == e) return false;
    size_t i = b;
    bool neg = false;
    if (s0[i] == '+' || s0[i] == '-') { neg = s0[i] == '-'; ++i; }
    if (i == e) return false;
    int64_t v = 0;
    for (; i < e; ++i) {
        if (!isdigit((unsigned char)s0[i])) return false;
        v = v * 10 + (s0[i] - '0');
        if (v > 2147483648ll) return false;
    }
    v = neg ? -v : v;
    if (v > INT32_MAX || v < INT32_MIN) return false;
    out = (int32_t)v;
    return true;
}
bool parse_i64(const std::string &s, int64_t &out)  // Rust `str::parse::<i64>`: [+-]?digits, no blanks
{
    size_t i = 0;
    const size_t e = s.size();
    if (e == 0) return false;
    bool neg = false;
    if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
    if (i == e) return false;
    unsigned __int128 v = 0;
    const unsigned __int128 lim = neg ? ((unsigned __int128)1 << 63) : (((unsigned __int128)1 << 63) - 1);
    for (; i < e; ++i) {
        if (!isdigit((unsigned char)s[i])) return false;
        v = v * 10 + (unsigned)(s[i] - '0');
        if (v > lim) return false;
    }
    out = neg ? (int64_t)(0 - (uint64_t)v) : (int64_t)v;
    return true;
}

// ryu::Buffer::format (the csv crate's f64 serialiser): shortest round-trip digits, then ryu's
// "pretty" layout -- 1.0, 12.34, 0.001234, 1e16, 1.234e-7, NaN, inf, -inf.
std::string ryu_f64(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), std::fabs(v), std::chars_format::scientific);
    std::string sci(buf, r.ptr);  // d[.ddd]e[+-]XX
    const size_t epos = sci.find('e');
    std::string digits;
    for (size_t i = 0; i < epos; ++i)
        if (sci[i] != '.') digits.push_back(sci[i]);
    const int exp10 = atoi(sci.c_str() + epos + 1);
    const int len = (int)digits.size();
    const int k = exp10 - (len - 1);  // value = digits * 10^k
    const int kk = len + k;           // position of the decimal point
    std::string out = std::signbit(v) ? "-" : "";
    if (0 <= k && kk <= 16) {         // 1234e7 -> 12340000000.0
        out += digits + std::string((size_t)k, '0') + ".0";
    } else if (0 < kk && kk <= 16) {  // 1234e-2 -> 12.34
        out += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (-5 < kk && kk <= 0) {  // 1234e-6 -> 0.001234
        out += "0." + std::string((size_t)(-kk), '0') + digits;
    } else if (len == 1) {            // 1e30
        out += digits + "e" + std::to_string(kk - 1);
    } else {                          // 1234e30 -> 1.234e33
        out += digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    }
    return out;
}

// csv::Writer quoting (QuoteStyle::Necessary): fields of digits, signs, dots and letters never need it
}  // namespace

extern "C" atsc_vsri *atsc_vsri_new(void) { return new (std::nothrow) atsc_vsri(); }
extern "C" void atsc_vsri_free(atsc_vsri *v) { delete v; }
extern "C" int32_t atsc_vsri_min(const atsc_vsri *v) { return v ? v->min_ts : 0; }
extern "C" int32_t atsc_vsri_max(const atsc_vsri *v) { return v ? v->max_ts : 0; }
extern "C" uint64_t atsc_vsri_segment_count(const atsc_vsri *v) { return v ? v->seg.size() : 0; }
extern "C" int atsc_vsri_segment(const atsc_vsri *v, uint64_t i, int32_t out[4])
{
    if (!v || !out || i >= v->seg.size()) return ATSC_E_INVALID;
    memcpy(out, v->seg[i].v, sizeof(int32_t) * 4);
    return ATSC_OK;
}
extern "C" int32_t atsc_vsri_get_sample_count(const atsc_vsri *v) { return v ? sample_count(v) : 0; }

// Vsri::update_for_point (lib.rs:236-273)
extern "C" int atsc_vsri_update_for_point(atsc_vsri *v, int32_t y)
{
    ATSC_API_BEGIN
    if (!v) return ATSC_E_INVALID;
    if (y < v->max_ts) return ATSC_E_INVALID;  // Error::UpdateIndexForPointError: a point in the past
    v->max_ts = y;
    if (v->seg.empty()) {
        v->min_ts = y;
        v->seg.push_back(Seg{{0, 0, y, 1}});  // create_fake_segment on an empty index
        return ATSC_OK;
    }
    Seg &last = v->seg.back();
    if (last.v[0] == 0) {
        // fake segment (one point, m unknown) + a second point: generate_segment (lib.rs:362-375)
        last = Seg{{wsub(y, last.v[2]), last.v[1], last.v[2], 2}};
        return ATSC_OK;
    }
    // fits_segment (lib.rs:394-413): the point must be the next one on the line
    int32_t x;
    if (!wdiv(wsub(y, calculate_b(last)), last.v[0], x)) return ATSC_E_INVALID;
    if (x == wadd(last.v[3], last.v[1])) {
        last.v[3] = wadd(last.v[3], 1);
        return ATSC_OK;
    }
    const Seg cur = last;
    v->seg.push_back(Seg{{0, wadd(cur.v[1], cur.v[3]), y, 1}});  // create_fake_segment (lib.rs:380-386)
    return ATSC_OK;
    ATSC_API_END
}

extern "C" int atsc_vsri_get_sample(const atsc_vsri *v, int32_t y, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    return get_sample(v, y, out);
}
extern "C" int atsc_vsri_get_next_sample(const atsc_vsri *v, int32_t y, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    return get_next_sample(v, y, out);
}
extern "C" int atsc_vsri_get_previous_sample(const atsc_vsri *v, int32_t y, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    return get_previous_sample(v, y, out);
}
// lib.rs:137-141: get_sample(y).or_else(|| get_next_sample(y))
extern "C" int atsc_vsri_get_this_or_next(const atsc_vsri *v, int32_t y, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    const int r = get_sample(v, y, out);
    return r != 0 ? r : get_next_sample(v, y, out);
}
// lib.rs:144-148: get_sample(y).or(get_previous_sample(y)) -- `or` evaluates its argument eagerly,
// which changes nothing observable here
extern "C" int atsc_vsri_get_this_or_previous(const atsc_vsri *v, int32_t y, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    const int r = get_sample(v, y, out);
    return r != 0 ? r : get_previous_sample(v, y, out);
}
// Vsri::get_time (lib.rs:320-341), including `y0 + m * x` with the absolute sample number x
extern "C" int atsc_vsri_get_time(const atsc_vsri *v, int32_t x, int32_t *out)
{
    if (!v || !out) return ATSC_E_INVALID;
    if (x == 0) { *out = v->min_ts; return 1; }
    const int32_t cnt = sample_count(v);
    if (x > cnt) return 0;
    if (x == cnt) { *out = v->max_ts; return 1; }
    for (const Seg &s : v->seg) {
        if (x >= s.v[1] && x < wadd(s.v[1], s.v[3])) {
            *out = wadd(s.v[2], wmul(s.v[0], x));
            return 1;
        }
    }
    return 0;
}
// Vsri::is_empty (lib.rs:198-232)
extern "C" int atsc_vsri_is_empty(const atsc_vsri *v, int32_t t0, int32_t t1)
{
    if (!v) return ATSC_E_INVALID;
    if (v->seg.size() == 1) {
        if ((t0 >= v->min_ts && t0 <= v->max_ts) || (t1 <= v->max_ts && t1 >= v->min_ts)) return 0;
        if (t0 < v->min_ts && t1 > v->max_ts) return 0;
        return 1;
    }
    int32_t prev_end = 0;
    for (size_t i = 0; i < v->seg.size(); ++i) {
        const Seg &s = v->seg[i];
        const int32_t y0 = s.v[2], end = seg_end(s);
        if (i >= 1 && (t0 > prev_end && t1 < y0)) return 1;
        if ((t0 >= y0 && t0 < end) || (t1 < end && t1 >= y0)) return 0;
        if (t0 < y0 && t1 > end) return 0;
        prev_end = end;
    }
    return 1;
}
// Vsri::get_all_timestamps (lib.rs:344-353)
extern "C" int atsc_vsri_get_all_timestamps(const atsc_vsri *v, int32_t **out, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!v || !out || !n) return ATSC_E_INVALID;
    std::vector<int32_t> t;
    for (const Seg &s : v->seg)
        for (int32_t f = 0; f < s.v[3]; ++f) t.push_back(wadd(wmul(f, s.v[0]), s.v[2]));
    int32_t *p = (int32_t *)malloc((t.size() ? t.size() : 1) * sizeof(int32_t));
    if (!p) return ATSC_E_NOMEM;
    if (!t.empty()) memcpy(p, t.data(), t.size() * sizeof(int32_t));
    *out = p;
    *n = t.size();
    return ATSC_OK;
    ATSC_API_END
}

// Vsri::flush_to (lib.rs:424-443): min, max, then one "m,x0,y0,count" line per segment
extern "C" int atsc_vsri_flush_to(const atsc_vsri *v, const char *path)
{
    ATSC_API_BEGIN
    if (!v || !path) return ATSC_E_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return ATSC_E_IO;
    bool ok = fprintf(f, "%d\n%d\n", v->min_ts, v->max_ts) > 0;
    for (const Seg &s : v->seg) ok = ok && fprintf(f, "%d,%d,%d,%d\n", s.v[0], s.v[1], s.v[2], s.v[3]) > 0;
    ok = (fclose(f) == 0) && ok;
    return ok ? ATSC_OK : ATSC_E_IO;
    ATSC_API_END
}
// Vsri::load (lib.rs:447-486); malformed numbers or a segment without exactly 4 fields are unwrap
// panics there, ATSC_E_FORMAT here
extern "C" int atsc_vsri_load(const char *path, atsc_vsri **out)
{
    ATSC_API_BEGIN
    if (!path || !out) return ATSC_E_INVALID;
    *out = nullptr;
    std::vector<std::string> lines;
    int rc = atsc_text::read_lines(path, lines);
    if (rc) return rc;
    atsc_vsri *v = new (std::nothrow) atsc_vsri();
    if (!v) return ATSC_E_NOMEM;
    for (size_t i = 0; i < lines.size(); ++i) {
        bool ok = true;
        if (i == 0) ok = parse_i32(lines[i], v->min_ts);
        else if (i == 1) ok = parse_i32(lines[i], v->max_ts);
        else {
            Seg s;
            size_t b = 0, k = 0;
            const std::string &l = lines[i];
            for (;;) {
                const size_t c = l.find(',', b);
                const std::string fld = l.substr(b, c == std::string::npos ? std::string::npos : c - b);
                int32_t val;
                if (k >= 4 || !parse_i32(fld, val)) { ok = false; break; }
                s.v[k++] = val;
                if (c == std::string::npos) break;
                b = c + 1;
            }
            ok = ok && k == 4;
            if (ok) v->seg.push_back(s);
        }
        if (!ok) { delete v; return ATSC_E_FORMAT; }
    }
    *out = v;
    return ATSC_OK;
    ATSC_API_END
}

// vsri::day_elapsed_seconds (lib.rs:49-57): seconds since midnight UTC of a unix timestamp.
// chrono accepts years -262143..=262142; outside of that the reference's unwrap panics.
extern "C" int atsc_day_elapsed_seconds(int64_t timestamp_sec, int32_t *out)
{
    if (!out) return ATSC_E_INVALID;
    const int64_t lo = -8334601228800ll, hi = 8210266876799ll;  // chrono's DateTime<Utc> range in seconds
    if (timestamp_sec < lo || timestamp_sec > hi) return ATSC_E_INVALID;
    int64_t r = timestamp_sec % 86400;
    if (r < 0) r += 86400;
    *out = (int32_t)r;
    return ATSC_OK;
}

// csv-compressor/src/csv.rs:41-45: csv::Reader with headers, records deserialised into
// Sample { timestamp: i64, value: f64 } by header name.
extern "C" int atsc_samples_csv_read(const char *path, int64_t **ts, double **val, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!path || !ts || !val || !n) return ATSC_E_INVALID;
    *ts = nullptr;
    *val = nullptr;
    *n = 0;
    std::vector<std::string> lines;
    int rc = atsc_text::read_lines(path, lines);
    if (rc) return rc;
    std::vector<std::string> f;
    size_t li = 0;
    while (li < lines.size() && lines[li].empty()) ++li;
    std::vector<int64_t> tv;
    std::vector<double> vv;
    if (li < lines.size()) {
        atsc_text::split_csv(lines[li++], f);
        const size_t width = f.size();
        size_t ct = width, cv = width;
        for (size_t k = 0; k < width; ++k) {
            if (f[k] == "timestamp" && ct == width) ct = k;
            if (f[k] == "value" && cv == width) cv = k;
        }
        for (; li < lines.size(); ++li) {
            if (lines[li].empty()) continue;
            atsc_text::split_csv(lines[li], f);
            if (f.size() != width) return ATSC_E_FORMAT;          // UnequalLengths
            if (ct == width || cv == width) return ATSC_E_FORMAT;  // missing field `timestamp` / `value`
            int64_t t;
            double v;
            if (!parse_i64(f[ct], t) || !atsc_text::parse_rust_f64(f[cv], v)) return ATSC_E_FORMAT;
            tv.push_back(t);
            vv.push_back(v);
        }
    }
    int64_t *pt = (int64_t *)malloc((tv.size() ? tv.size() : 1) * sizeof(int64_t));
    double *pv = (double *)malloc((vv.size() ? vv.size() : 1) * sizeof(double));
    if (!pt || !pv) { free(pt); free(pv); return ATSC_E_NOMEM; }
    if (!tv.empty()) {
        memcpy(pt, tv.data(), tv.size() * sizeof(int64_t));
        memcpy(pv, vv.data(), vv.size() * sizeof(double));
    }
    *ts = pt;
    *val = pv;
    *n = tv.size();
    return ATSC_OK;
    ATSC_API_END
}

// csv-compressor/src/csv.rs:48-56: header from the struct's field names, itoa / ryu numbers, "\n"
extern "C" int atsc_samples_csv_write(const char *path, const int64_t *ts, const double *val, uint64_t n)
{
    ATSC_API_BEGIN
    if (!path || (n && (!ts || !val))) return ATSC_E_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return ATSC_E_IO;
    bool ok = true;
    // the csv writer emits the header with the first record: no records, empty file
    for (uint64_t i = 0; i < n && ok; ++i) {
        if (i == 0) ok = fputs("timestamp,value\n", f) >= 0;
        ok = ok && fprintf(f, "%lld,%s\n", (long long)ts[i], ryu_f64(val[i]).c_str()) > 0;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? ATSC_OK : ATSC_E_IO;
    ATSC_API_END
}

// Metric::append_samples (metric.rs:53-65), index part: every sample's millisecond timestamp becomes
// seconds since midnight and extends the index; the values go to the WavBrro unchanged.
extern "C" int atsc_metric_index_samples(atsc_vsri *index, const int64_t *ts_ms, uint64_t n, uint64_t *failed_at)
{
    ATSC_API_BEGIN
    if (!index || (n && !ts_ms)) return ATSC_E_INVALID;
    for (uint64_t i = 0; i < n; ++i) {
        int32_t sec;
        int rc = atsc_day_elapsed_seconds(ts_ms[i] / 1000, &sec);
        if (rc == ATSC_OK) rc = atsc_vsri_update_for_point(index, sec);
        if (rc) {
            if (failed_at) *failed_at = i;
            return rc;  // Error::UpdateForPointError(sample)
        }
    }
    return ATSC_OK;
    ATSC_API_END
}
// Metric::get_samples (metric.rs:83-97): timestamp of sample i = index.get_time(i); None is an
// unwrap panic there
extern "C" int atsc_metric_sample_times(const atsc_vsri *index, uint64_t n, int64_t *out)
{
    ATSC_API_BEGIN
    if (!index || (n && !out)) return ATSC_E_INVALID;
    for (uint64_t i = 0; i < n; ++i) {
        int32_t t;
        if (i > (uint64_t)INT32_MAX) return ATSC_E_INVALID;
        const int r = atsc_vsri_get_time(index, (int32_t)i, &t);
        if (r != 1) return ATSC_E_INVALID;
        out[i] = (int64_t)t;
    }
    return ATSC_OK;
    ATSC_API_END
}
