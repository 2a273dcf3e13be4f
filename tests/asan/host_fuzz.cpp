// tests/asan/host_fuzz.cpp -- drives the host-side parsers of libatsc_hip.so over mutated inputs in a
// build with AddressSanitizer + UBSan (g++, no GPU).  Every file of the corpus directory (valid .bro /
// .wbro / .vsri / .csv files written by tests/test_malformed_streams.py) is fed through the entry point
// for its format as it is, then truncated at every length, with single-bit flips, and with every varint
// field of the BRO framing replaced by hostile values (wrapping 64-bit lengths, inflated counts).
// A parser may answer ATSC_OK or a negative code; it may not crash, hang, read out of bounds or overflow.
// Usage: host_fuzz <corpus dir> <scratch dir>
#include <dirent.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/atsc_hip.h"
#include "../../atsc_amd/csrc/atsc_internal.h"

static uint64_t g_calls = 0, g_ok = 0, g_rejected = 0;
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void tally(int rc)
{
    ++g_calls;
    if (rc == ATSC_OK) ++g_ok;
    else if (rc < 0 && rc >= ATSC_E_IO) ++g_rejected;
    else { fprintf(stderr, "unexpected return code %d\n", rc); exit(3); }
}

static void feed_bro(const std::vector<uint8_t> &b)
{
    // heap copy of the exact length: reads one past the end are caught by the sanitizer
    uint8_t *p = (uint8_t *)malloc(b.size() ? b.size() : 1);
    if (!b.empty()) memcpy(p, b.data(), b.size());
    uint64_t nf = 0, ns = 0, off = 0;
    tally(atsc_bro_open(p, b.size(), &off, &nf));
    tally(atsc_bro_scan(p, b.size(), &nf, &ns));
    if (b.size() >= 9) {
        tally(atsc_internal_dplan_parse(p + 9, b.size() - 9, 1, &nf, &ns));  // as atsc_decompress_data hands it on
        uint64_t o2 = 0, n2 = 0;
        if (atsc_bro_open(p, b.size(), &o2, &n2) == ATSC_OK && o2 <= b.size())
            tally(atsc_internal_dplan_parse(p + o2, b.size() - o2, 0, &nf, &ns));  // records without the count
    }
    free(p);
}
static void feed_wbro(const std::vector<uint8_t> &b)
{
    uint8_t *p = (uint8_t *)malloc(b.size() ? b.size() : 1);
    if (!b.empty()) memcpy(p, b.data(), b.size());
    double *out = nullptr;
    uint64_t n = 0;
    const int rc = atsc_wbro_from_bytes(p, b.size(), &out, &n);
    tally(rc);
    if (rc == ATSC_OK) {
        volatile double sink = 0;
        for (uint64_t i = 0; i < n; ++i) sink = sink + out[i];  // the whole result is readable
        atsc_free(out);
    }
    free(p);
}
static void write_file(const std::string &path, const std::vector<uint8_t> &b)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(4); }
    if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
}
static void feed_vsri(const std::vector<uint8_t> &b, const std::string &scratch)
{
    const std::string path = scratch + "/m.vsri";
    write_file(path, b);
    atsc_vsri *v = nullptr;
    const int rc = atsc_vsri_load(path.c_str(), &v);
    tally(rc);
    if (rc == ATSC_OK) {
        int32_t o = 0, *all = nullptr;
        uint64_t cnt = 0;
        (void)atsc_vsri_get_sample_count(v);
        (void)atsc_vsri_get_sample(v, atsc_vsri_min(v), &o);
        (void)atsc_vsri_get_time(v, 0, &o);
        (void)atsc_vsri_get_this_or_next(v, 1, &o);
        (void)atsc_vsri_get_this_or_previous(v, 86399, &o);
        (void)atsc_vsri_is_empty(v, 0, 86400);
        if (atsc_vsri_segment_count(v) < 64 && atsc_vsri_get_sample_count(v) >= 0 && atsc_vsri_get_sample_count(v) < 100000 &&
            atsc_vsri_get_all_timestamps(v, &all, &cnt) == ATSC_OK)
            atsc_free(all);
        (void)atsc_vsri_update_for_point(v, (int32_t)(rnd() % 86400));
        atsc_vsri_free(v);
    }
}
static void feed_csv(const std::vector<uint8_t> &b, const std::string &scratch)
{
    const std::string path = scratch + "/m.csv";
    write_file(path, b);
    double *vals = nullptr;
    uint64_t n = 0;
    int rc = atsc_csv_read(path.c_str(), 1, "time", "value", &vals, &n);
    tally(rc);
    if (rc == ATSC_OK) atsc_free(vals);
    rc = atsc_csv_read(path.c_str(), 0, nullptr, nullptr, &vals, &n);
    tally(rc);
    if (rc == ATSC_OK) atsc_free(vals);
    int64_t *ts = nullptr;
    rc = atsc_samples_csv_read(path.c_str(), &ts, &vals, &n);
    tally(rc);
    if (rc == ATSC_OK) { atsc_free(ts); atsc_free(vals); }
}

// positions and widths of the varint fields of a well-formed .bro image
struct Field { size_t off; unsigned width; };
static std::vector<Field> bro_fields(const std::vector<uint8_t> &b)
{
    std::vector<Field> f;
    uint64_t pos = 9, v = 0;
    auto take = [&](uint64_t &out) {
        const uint64_t at = pos;
        if (!atsc::host_varint(b.data(), b.size(), pos, out)) return false;
        f.push_back({(size_t)at, (unsigned)(pos - at)});
        return true;
    };
    uint64_t nf = 0;
    if (b.size() < 10 || !take(nf)) return f;
    for (uint64_t i = 0; i < nf; ++i) {
        uint64_t len = 0;
        if (!take(v) || !take(v) || !take(v) || !take(len)) break;
        if (len > b.size() - pos) break;
        pos += len;
    }
    return f;
}
static void put_varint(std::vector<uint8_t> &o, uint64_t v)
{
    if (v < 251) { o.push_back((uint8_t)v); return; }
    unsigned nb;
    if (v < (1ull << 16)) { o.push_back(251); nb = 2; }
    else if (v < (1ull << 32)) { o.push_back(252); nb = 4; }
    else { o.push_back(253); nb = 8; }
    for (unsigned i = 0; i < nb; ++i) o.push_back((uint8_t)(v >> (8 * i)));
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_fuzz <corpus dir> <scratch dir>\n"); return 2; }
    const std::string dir = argv[1], scratch = argv[2];
    DIR *d = opendir(dir.c_str());
    if (!d) { perror(dir.c_str()); return 2; }
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
    // the advisor's reproducer (round 1): a 30-byte BRO whose payload length is 2^64 - 20 / 2^64 - 12
    for (uint64_t evil : {~0ull - 19, ~0ull - 11, ~0ull, 1ull << 63, 0xFFFFFFFFull, 0x100000000ull}) {
        std::vector<uint8_t> b = {'B', 'R', 'R', 'O', 1, 0, 0, 0, 1, 1, 41, 251, 0, 4, 3};
        put_varint(b, evil);
        while (b.size() < 30) b.push_back(30);
        feed_bro(b);
        b[9] = 253;  // and the frame count itself as a 64-bit field running off the end
        feed_bro(b);
    }
    for (const std::string &nm : names) {
        const std::string path = dir + "/" + nm;
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) continue;
        std::vector<uint8_t> b;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + got);
        fclose(f);
        const std::string ext = nm.substr(nm.rfind('.') == std::string::npos ? 0 : nm.rfind('.'));
        auto feed = [&](const std::vector<uint8_t> &x) {
            if (ext == ".bro") feed_bro(x);
            else if (ext == ".wbro") feed_wbro(x);
            else if (ext == ".vsri") feed_vsri(x, scratch);
            else feed_csv(x, scratch);
        };
        const bool files = (ext == ".vsri" || ext == ".csv");  // these go through the file system: fewer rounds
        feed(b);
        // truncations: every length for short inputs, 512 sampled lengths otherwise
        const size_t tstep = std::max<size_t>(1, b.size() / (files ? 64 : 512));
        for (size_t len = 0; len < b.size(); len += (b.size() <= 4096 && !files) ? 1 : tstep)
            feed(std::vector<uint8_t>(b.begin(), b.begin() + len));
        // single-bit flips and byte stomps
        const int rounds = files ? 200 : 4000;
        for (int r = 0; r < rounds && !b.empty(); ++r) {
            std::vector<uint8_t> m = b;
            const int k = 1 + (int)(rnd() % 3);
            for (int j = 0; j < k; ++j) {
                const size_t at = rnd() % m.size();
                if (rnd() & 1) m[at] ^= (uint8_t)(1u << (rnd() % 8));
                else m[at] = (uint8_t)rnd();
            }
            feed(m);
        }
        if (ext == ".bro") {
            // every varint of the framing replaced by hostile values, the rest of the file kept
            static const uint64_t evil[] = {0, 1, 250, 251, 65535, 65536, 131072, 131073, 0xFFFFFFFFull, 0x100000000ull,
                                            1ull << 62, 1ull << 63, ~0ull, ~0ull - 8, ~0ull - 19, ~0ull - 11};
            const std::vector<Field> flds = bro_fields(b);
            const size_t fstep = std::max<size_t>(1, flds.size() / 400);
            for (size_t i = 0; i < flds.size(); i += fstep)
                for (uint64_t e : evil) {
                    std::vector<uint8_t> m(b.begin(), b.begin() + flds[i].off);
                    put_varint(m, e);
                    m.insert(m.end(), b.begin() + flds[i].off + flds[i].width, b.end());
                    feed(m);
                }
        }
    }
    printf("host_fuzz: %llu parser calls, %llu accepted, %llu rejected, %zu corpus files\n",
           (unsigned long long)g_calls, (unsigned long long)g_ok, (unsigned long long)g_rejected, names.size());
    return 0;
}
