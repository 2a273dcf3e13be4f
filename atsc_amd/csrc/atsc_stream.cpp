// atsc_stream.cpp -- host side above the batch ABI: the reference's calling surface restated in C++
// (Rust is not available in this image): CompressedStream (atsc/src/data.rs), compress_data /
// decompress_data (atsc/src/main.rs:130-172), WBRO files (wavbrro/src/*.rs), BRO file sniffing
// (atsc/src/utils/readers/bro_reader.rs) and the CSV reader (atsc/src/csv.rs).  All compression goes
// through atsc_compress_frames / atsc_decompress_frames, i.e. the GPU; nothing here computes a codec.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <thread>
#include <vector>

#include "../../include/atsc_hip.h"
#include "atsc_internal.h"

namespace {

uint32_t put_varint(uint8_t *p, uint64_t v)
{
    if (v < 251) { p[0] = (uint8_t)v; return 1; }
    uint32_t nb;
    if (v < (1ull << 16)) { p[0] = 251; nb = 2; }
    else if (v < (1ull << 32)) { p[0] = 252; nb = 4; }
    else { p[0] = 253; nb = 8; }
    for (uint32_t i = 0; i < nb; ++i) p[1 + i] = (uint8_t)(v >> (8 * i));
    return nb + 1;
}
bool get_varint(const uint8_t *b, uint64_t len, uint64_t &pos, uint64_t &v)
{
    if (pos >= len) return false;
    const uint8_t t = b[pos];
    uint32_t nb;
    if (t < 251) { v = t; pos += 1; return true; }
    if (t == 251) nb = 2;
    else if (t == 252) nb = 4;
    else if (t == 253) nb = 8;
    else return false;
    if (pos + 1 + nb > len) return false;
    v = 0;
    for (uint32_t i = 0; i < nb; ++i) v |= (uint64_t)b[pos + 1 + i] << (8 * i);
    pos += 1 + nb;
    return true;
}

struct Item {
    // an encoded frame record (varint(41) varint(n) varint(tag) varint(len) payload) ...
    std::vector<uint8_t> record;
    // ... or a queued chunk waiting for the GPU batch
    std::vector<double> chunk;
    int compressor = 0, bounded = 0, level = 0;
    float max_error = 0.0f;
    bool pending = false;
};

}  // namespace

struct atsc_stream {
    atsc_ctx *ctx = nullptr;
    std::vector<Item> items;
};

namespace atsc {
static std::mutex g_big_mu;
static std::map<void *, size_t> g_big_live;   // blocks handed out by big_alloc
static void *g_big_kept = nullptr;            // the block released last, still resident
static size_t g_big_kept_size = 0;
static const size_t BIG_MIN = 8u << 20;
static size_t g_big_keep_max = 2ull << 30;    // ATSC_BIG_KEEP_MAX (bytes) / atsc_release_caches
void *big_alloc(size_t bytes)
{
    if (bytes < BIG_MIN) return malloc(bytes ? bytes : 1);
    static const bool env_read = [] {
        if (const char *e = getenv("ATSC_BIG_KEEP_MAX")) g_big_keep_max = (size_t)strtoull(e, nullptr, 10);
        return true;
    }();
    (void)env_read;
    std::lock_guard<std::mutex> g(g_big_mu);
    void *p = nullptr;
    size_t sz = bytes;
    if (g_big_kept && g_big_kept_size >= bytes && g_big_kept_size <= 2 * bytes) {
        p = g_big_kept;
        sz = g_big_kept_size;
        g_big_kept = nullptr;
        g_big_kept_size = 0;
    } else {
        p = malloc(bytes);
    }
    if (p) g_big_live[p] = sz;
    return p;
}
bool big_release(void *p)
{
    std::lock_guard<std::mutex> g(g_big_mu);
    auto it = g_big_live.find(p);
    if (it == g_big_live.end()) return false;
    const size_t sz = it->second;
    g_big_live.erase(it);
    if (sz <= g_big_keep_max && sz >= g_big_kept_size) {  // keep the larger of the two
        if (g_big_kept) free(g_big_kept);
        g_big_kept = p;
        g_big_kept_size = sz;
    } else {
        free(p);
    }
    return true;
}
// the kept block goes back to the allocator (atsc_release_caches)
void big_trim()
{
    std::lock_guard<std::mutex> g(g_big_mu);
    if (g_big_kept) free(g_big_kept);
    g_big_kept = nullptr;
    g_big_kept_size = 0;
}
}  // namespace atsc

extern "C" void atsc_free(void *p)
{
    if (p && !atsc::big_release(p)) free(p);
}

extern "C" int atsc_stream_new(atsc_ctx *ctx, atsc_stream **out)
{
    if (!ctx || !out) return ATSC_E_INVALID;
    atsc_stream *s = new (std::nothrow) atsc_stream();
    if (!s) return ATSC_E_NOMEM;
    s->ctx = ctx;
    *out = s;
    return ATSC_OK;
}
extern "C" void atsc_stream_free(atsc_stream *s) { delete s; }
extern "C" uint64_t atsc_stream_frame_count(const atsc_stream *s) { return s ? s->items.size() : 0; }

static int queue_chunk(atsc_stream *s, const double *chunk, uint64_t n, int compressor, int bounded,
                       float max_error, int level)
{
    if (!s || !chunk || n == 0) return ATSC_E_INVALID;
    if (compressor < 0 || compressor > 6) return ATSC_E_INVALID;
    if (compressor == ATSC_AUTO && !bounded) return ATSC_E_INVALID;  // compressor/mod.rs:72 todo!()
    if (level < 0 || level > 6) return ATSC_E_INVALID;
    Item it;
    it.chunk.assign(chunk, chunk + n);
    it.compressor = compressor;
    it.bounded = bounded;
    it.max_error = max_error;
    it.level = level;
    it.pending = true;
    s->items.push_back(std::move(it));
    return ATSC_OK;
}
extern "C" int atsc_stream_compress_chunk(atsc_stream *s, const double *chunk, uint64_t n)
{
    ATSC_API_BEGIN
    return queue_chunk(s, chunk, n, ATSC_NOOP, 0, 0.0f, 0);
    ATSC_API_END
}
extern "C" int atsc_stream_compress_chunk_with(atsc_stream *s, const double *chunk, uint64_t n, int compressor)
{
    ATSC_API_BEGIN
    return queue_chunk(s, chunk, n, compressor, 0, 0.0f, 0);
    ATSC_API_END
}
extern "C" int atsc_stream_compress_chunk_bounded_with(atsc_stream *s, const double *chunk, uint64_t n,
                                                       int compressor, float max_error, int compression_speed)
{
    ATSC_API_BEGIN
    // data.rs:68-72: Auto -> compress_best, everything else -> compress_bounded
    return queue_chunk(s, chunk, n, compressor, 1, max_error, compression_speed);
    ATSC_API_END
}

// Runs every queued chunk through the GPU, one batch per distinct (compressor, bounded, error, level).
static int flush(atsc_stream *s)
{
    typedef std::tuple<int, int, uint32_t, int> Key;
    std::map<Key, std::vector<size_t>> groups;
    for (size_t i = 0; i < s->items.size(); ++i) {
        const Item &it = s->items[i];
        if (!it.pending) continue;
        uint32_t eb;
        memcpy(&eb, &it.max_error, 4);
        groups[Key(it.compressor, it.bounded, eb, it.level)].push_back(i);
    }
    for (auto &g : groups) {
        const std::vector<size_t> &idx = g.second;
        std::vector<uint64_t> off(idx.size() + 1, 0);
        for (size_t k = 0; k < idx.size(); ++k) off[k + 1] = off[k] + s->items[idx[k]].chunk.size();
        std::vector<double> flat(off.back());
        for (size_t k = 0; k < idx.size(); ++k) {
            const std::vector<double> &c = s->items[idx[k]].chunk;
            memcpy(flat.data() + off[k], c.data(), c.size() * sizeof(double));
        }
        uint8_t *body = nullptr;
        std::vector<uint64_t> rec_off(idx.size() + 1);
        uint64_t blen = 0;
        const Item &first = s->items[idx[0]];
        int rc = atsc_internal_compress_frames_scan(s->ctx, flat.data(), off.data(), idx.size(), first.compressor,
                                                    first.bounded, first.max_error, first.level, 0, &body, &blen,
                                                    rec_off.data(), nullptr);
        if (rc) return rc;
        std::unique_ptr<uint8_t, void (*)(void *)> body_owner(body, free);
        for (size_t k = 0; k < idx.size(); ++k) {
            Item &it = s->items[idx[k]];
            it.record.assign(body + rec_off[k], body + rec_off[k + 1]);
            it.pending = false;
            std::vector<double>().swap(it.chunk);
        }
    }
    return ATSC_OK;
}

extern "C" int atsc_stream_to_bytes(atsc_stream *s, uint8_t **out, uint64_t *len)
{
    ATSC_API_BEGIN
    if (!s || !out || !len) return ATSC_E_INVALID;
    int rc = flush(s);
    if (rc) return rc;
    uint64_t total = 18;
    for (const Item &it : s->items) total += it.record.size();
    uint8_t *buf = (uint8_t *)malloc(total);
    if (!buf) return ATSC_E_NOMEM;
    uint64_t pos = atsc_bro_prefix(s->items.size(), buf);  // header.rs:60-67 + data.rs:83
    for (const Item &it : s->items) {
        memcpy(buf + pos, it.record.data(), it.record.size());
        pos += it.record.size();
    }
    *out = buf;
    *len = pos;
    return ATSC_OK;
    ATSC_API_END
}

extern "C" int atsc_stream_from_bytes(atsc_ctx *ctx, const uint8_t *bro, uint64_t len, atsc_stream **out)
{
    ATSC_API_BEGIN
    if (!ctx || !bro || !out) return ATSC_E_INVALID;
    *out = nullptr;
    uint64_t pos = 0, nf = 0;
    int rc = atsc_bro_open(bro, len, &pos, &nf);
    if (rc) return rc;
    if (nf > len / 4) return ATSC_E_FORMAT;  // untrusted count: every record takes at least 4 bytes
    std::unique_ptr<atsc_stream> s(new (std::nothrow) atsc_stream());
    if (!s) return ATSC_E_NOMEM;
    s->ctx = ctx;
    for (uint64_t f = 0; f < nf; ++f) {
        atsc::HostRecord hr;
        if (!atsc::host_next_record(bro, len, pos, hr)) return ATSC_E_FORMAT;  // bincode decode .unwrap() (data.rs:98)
        Item it;
        it.record.assign(bro + hr.start, bro + pos);
        s->items.push_back(std::move(it));
    }
    *out = s.release();
    return ATSC_OK;
    ATSC_API_END
}

// CompressedStream::from_bytes (data.rs:89-103) as a dry run: header, version, frame count and the walk over
// the records, nothing decoded.  Host only.
extern "C" int atsc_bro_scan(const uint8_t *bro, uint64_t len, uint64_t *n_frames, uint64_t *n_samples)
{
    ATSC_API_BEGIN
    if (!bro) return ATSC_E_INVALID;
    uint64_t pos = 0, nf = 0, ns = 0;
    int rc = atsc_bro_open(bro, len, &pos, &nf);
    if (rc) return rc;
    if (nf > len / 4) return ATSC_E_FORMAT;
    for (uint64_t f = 0; f < nf; ++f) {
        atsc::HostRecord hr;
        if (!atsc::host_next_record(bro, len, pos, hr)) return ATSC_E_FORMAT;
        if (hr.tag > 6 || hr.tag == ATSC_AUTO) return ATSC_E_FORMAT;
        if (ns + hr.sample_count < ns) return ATSC_E_FORMAT;
        ns += hr.sample_count;
    }
    if (n_frames) *n_frames = nf;
    if (n_samples) *n_samples = ns;
    return ATSC_OK;
    ATSC_API_END
}

extern "C" int atsc_stream_decompress(atsc_stream *s, double **out, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!s || !out || !n) return ATSC_E_INVALID;
    int rc = flush(s);
    if (rc) return rc;
    std::vector<uint8_t> body;
    uint64_t samples = 0;
    for (const Item &it : s->items) {
        body.insert(body.end(), it.record.begin(), it.record.end());
        uint64_t pos = 0, fs, sc = 0;
        get_varint(it.record.data(), it.record.size(), pos, fs);
        get_varint(it.record.data(), it.record.size(), pos, sc);
        samples += sc;
    }
    *out = nullptr;
    *n = 0;
    if (s->items.empty()) { *out = (double *)malloc(8); return *out ? ATSC_OK : ATSC_E_NOMEM; }
    // Noop frames decode to their stored count, not sample_count (noop.rs:79-83): ask the decoder
    uint64_t cap = samples + 16;
    double *buf = (double *)atsc::big_alloc(cap * sizeof(double));
    if (!buf) return ATSC_E_NOMEM;
    uint64_t got = 0;
    rc = atsc_decompress_frames(s->ctx, body.data(), body.size(), 0, buf, cap, &got);
    if (rc == ATSC_E_CAPACITY) {
        atsc_free(buf);
        cap = got;
        buf = (double *)atsc::big_alloc((cap ? cap : 1) * sizeof(double));
        if (!buf) return ATSC_E_NOMEM;
        rc = atsc_decompress_frames(s->ctx, body.data(), body.size(), 0, buf, cap, &got);
    }
    if (rc) { atsc_free(buf); return rc; }
    *out = buf;
    *n = got;
    return ATSC_OK;
    ATSC_API_END
}

// ------------------------------------------------------------------------------------------
// main.rs:130-172
// ------------------------------------------------------------------------------------------
extern "C" int atsc_compress_data(atsc_ctx *ctx, const double *data, uint64_t n, int compressor,
                                  uint8_t error_pct, int sample_level, uint8_t **bro, uint64_t *len)
{
    ATSC_API_BEGIN
    if (!ctx || (!data && n) || !bro || !len) return ATSC_E_INVALID;
    *bro = nullptr;
    *len = 0;
    if (compressor < 0 || compressor > 6 || sample_level < 0 || sample_level > 6) return ATSC_E_INVALID;
    // OptimizerPlan::plan drops NaN and infinite samples before it chunks (optimizer/mod.rs:47-49,64-71).  A
    // series almost never holds one, and a host pass that finds out costs one core's memory speed over all of it
    // (tens of milliseconds for 84 MB).  So the samples go to the GPU as they are, the check runs there beside the
    // codecs (k_nonfinite_flag), and only when it fires is the result thrown away and the series compressed
    // again from a cleaned copy.
    std::vector<double> clean;
    const double *src = data;
    uint64_t cn = n;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint64_t nch = atsc_chunk_sizes(cn, nullptr, 0);
        if (nch == 0) {  // an empty stream: header + count 0 (data.rs:79-85)
            uint8_t *buf = (uint8_t *)malloc(18);
            if (!buf) return ATSC_E_NOMEM;
            *len = atsc_bro_prefix(0, buf);
            *bro = buf;
            return ATSC_OK;
        }
        std::vector<uint64_t> sizes(nch), off(nch + 1, 0);
        atsc_chunk_sizes(cn, sizes.data(), nch);
        for (uint64_t c = 0; c < nch; ++c) off[c + 1] = off[c] + sizes[c];
        // main.rs:146-163: every chunk of the plan goes through the same call, bounded for the lossy
        // codecs and Auto, plain otherwise -- one batch for the GPU, written behind the stream prefix
        const bool lossy = compressor == ATSC_FFT || compressor == ATSC_POLYNOMIAL || compressor == ATSC_IDW ||
                           compressor == ATSC_AUTO;                            // main.rs:150-162
        const float max_error = lossy ? (float)error_pct / 100.0f : 0.0f;      // main.rs:157
        uint8_t prefix[18];
        const uint64_t pre = atsc_bro_prefix(nch, prefix);
        uint8_t *buf = nullptr;
        uint64_t blen = 0;
        int dirty = 0;
        const int rc = atsc_internal_compress_frames_scan(ctx, src, off.data(), nch, compressor, lossy ? 1 : 0, max_error,
                                                          lossy ? sample_level : 0, pre, &buf, &blen, nullptr, &dirty);
        if (dirty && attempt == 0) {  // compressed with samples the reference drops: again, from the cleaned copy
            free(buf);
            clean.resize(n);
            cn = atsc_clean_data(data, n, clean.data());
            src = clean.data();
            continue;
        }
        if (rc) { free(buf); return rc; }
        memcpy(buf, prefix, pre);
        *bro = buf;
        *len = pre + blen;
        return ATSC_OK;
    }
    return ATSC_E_INVALID;  // not reached: a cleaned copy holds nothing left to drop
    ATSC_API_END
}

extern "C" int atsc_decompress_data(atsc_ctx *ctx, const uint8_t *bro, uint64_t len, double **out, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!ctx || !bro || !out || !n) return ATSC_E_INVALID;
    *out = nullptr;
    *n = 0;
    uint64_t body_off = 0, n_frames = 0;
    int rc = atsc_bro_open(bro, len, &body_off, &n_frames);  // data.rs:89-97, header.rs:69-84
    if (rc) return rc;
    if (n_frames == 0) {
        *out = (double *)malloc(8);
        return *out ? ATSC_OK : ATSC_E_NOMEM;
    }
    // The records go to the GPU as they lie in the file, from the frame-count varint (offset 9, right
    // after the header) on: exactly that many records are decoded and anything behind them is ignored,
    // as bincode's decode_from_slice does (data.rs:98); the output is sized by the decoder.
    (void)body_off;
    return atsc_decompress_frames_alloc(ctx, bro + 9, len - 9, 1, out, n);
    ATSC_API_END
}

// ------------------------------------------------------------------------------------------
// files
// ------------------------------------------------------------------------------------------
static int read_file(const char *path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return ATSC_E_IO;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return ATSC_E_IO; }
    buf.resize((size_t)sz);
    const size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == (size_t)sz ? ATSC_OK : ATSC_E_IO;
}
static int write_file(const char *path, const uint8_t *p, uint64_t len)
{
    FILE *f = fopen(path, "wb");
    if (!f) return ATSC_E_IO;
    const size_t w = len ? fwrite(p, 1, len, f) : 0;
    fclose(f);
    return w == len ? ATSC_OK : ATSC_E_IO;
}

// rkyv 0.7.44 archive written by WavBrro::to_bytes (wavbrro.rs:126-132), layout per SURVEY App. B:
//   [chunk 0 f64 data][chunk 1 ...] [n_chunks x {rel_off:i32, len:u32}] [root: {rel_off:i32,
//   n_chunks:u32, sample_count:u32, bitdepth:u8, pad[3]}]   -- relative offsets are from the field
extern "C" int atsc_wbro_to_bytes(const double *data, uint64_t n, uint8_t **out, uint64_t *len)
{
    ATSC_API_BEGIN
    if ((!data && n) || !out || !len) return ATSC_E_INVALID;
    const uint64_t CH = 2048;  // MAX_CHUNK_SIZE, wavbrro.rs:24
    const uint64_t nch = (n + CH - 1) / CH;
    // rkyv's relative offsets are i32 (and sample_count u32): an archive above 2 GiB cannot be expressed;
    // the reference's serializer fails there as well
    if (n > 0xFFFFFFFFull || 8 * n + 8 * nch + 16 > 0x7FFFFFFFull) return ATSC_E_UNSUPPORTED;
    const uint64_t total = 12 + 8 * n + 8 * nch + 16;
    uint8_t *buf = (uint8_t *)calloc(total, 1);
    if (!buf) return ATSC_E_NOMEM;
    memcpy(buf, "WBRO0000WBRO", 12);  // write.rs:22
    uint8_t *body = buf + 12;
    if (n) memcpy(body, data, 8 * n);
    const uint64_t ent0 = 8 * n;
    for (uint64_t c = 0; c < nch; ++c) {
        const uint64_t ent = ent0 + 8 * c;
        const int32_t rel = (int32_t)((int64_t)(8 * CH * c) - (int64_t)ent);
        const uint32_t cl = (uint32_t)std::min<uint64_t>(CH, n - CH * c);
        memcpy(body + ent, &rel, 4);
        memcpy(body + ent + 4, &cl, 4);
    }
    const uint64_t root = ent0 + 8 * nch;
    const int32_t rrel = (int32_t)((int64_t)ent0 - (int64_t)root);
    const uint32_t nch32 = (uint32_t)nch, sc = (uint32_t)n;  // `sample_count as u32` (wavbrro.rs:81)
    memcpy(body + root, &rrel, 4);
    memcpy(body + root + 4, &nch32, 4);
    memcpy(body + root + 8, &sc, 4);
    body[root + 12] = 5;  // bitdepth: f64
    *out = buf;
    *len = total;
    return ATSC_OK;
    ATSC_API_END
}

extern "C" int atsc_wbro_from_bytes(const uint8_t *file, uint64_t len, double **out, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!file || !out || !n) return ATSC_E_INVALID;
    if (len < 12 + 16 || memcmp(file, "WBRO", 4) != 0 || memcmp(file + 8, "WBRO", 4) != 0)
        return ATSC_E_FORMAT;  // read.rs:23-29 -> Error::FormatError
    const uint8_t *body = file + 12;
    const uint64_t blen = len - 12;
    const uint64_t root = blen - 16;
    int32_t rrel;
    uint32_t nch, sc;
    memcpy(&rrel, body + root, 4);
    memcpy(&nch, body + root + 4, 4);
    memcpy(&sc, body + root + 8, 4);
    const int64_t ent0 = (int64_t)root + rrel;
    if (ent0 < 0 || (uint64_t)ent0 + 8ull * nch > root) return ATSC_E_FORMAT;
    double *buf = (double *)malloc(((uint64_t)sc ? sc : 1) * sizeof(double));
    if (!buf) return ATSC_E_NOMEM;
    uint64_t k = 0;
    for (uint32_t c = 0; c < nch; ++c) {
        const int64_t ent = ent0 + 8ll * c;
        int32_t rel;
        uint32_t cl;
        memcpy(&rel, body + ent, 4);
        memcpy(&cl, body + ent + 4, 4);
        const int64_t start = ent + rel;
        if (start < 0 || (uint64_t)start + 8ull * cl > blen || k + cl > sc) { free(buf); return ATSC_E_FORMAT; }
        memcpy(buf + k, body + start, 8ull * cl);
        k += cl;
    }
    *out = buf;
    *n = k;
    return ATSC_OK;
    ATSC_API_END
}
extern "C" int atsc_wbro_read(const char *path, double **out, uint64_t *n)
{
    ATSC_API_BEGIN
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    return atsc_wbro_from_bytes(buf.data(), buf.size(), out, n);
    ATSC_API_END
}
extern "C" int atsc_wbro_write(const char *path, const double *data, uint64_t n)
{
    ATSC_API_BEGIN
    uint8_t *buf = nullptr;
    uint64_t len = 0;
    int rc = atsc_wbro_to_bytes(data, n, &buf, &len);
    if (rc) return rc;
    rc = write_file(path, buf, len);
    free(buf);
    return rc;
    ATSC_API_END
}

// utils/readers/bro_reader.rs:31-46: needs 12 readable bytes; a BRO file starts with "BRRO"
extern "C" int atsc_bro_read_file(const char *path, uint8_t **out, uint64_t *len)
{
    ATSC_API_BEGIN
    if (!path || !out || !len) return ATSC_E_INVALID;
    *out = nullptr;
    *len = 0;
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    if (buf.size() < 12) return ATSC_E_IO;  // read_exact on a 12-byte header fails
    if (memcmp(buf.data(), "BRRO", 4) != 0) return ATSC_OK;  // Ok(None): not a BRO file, skipped
    uint8_t *p = (uint8_t *)malloc(buf.size());
    if (!p) return ATSC_E_NOMEM;
    memcpy(p, buf.data(), buf.size());
    *out = p;
    *len = buf.size();
    return ATSC_OK;
    ATSC_API_END
}

// Rust's `str::parse::<f64>` grammar: [+-]? ( "inf" | "infinity" | "nan" (any case) | digits [. digits?]
// [e[+-]digits] | . digits [e...] ).  No surrounding whitespace, no hex.
static bool parse_rust_f64(const std::string &s, double &v)
{
    size_t i = 0;
    const size_t n = s.size();
    if (n == 0) return false;
    if (s[i] == '+' || s[i] == '-') ++i;
    if (i >= n) return false;
    std::string rest;
    for (size_t k = i; k < n; ++k) rest.push_back((char)tolower((unsigned char)s[k]));
    if (rest == "inf" || rest == "infinity" || rest == "nan") {
        v = rest == "nan" ? NAN : (s[0] == '-' ? -INFINITY : INFINITY);
        return true;
    }
    size_t digits = 0;
    while (i < n && isdigit((unsigned char)s[i])) { ++i; ++digits; }
    if (i < n && s[i] == '.') {
        ++i;
        while (i < n && isdigit((unsigned char)s[i])) { ++i; ++digits; }
    }
    if (digits == 0) return false;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
        size_t ed = 0;
        while (i < n && isdigit((unsigned char)s[i])) { ++i; ++ed; }
        if (ed == 0) return false;
    }
    if (i != n) return false;
    v = strtod(s.c_str(), nullptr);  // correctly rounded, as Rust's parser
    return true;
}

// one CSV record -> fields (RFC 4180 quoting as the csv crate's default reader)
static void split_csv(const std::string &line, std::vector<std::string> &f)
{
    f.clear();
    std::string cur;
    bool q = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (q) {
            if (c == '"' && i + 1 < line.size() && line[i + 1] == '"') { cur.push_back('"'); ++i; }
            else if (c == '"') q = false;
            else cur.push_back(c);
        } else if (c == '"' && cur.empty()) q = true;
        else if (c == ',') { f.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    f.push_back(cur);
}

// shared with atsc_vsri.cpp (the csv-compressor front end)
namespace atsc_text {
bool parse_rust_f64(const std::string &s, double &v) { return ::parse_rust_f64(s, v); }
void split_csv(const std::string &line, std::vector<std::string> &f) { ::split_csv(line, f); }
// BufRead::lines / the csv reader's record split: "\n" or "\r\n" ends a line, no empty line after
// the final terminator
int read_lines(const char *path, std::vector<std::string> &lines)
{
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    std::string cur;
    for (uint8_t c : buf) {
        if (c == '\n') {
            if (!cur.empty() && cur.back() == '\r') cur.pop_back();
            lines.push_back(cur);
            cur.clear();
        } else {
            cur.push_back((char)c);
        }
    }
    if (!cur.empty()) lines.push_back(cur);
    return ATSC_OK;
}
}  // namespace atsc_text

extern "C" int atsc_csv_read(const char *path, int has_header, const char *time_field, const char *value_field,
                             double **out, uint64_t *n)
{
    ATSC_API_BEGIN
    if (!path || !out || !n) return ATSC_E_INVALID;
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;  // Error::OpenFileFailed
    std::vector<std::string> lines;
    {
        std::string cur;
        for (uint8_t c : buf) {
            if (c == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); lines.push_back(cur); cur.clear(); }
            else cur.push_back((char)c);
        }
        if (!cur.empty()) lines.push_back(cur);
    }
    std::vector<std::string> f;
    size_t li = 0, col = 0, width = 0;
    while (li < lines.size() && lines[li].empty()) ++li;  // the csv crate skips empty lines
    if (has_header) {
        if (li >= lines.size()) return ATSC_E_FORMAT;
        split_csv(lines[li++], f);
        width = f.size();
        bool have_t = false, have_v = false;
        for (size_t k = 0; k < f.size(); ++k) {
            if (time_field && !have_t && f[k] == time_field) have_t = true;
            if (value_field && !have_v && f[k] == value_field) { have_v = true; col = k; }
        }
        if (!have_t || !have_v) return ATSC_E_FORMAT;  // TimestampFieldNotFound / ValueFieldNotFound
    }
    std::vector<double> vals;
    for (; li < lines.size(); ++li) {
        if (lines[li].empty()) continue;
        split_csv(lines[li], f);
        if (width == 0) width = f.size();
        if (f.size() != width || col >= f.size()) return ATSC_E_FORMAT;  // csv: unequal record lengths
        double v;
        if (!parse_rust_f64(f[col], v)) return ATSC_E_FORMAT;  // Error::ParsingValueFailed
        vals.push_back(v);
    }
    double *p = (double *)malloc((vals.size() ? vals.size() : 1) * sizeof(double));
    if (!p) return ATSC_E_NOMEM;
    if (!vals.empty()) memcpy(p, vals.data(), vals.size() * sizeof(double));
    *out = p;
    *n = vals.size();
    return ATSC_OK;
    ATSC_API_END
}
