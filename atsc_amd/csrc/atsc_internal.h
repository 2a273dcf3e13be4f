// Internal structures shared by the host glue and the HIP kernels.
#pragma once
#include <stdint.h>
#include <new>
#include <stdexcept>

namespace atsc {

// One entry per distinct frame length in a plan.  Everything the kernels need that
// depends only on n (reference: fft.rs:298-311, utils/mod.rs:32-49, polynomial.rs:220-224).
struct DevPlan {
    uint32_t n;        // samples in the frame
    uint32_t L;        // transform length: next_size(n) if n >= 128 else n  (fft.rs:305-311)
    uint32_t pre;      // Gibbs prefix padding (fft.rs:187-190)
    uint32_t bins;     // L/2 + 1 (fft.rs:327)
    uint32_t mf;       // max(3, n/100) (fft.rs:298-302)
    uint32_t dk1;      // max(mf/2, 1)  (fft.rs:349)
    uint32_t dk2;      // max(mf/10, 1) (fft.rs:350)
    uint32_t kcap;     // min(bins, mf + 17*dk1 + 5*dk2): most bins the ladder can ever store
    uint32_t direct;   // 1: O(n^2) DFT (n < 128, any n incl. primes); 0: Stockham 2^a 3^b
    uint32_t half;     // 1: L even -> one complex FFT of length M = L/2 over the packed real signal
    uint32_t M;        // FFT length actually run through the Stockham stages (L or L/2)
    uint32_t sc;       // L / M: index scale into the length-L twiddle table
    uint32_t nstages;
    uint32_t radix[14];
    uint32_t stmagic[14];  // floor(2^32 / stride_s) + 1 for t / stride_s (stride_s = product of earlier radices)
    uint32_t p2bins;   // power of two >= bins (sort network size)
    uint32_t p2n;      // power of two >= n
    uint32_t magicL;   // floor(2^32 / L) + 1 (L >= 2): x mod L without a divide
    uint32_t f4_m1, f4_m2;  // M = f4_m1 * f4_m2, f4_m1 <= f4_m2 as close as the factors allow (0: no usable
                            // split): the large tier's LDS-tiled two-pass transform
    uint32_t sp_mf, sp_md;  // M = sp_mf * sp_md, sp_mf the largest divisor <= 992 (0: none): the large
                            // tier's inverse transform from the sparse list of admitted bins -- a direct
                            // sum over sp_md, then LDS transforms of length sp_mf (atsc_large.hip)
    uint32_t lds_bytes;
    // LDS carve offsets (bytes, 16-aligned).  AB = two FFT work buffers of ab_half bytes each,
    // later reused for spline tables, RLE run records and the RLE hash table.
    uint32_t o_xs, o_tw, o_ab, ab_half, ab_bytes, o_sel, o_aux, o_red, o_hist;
    uint64_t tw_off;   // offset (in float2 entries) of this L's table in the twiddle pool
    // The polynomial ladder of a frame of n samples (polynomial.rs:209-277) visits the same
    // (points, step, K) sequence whatever the data: points = base + jump, jump += n/10 (trips 1..17)
    // or n/100 (18..22).  Everything a trip derives from (n, step) is tabulated once per length.
    double inv_n;         // 1.0 / n
    double pry[23];       // 1.0 / step
    double pryL[23];      // 1.0 / gap of the last segment
    uint32_t pstep[23];   // max(n / points, 1)
    uint32_t pK[23];      // number of stored points: ceil(n / step) (+1 when the last sample is not a knot)
    uint32_t pmagic[23];  // floor(2^32 / step) + 1 (step >= 2): i / step without a divide
    uint32_t pgap[23];    // (n - 1) - (K - 2) * step: length of the last segment
};

// LDS carve-up of one compressor frame (bytes).  One definition for the host (DevPlan::o_*) and for
// the kernel instantiations that know the frame length at compile time.  Nothing is kept that two
// phases can share -- the footprint bounds the frames in flight per CU (160 KB / total):
//   red  reduction scratch (one-wavefront classes reduce through DPP: 16 B for the scan total)
//   xs   the frame, f64[n]
//   tw   twiddles float2[L] from the forward FFT to the end of its ladder; otherwise the RLE group
//        table u32[n] and, above it, aux u32[n + 2]
//   ab   two FFT work buffers of ab_half bytes (8 B per complex point, +8 so the second one can hold
//        M + 1 bins); later spline tables, RLE run records (>= 8n + 64 bytes) and the RLE hash table
//   sel  admitted bins, 12 B each
//   hist multi-wavefront classes only: 256 digit counters + 4 words for the radix select that picks
//        the kcap bins the ladder can ever admit before they are sorted
struct EncLds {
    uint32_t o_red, o_xs, o_tw, o_aux, o_ab, ab_half, ab_bytes, o_sel, o_hist, total;
};
#if defined(__HIPCC__)
#define ATSC_HD __host__ __device__
#else
#define ATSC_HD
#endif
ATSC_HD constexpr uint32_t enc_align16(uint32_t v) { return (v + 15u) & ~15u; }
ATSC_HD constexpr uint32_t enc_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
ATSC_HD constexpr EncLds enc_lds(uint32_t n, uint32_t L, uint32_t fft_points /* bins if direct, else M */,
                                 bool direct, uint32_t kcap, bool one_wave)
{
    EncLds e{};
    e.ab_half = enc_align16(direct ? 8 * fft_points : 8 * fft_points + 8);
    e.ab_bytes = enc_max(2 * e.ab_half, enc_align16(8 * n + 64));
    uint32_t o = 0;
    e.o_red = o; o += one_wave ? 16 : 384;
    e.o_xs = o; o += enc_align16(8 * n);
    e.o_tw = o; o += enc_align16(enc_max(8 * L, 4 * n + 4 * (n + 2)));
    e.o_aux = e.o_tw + 4 * n;
    e.o_ab = o; o += e.ab_bytes;
    e.o_sel = o; o += enc_align16(12 * enc_max(kcap, 1u));
    e.o_hist = o; o += one_wave ? 0 : 1040;
    e.total = o;
    return e;
}

struct DevFrame {
    uint64_t sample_off;
    uint64_t slot_off;  // byte offset of the frame's payload slot in the scratch arena
    uint32_t n;
    uint32_t plan;
};

struct DevResult {
    double err;
    uint32_t len;     // payload bytes in the slot
    uint32_t chosen;  // compressor wire id
};

struct KParams {
    double max_err;      // (double)(float) max_error
    double poly_target;  // round_f64(max_err, 3)  (polynomial.rs:230)
    double poly_q_hi;    // largest integer q with q / 10000.0 <= poly_target:  target < round(err,4)  <=>  round(err*1e4) > q_hi
    double poly_q_lo;    // smallest integer q with q / 10000.0 >= poly_target: target > round(err,4)  <=>  round(err*1e4) < q_lo
    int32_t max_err_m;   // (max_err * 1000.0) as i32  (fft.rs:334)
    int32_t mode;        // ATSC_* compressor id
    int32_t bounded;
    int32_t want_diag;
    int32_t debug_stop;  // profiling aid: leave the kernel after phase N (0 = run everything)
    // sample-level selection (frame/mod.rs:89-111)
    int32_t trial;               // 1: this launch is the trial on the frame prefixes: no Constant
                                 //    shortcut, no payload emission, only res[].chosen matters
    uint32_t trial_min_n;        // COMPRESSION_SPEED[level]; frames at least this long use the trial's codec
    const DevResult *trial_res;  // results of the trial launch (indexed like the frames), or null
    uint32_t *cost;              // per frame: shader clocks / 64 this launch took (scheduling hint), or null
    uint32_t large_tiled;        // large tier: in-kernel transforms take the LDS-tiled two-pass form
    uint32_t sparse_inv;         // large tier: per-trip inverse transform from the sparse bin list
    uint32_t prestats;           // large tier, split run: the statistics (LargeStats) and the chunk sums of the first
                                 // polynomial trip are in the frame's workspace slot (k_large_stats, k_large_poly1)
    uint32_t tile_stats;         // large tier, fast path: the frame statistics are the column tiles' records (TileStats)
    uint32_t fast_skip;          // large tier: k_compress_large<0> runs behind the fast path and skips the frames that
                                 // path finished (FastState::status == 2)
    uint32_t prefft;             // large tier: forward transform, untangle and norms were done by the
                                 // batched pre-pass kernels (spectrum, norm bits and non-zero count are
                                 // in the frame's workspace slot)
};
// grid extents of the large tier's pre-pass over a plan's large frames (0 tiles: no pre-pass)
struct LargePre {
    uint32_t tiles1, tiles2, chunks;  // max over frame lengths: column tiles, row tiles, 256-bin chunks
    uint32_t m1_max, m2_max;          // longest sub-transforms: size the tile buffers in LDS
    uint32_t sp_tiles;                // most tiles (8 output columns each) the sparse inverse of a frame has
    uint32_t chunks_n;                // most 4096-sample chunks a frame has (k_large_stats, k_large_poly1)
    uint32_t cols243;                 // 1: every large frame length splits as M = 243 x M2 (k_large_cols243)
    uint32_t rows9p;                  // when every large frame length has M = 243 x 9 P, P = 2 .. 32 a power of two: the
                                      // P values present, as a bit mask (k_large_rows9p<P>); else 0
    uint32_t even_off;                // 1: every large frame starts on an even sample (16-byte pairs, given an aligned base)
};
constexpr uint32_t LARGE_SPLIT_MAX = 128;  // large frames per launch up to which the first FFT trip's tiles
                                           // run as a (tile, frame) grid (launch_compress_large)

// Uniform launch: every frame of the class has the same length and frame f of the class sits at
// sample_off0 + f*n with its slot at slot_off0 + f*slot_stride; the frame descriptor is then
// computed instead of being loaded through ids[] -> frames[] (two dependent loads per workgroup).
struct UniArgs {
    uint32_t enabled;
    uint32_t adaptive;  // the class is walked in the order of ids[] (costliest frames first)
    uint32_t fid0;
    uint64_t sample_off0;
    uint64_t slot_off0;
    uint64_t slot_stride;
    uint32_t n;
    uint32_t plan;
    uint32_t spread, count;  // spread != 0: workgroup b takes frame (b * spread) % count of the class (spread coprime to count)
    // resident launch (k_compress_resident): q_grid workgroups take frames from the counters at queue
    // (RESIDENT_Q_WORDS words, zero between launches; one set per stream)
    uint32_t *queue;
    uint32_t q_grid, q_pad;
};
constexpr uint32_t RESIDENT_Q_WORDS = 288;  // eight heads and the exit count, a 128-byte line each
// workgroups a resident launch of class `cls` with frames of n samples and `lds` bytes of LDS takes (0: that class /
// length has no resident instantiation)
uint32_t resident_grid(int cls, uint32_t n, uint32_t lds);

// parsed frame record for decompression
struct DevDFrame {
    uint64_t payload_off;  // byte offset of the payload in the body
    uint64_t out_off;      // sample offset in the output
    uint32_t payload_len;
    uint32_t n;            // sample_count
    uint32_t tag;          // compressor id
    uint32_t plan;         // index into DevPlan table (by n)
};

static inline uint32_t varint_len_u64(uint64_t v)
{
    return v < 251 ? 1u : v < (1ull << 16) ? 3u : v < (1ull << 32) ? 5u : 9u;
}

// No C++ exception may cross the C ABI (std::bad_alloc / std::length_error out of a container sized from
// untrusted bytes would otherwise end the host process): every extern "C" body that can allocate sits
// between these two.
#define ATSC_API_BEGIN try {
#define ATSC_API_END \
    } catch (const std::bad_alloc &) { return ATSC_E_NOMEM; } \
    catch (const std::length_error &) { return ATSC_E_NOMEM; } \
    catch (...) { return ATSC_E_INVALID; }
// One encoded frame record as it lies in a BRO body (frame/mod.rs:25-33):
// varint(frame_size) varint(sample_count) varint(compressor) varint(len) payload[len].
struct HostRecord {
    uint64_t start;        // offset of the record's first byte
    uint64_t payload_off;  // offset of the payload
    uint64_t payload_len;  // <= UINT32_MAX, and the payload lies inside the body
    uint64_t sample_count;
    uint64_t tag;
};
// bincode varint; false on a truncated field or a marker byte above 253.  `pos` only
// ever moves forward and stays <= len.
static inline bool host_varint(const uint8_t *b, uint64_t len, uint64_t &pos, uint64_t &v)
{
    if (pos >= len) return false;
    const uint8_t t = b[pos];
    uint32_t nb;
    if (t < 251) { v = t; pos += 1; return true; }
    if (t == 251) nb = 2;
    else if (t == 252) nb = 4;
    else if (t == 253) nb = 8;
    else return false;
    if (len - pos < 1ull + nb) return false;
    v = 0;
    for (uint32_t i = 0; i < nb; ++i) v |= (uint64_t)b[pos + 1 + i] << (8 * i);
    pos += 1 + nb;
    return true;
}
// Reads the record at `pos` of untrusted bytes and moves `pos` behind it.  Every bound is checked in the
// overflow-free form (a length field of 2^64 - k must not wrap `pos + len`).
static inline bool host_next_record(const uint8_t *b, uint64_t len, uint64_t &pos, HostRecord &r)
{
    uint64_t fs, p = pos;
    r.start = pos;
    if (!host_varint(b, len, p, fs) || !host_varint(b, len, p, r.sample_count) ||
        !host_varint(b, len, p, r.tag) || !host_varint(b, len, p, r.payload_len))
        return false;
    if (r.payload_len > len - p || r.payload_len > 0xFFFFFFFFull) return false;
    r.payload_off = p;
    pos = p + r.payload_len;
    return true;
}

}  // namespace atsc

// Large host results handed to the caller (decoded samples) come from big_alloc and go back through atsc_free
// -> big_release: the block released last is kept and handed out again when the next result is about the same
// size.  A fresh allocation of 84 MB is untouched address space whose pages fault in one by one during the
// device-to-host copy (milliseconds, more than the copy itself); a recycled block is resident.  At most one
// block (up to BIG_KEEP_MAX bytes) is held.
namespace atsc {
void *big_alloc(size_t bytes);
bool big_release(void *p);  // true: p was a big_alloc block and has been taken care of
void big_trim();            // frees the kept block
}  // namespace atsc

// atsc_compress_frames with the output allocated by the library once its length is known (*out: malloc'd,
// head_room bytes left free in front of the records) and, when nonfinite != NULL, a device-side flag for NaN /
// infinite samples (atsc_compress_data's clean_data check)
struct atsc_ctx;
extern "C" int atsc_internal_compress_frames_scan(atsc_ctx *ctx, const double *samples, const uint64_t *frame_off,
                                                  uint64_t n_frames, int compressor, int bounded, float max_error,
                                                  int sample_level, uint64_t head_room, uint8_t **out,
                                                  uint64_t *body_len, uint64_t *rec_off, int *nonfinite);

// test hook (tests/asan): the host half of atsc_dplan_create on untrusted bytes, no GPU needed
extern "C" int atsc_internal_dplan_parse(const uint8_t *body, uint64_t body_len, int has_count,
                                         uint64_t *n_frames, uint64_t *n_samples);
// test hook: the order in which the kernels' heap replay (hp_*, atsc_device.h) pops `k` of `bins` entries
// with the given f32 norms (runs a one-wavefront kernel; needs a GPU)
extern "C" int atsc_internal_heap_order(const float *norms, uint32_t bins, uint32_t k, uint32_t *order);
