/*
 * atsc_hip.h -- C ABI of the MI355X-native ATSC compression core (libatsc_hip.so).
 *
 * This is the drop-in boundary for the per-frame compressor path of
 * instaclustr/atsc.  The reference has no FFI of its own (it is a single Rust
 * process); the seam a replacement must honour is the Rust API both CLIs call:
 *     CompressedStream::{compress_chunk_with, compress_chunk_bounded_with,
 *                        to_bytes, from_bytes, decompress}   atsc/src/data.rs:47-109
 *     OptimizerPlan::{plan, get_execution}                   atsc/src/optimizer/mod.rs:47-109
 * The reference compresses one chunk per call; a GPU wants a batch, so the ABI
 * takes the whole chunk list ("frames") of one or many series at once.  Frame
 * semantics (codec choice, payload bytes, error bound) are per frame and equal
 * the reference's.  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add and where it replaces the loop at atsc/src/main.rs:146-163.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative ATSC_E_* code and never aborts/throws across the ABI
 * (the reference panics instead: data.rs:98, header.rs:34-37,72-74).
 * The caller owns every buffer it passes; the library owns device scratch.
 * One atsc_ctx per host thread (mirrors `&mut self`).  `*_dev` entry points
 * take DEVICE pointers and a hipStream_t (as void*), enqueue work and return
 * without synchronising; the others take HOST pointers and are synchronous.
 *
 * All citations are relative to the reference repository root.
 */
#ifndef ATSC_HIP_H
#define ATSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compressor wire ids == enum Compressor discriminants, atsc/src/compressor/mod.rs:35-44 */
enum {
    ATSC_NOOP = 0,
    ATSC_FFT = 1,
    ATSC_IDW = 2,
    ATSC_CONSTANT = 3,
    ATSC_POLYNOMIAL = 4,
    ATSC_AUTO = 5,
    ATSC_RLE = 6
};

enum {
    ATSC_OK = 0,
    ATSC_E_INVALID = -1,      /* bad argument (null pointer, empty frame, bad level) */
    ATSC_E_NOMEM = -2,        /* host or device allocation failed */
    ATSC_E_UNSUPPORTED = -3,  /* valid in the reference, not implemented here (see atsc_strerror) */
    ATSC_E_NO_DEVICE = -4,    /* no usable HIP device; the library never falls back to a CPU path */
    ATSC_E_HIP = -5,          /* a HIP runtime call failed; see atsc_ctx_last_error */
    ATSC_E_CAPACITY = -6,     /* caller buffer too small */
    ATSC_E_FORMAT = -7,       /* malformed BRO / WBRO / CSV bytes (reference: panic / Err) */
    ATSC_E_VERSION = -8,      /* BRO version newer than 1 (header.rs:30-42) */
    ATSC_E_IO = -9            /* file could not be read / written */
};

typedef struct atsc_ctx atsc_ctx;
typedef struct atsc_plan atsc_plan;
typedef struct atsc_dplan atsc_dplan;
typedef struct atsc_stream atsc_stream;

const char *atsc_strerror(int rc);
const char *atsc_version(void);

/* ------------------------------------------------------------------------ */
/* context                                                                  */
/* ------------------------------------------------------------------------ */

/* Creates a context on HIP device `device` (0-based).  Fails with
 * ATSC_E_NO_DEVICE when there is no GPU: there is no CPU fallback. */
int atsc_ctx_create(atsc_ctx **out, int device);
/* Plans, decode plans and streams created on a context hold device memory of its pool: destroy them
 * before the context. */
void atsc_ctx_destroy(atsc_ctx *ctx);
const char *atsc_ctx_last_error(const atsc_ctx *ctx);
/* What the library keeps between calls, and how to give it back.  A context keeps freed device blocks in a
 * pool (up to 8 GiB) and, for the host-pointer entry points, up to four plans by frame layout with their
 * scratch sets, workspaces and tables; the process keeps the host block the caller released last through
 * atsc_free (decoded results; up to 2 GiB, ATSC_BIG_KEEP_MAX overrides) and the twiddle tables by
 * transform length (up to 128 MB).  atsc_ctx_trim frees the context's share (it synchronises the
 * device; plans the caller created stay valid), atsc_release_caches the process-wide caches. */
int atsc_ctx_trim(atsc_ctx *ctx);
void atsc_release_caches(void);
/* Page-locks / releases caller memory (hipHostRegister): host-pointer calls on registered buffers copy at
 * the link's rate, asynchronously.  The reference's caller owns its chunk (data.rs:56-76); so does this one. */
int atsc_host_register(void *p, uint64_t bytes);
int atsc_host_unregister(void *p);

/* ------------------------------------------------------------------------ */
/* compress: CompressorFrame::{compress, compress_bounded, compress_best}    */
/*           atsc/src/frame/mod.rs:59-149 over a batch of frames             */
/* ------------------------------------------------------------------------ */

/* Frame layout of one batch.  frame_off[n_frames+1] are prefix offsets (in
 * samples) into the sample array: frame i = samples[frame_off[i] .. frame_off[i+1]).
 * Replaces the chunk list of OptimizerPlan::get_execution (optimizer/mod.rs:101-109). */
int atsc_plan_create(atsc_ctx *ctx, const uint64_t *frame_off, uint64_t n_frames,
                     atsc_plan **out);
void atsc_plan_destroy(atsc_plan *plan);
uint64_t atsc_plan_n_frames(const atsc_plan *plan);
uint64_t atsc_plan_n_samples(const atsc_plan *plan);
/* Worst-case bytes of the encoded frame records of this plan (size d_body with it). */
uint64_t atsc_plan_body_bound(const atsc_plan *plan);
/* Worst-case payload bytes of one frame of n samples (RLE, all values distinct). */
uint64_t atsc_payload_bound_bytes(uint64_t n_samples_in_frame);

/* Compresses every frame of `plan` on the GPU.
 *   compressor   ATSC_* id.  ATSC_AUTO = compress_best (frame/mod.rs:71-149).
 *   bounded      1 = compress_chunk_bounded_with (data.rs:56-76), 0 = compress_chunk_with
 *                (data.rs:47-53).  The atsc CLI uses bounded for fft/polynomial/idw/auto and
 *                unbounded for noop/constant/rle (main.rs:150-162).
 *   max_error    the f32 the reference passes: `e as f32 / 100.0` (main.rs:157)
 *   sample_level 0..6, index into COMPRESSION_SPEED (frame/mod.rs:22); only used by ATSC_AUTO
 * Outputs (device memory):
 *   d_body       encoded frame records in frame order, each exactly the bincode image of
 *                CompressorFrame (frame/mod.rs:25-33): varint(41) varint(sample_count)
 *                varint(compressor) varint(len) payload.  A .bro file is
 *                "BRRO" u32le(1) u8(n_frames) varint(n_frames) followed by these bytes
 *                (header.rs:60-67, data.rs:79-85); see atsc_bro_wrap.
 *   d_rec_off    n_frames+1 byte offsets of the records in d_body; [n_frames] = total bytes
 *   d_chosen     n_frames, compressor id actually used (may be NULL)
 *   d_err        n_frames, CompressorResult.error of the chosen codec (may be NULL)
 * Nothing is synchronised; the work is enqueued on `stream`. */
int atsc_compress_plan_dev(atsc_ctx *ctx, const atsc_plan *plan, const double *d_samples,
                           int compressor, int bounded, float max_error, int sample_level,
                           uint8_t *d_body, uint64_t body_cap, uint64_t *d_rec_off,
                           uint8_t *d_chosen, double *d_err, void *stream);

/* The same call for back-to-back batches (a compression service's steady state; main.rs:146-163
 * run over file after file).  Consecutive calls on a plan go round-robin over up to four *chains*
 * inside the plan: a stream owned by the context plus everything a batch in flight owns (payload
 * slots, results, scan scratch, large-tier workspace).  A call's kernels -- codecs and packing -- run on
 * its chain's stream, ordered after everything enqueued on `stream` before the call by one event (only
 * when `stream` still has work in flight), and every chain has two scratch sets, so up to 2 x chains batches
 * are queued or in flight: the gap a single queue leaves between dependent launches (6-10 us here) and the
 * drain of a launch's last workgroups are covered by the other chains' kernels.  Results are those of
 * atsc_compress_plan_dev, byte for byte.
 *   - A call may block the host until the batch that last used its scratch set (2 x chains calls
 *     earlier) has been packed; nothing else is synchronised.
 *   - The outputs of a call are complete once a stream has passed an atsc_plan_join enqueued
 *     after it (or after hipDeviceSynchronize).  Give the calls in flight their own output buffers:
 *     2 x chains sets (eight always suffice).
 *   - d_samples is read by kernels on the context's streams, NOT in `stream` order: it must not be
 *     overwritten until a stream has passed an atsc_plan_input_release (or atsc_plan_join)
 *     enqueued after the call.  (atsc_compress_plan_dev reads it in `stream` order; the large tier's
 *     frames, which it also deals over the context's streams, are joined back before it returns.)
 *   - atsc_compress_plan_dev may be mixed in; it orders itself after the pending batches. */
int atsc_compress_plan_dev_pipelined(atsc_ctx *ctx, const atsc_plan *plan, const double *d_samples,
                                     int compressor, int bounded, float max_error,
                                     int sample_level, uint8_t *d_body, uint64_t body_cap,
                                     uint64_t *d_rec_off, uint8_t *d_chosen, double *d_err,
                                     void *stream);
/* Chains the pipelined calls of this context rotate over (1..4; default 2, or 4 in a process started with
 * GPU_MAX_HW_QUEUES >= 8 -- four chains plus the caller's streams need more hardware queues than the runtime's default
 * four to pay; ATSC_CHAINS overrides the default).  1 keeps every batch on one stream of the context's. */
int atsc_ctx_set_chains(atsc_ctx *ctx, int n);
/* Pipelined calls record how many shader clocks every frame took and start the next batches of the
 * same plan with a class's costliest frames first (frame i of a recurring batch is the same series,
 * one window later); otherwise the frames that run longest start last and the GPU drains half
 * empty.  Only the order of execution changes, never a result.  OFF by default since round 3 (with two chains in
 * flight the next batch's first frames fill the drain the order was there to shorten, and the hint only predicts
 * where the layout recurs -- slot i the same series with the same behaviour, batch after batch); 1 turns it on. */
int atsc_ctx_set_adaptive_order(atsc_ctx *ctx, int on);
/* Makes `stream` wait (device side, no host block) for every batch enqueued so far by
 * atsc_compress_plan_dev_pipelined on `plan`: its records are packed. */
int atsc_plan_join(atsc_ctx *ctx, const atsc_plan *plan, void *stream);
/* Makes `stream` wait (device side) until no kernel of the pipelined calls enqueued so far on `plan`
 * reads their d_samples any more: work enqueued on `stream` afterwards may overwrite the inputs
 * (the reference's caller owns the chunk for the duration of compress_chunk_*, data.rs:47-76). */
int atsc_plan_input_release(atsc_ctx *ctx, const atsc_plan *plan, void *stream);

/* Per-frame diagnostics of the last atsc_compress_plan_dev on this ctx (host copy,
 * synchronises the stream).  One record per frame; used by the parity tests. */
typedef struct {
    uint32_t fft_size, poly_size, rle_size; /* candidate payload bytes; 0xFFFFFFFF = not run */
    uint16_t fft_trips, fft_k;              /* ladder trips (fft.rs:334-353), stored bins */
    uint16_t poly_trips, poly_step;         /* ladder trips (polynomial.rs:231-270), point_step */
    uint32_t poly_points;
    double fft_err, poly_err;
} atsc_frame_diag;
/* Diagnostics cost 40 B/frame of HBM writes, so they are off unless enabled here (or ATSC_DIAG is set). */
int atsc_ctx_enable_diag(atsc_ctx *ctx, int on);
int atsc_ctx_last_diag(atsc_ctx *ctx, atsc_frame_diag *out, uint64_t n_frames);

/* Kernel timing for the roofline report: when on, every compress call hands a HIP event pair to
 * the k_compress dispatch of the frame class holding the most frames (hipExtLaunchKernel start /
 * stop events on the launch stream: the kernel's own timestamps, no marker packets around it; the
 * large-frame tier, several launches, is bracketed by recorded events).  atsc_ctx_profile_read waits
 * for them, returns the summed milliseconds and the number of launches, and resets the counters. */
int atsc_ctx_set_profiling(atsc_ctx *ctx, int on);
int atsc_ctx_profile_read(atsc_ctx *ctx, double *total_ms, uint64_t *launches);

/* Host-pointer convenience: plan + H2D + compress + D2H, synchronous.
 * body_len receives the number of bytes written to `body`. */
int atsc_compress_frames(atsc_ctx *ctx, const double *samples, const uint64_t *frame_off,
                         uint64_t n_frames, int compressor, int bounded, float max_error,
                         int sample_level, uint8_t *body, uint64_t body_cap, uint64_t *body_len,
                         uint64_t *rec_off, uint8_t *chosen, double *err);

/* Several GPUs.  Frames are independent (the loop at main.rs:146-163 shares nothing between chunks), so a
 * batch shards by contiguous frame ranges and the encoded stream is the shards' records laid end to
 * end.  atsc_shard_range is the split every layer uses: unit ranges of sizes differing by at most one,
 * rank order = frame order.  A multi-process host (one process and one atsc_ctx per GPU, as bench.py
 * and atsc_amd/parallel.py run it) calls the single-context entry points on its own range and gathers
 * the record bytes to rank 0.  A single-process host hands one context per device to
 * atsc_compress_frames_sharded: each shard runs on its own host thread, and the outputs are byte for
 * byte those of atsc_compress_frames on one context. */
void atsc_shard_range(uint64_t n_units, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end);
/* The same split with rank 0 carrying root_weight_milli / 1000 times a peer's share: when the records are
 * gathered to rank 0, every peer's bytes cross one link while the root's stay, so the root can take more
 * frames (1000 = atsc_shard_range up to rounding). */
void atsc_shard_range_weighted(uint64_t n_units, uint32_t rank, uint32_t world, uint32_t root_weight_milli,
                               uint64_t *begin, uint64_t *end);
int atsc_compress_frames_sharded(atsc_ctx *const *ctxs, uint32_t n_ctx, const double *samples,
                                 const uint64_t *frame_off, uint64_t n_frames, int compressor, int bounded,
                                 float max_error, int sample_level, uint8_t *body, uint64_t body_cap,
                                 uint64_t *body_len, uint64_t *rec_off, uint8_t *chosen, double *err);

/* ------------------------------------------------------------------------ */
/* decompress: CompressedStream::decompress (data.rs:104-109) ->             */
/*             CompressorFrame::decompress (frame/mod.rs:152-158)            */
/* ------------------------------------------------------------------------ */

/* Parses frame records (host bytes, as produced above, without the leading
 * varint(n_frames) unless has_count != 0) and builds the per-frame table. */
int atsc_dplan_create(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count,
                      atsc_dplan **out);
void atsc_dplan_destroy(atsc_dplan *dp);
uint64_t atsc_dplan_n_frames(const atsc_dplan *dp);
uint64_t atsc_dplan_n_samples(const atsc_dplan *dp);
/* d_body: the same bytes on the device; d_out: atsc_dplan_n_samples doubles. */
int atsc_decompress_plan_dev(atsc_ctx *ctx, const atsc_dplan *dp, const uint8_t *d_body,
                             double *d_out, void *stream);
/* Host-pointer convenience (synchronous). out_n receives the sample count.  On an error `out` may hold
 * samples of the frames in front of the failing record (a destination registered with atsc_host_register
 * is filled part by part while the later records are still being parsed; a pageable one is only written
 * once every payload has decoded): *out_n = 0 then, so that nothing in `out` can be mistaken for a result. */
int atsc_decompress_frames(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count,
                           double *out, uint64_t out_cap, uint64_t *out_n);
/* The same with the output allocated by the library at exactly the decoded length (atsc_free):
 * what CompressedStream::decompress returns (data.rs:104-109), without a sizing pass by the caller. */
int atsc_decompress_frames_alloc(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count,
                                 double **out, uint64_t *out_n);

/* ------------------------------------------------------------------------ */
/* CompressedStream mirror (atsc/src/data.rs:29-110)                          */
/* ------------------------------------------------------------------------ */
/* The reference compresses each chunk when it is added.  Here chunks are copied and queued;
 * the GPU batch runs when the bytes are asked for (atsc_stream_to_bytes) -- observable results are
 * the same, one frame per call in call order. */
int atsc_stream_new(atsc_ctx *ctx, atsc_stream **out);                           /* data.rs:30-35 */
/* data.rs:89-103 ; ATSC_E_FORMAT / ATSC_E_VERSION where the reference panics (header.rs:34-37,72-74) */
int atsc_stream_from_bytes(atsc_ctx *ctx, const uint8_t *bro, uint64_t len, atsc_stream **out);
void atsc_stream_free(atsc_stream *s);
/* data.rs:37-44 : CompressorFrame::new(None) = the default compressor, Noop */
int atsc_stream_compress_chunk(atsc_stream *s, const double *chunk, uint64_t n);
/* data.rs:47-53 ; ATSC_AUTO here is `todo!()` in the reference -> ATSC_E_INVALID */
int atsc_stream_compress_chunk_with(atsc_stream *s, const double *chunk, uint64_t n, int compressor);
/* data.rs:56-76 */
int atsc_stream_compress_chunk_bounded_with(atsc_stream *s, const double *chunk, uint64_t n,
                                            int compressor, float max_error, int compression_speed);
uint64_t atsc_stream_frame_count(const atsc_stream *s);
/* data.rs:79-85 ; *out is malloc'd, release with atsc_free */
int atsc_stream_to_bytes(atsc_stream *s, uint8_t **out, uint64_t *len);
/* data.rs:104-109 ; *out is malloc'd, release with atsc_free */
int atsc_stream_decompress(atsc_stream *s, double **out, uint64_t *n);
void atsc_free(void *p);

/* compress_data / decompress_data of the atsc CLI (atsc/src/main.rs:130-172): clean (drop NaN/Inf),
 * chunk, compress every chunk with `compressor` (bounded with (float)error_pct/100 for
 * fft/polynomial/idw/auto, unbounded for noop/constant/rle), whole .bro image out. */
int atsc_compress_data(atsc_ctx *ctx, const double *data, uint64_t n, int compressor, uint8_t error_pct,
                       int sample_level, uint8_t **bro, uint64_t *len);
int atsc_decompress_data(atsc_ctx *ctx, const uint8_t *bro, uint64_t len, double **out, uint64_t *n);

/* WBRO files (wavbrro/src/wavbrro.rs:103-132, read.rs:23-37, write.rs:21-27):
 * "WBRO0000WBRO" + rkyv 0.7.44 archive of {sample_count:u32, bitdepth:u8 = 5, chunks:Vec<Vec<f64>>}
 * (chunks of 2048 samples).  Byte-identical to the reference's writer. */
int atsc_wbro_from_bytes(const uint8_t *file, uint64_t len, double **out, uint64_t *n);
int atsc_wbro_to_bytes(const double *data, uint64_t n, uint8_t **out, uint64_t *len);
int atsc_wbro_read(const char *path, double **out, uint64_t *n);
int atsc_wbro_write(const char *path, const double *data, uint64_t n);
/* utils/readers/bro_reader.rs:31-46 : *out = NULL and rc 0 when the file is not a BRO file */
int atsc_bro_read_file(const char *path, uint8_t **out, uint64_t *len);
/* atsc/src/csv.rs:36-98 : value column of a comma separated file.  has_header != 0: both field names
 * must be present, values come from value_field; else column 0.  Values parse as Rust's str::parse::<f64>. */
int atsc_csv_read(const char *path, int has_header, const char *time_field, const char *value_field,
                  double **out, uint64_t *n);

/* ------------------------------------------------------------------------ */
/* host-side format helpers (no GPU needed)                                 */
/* ------------------------------------------------------------------------ */

/* OptimizerPlan::get_chunks_sizes, optimizer/mod.rs:78-98.  Returns the chunk count;
 * writes at most cap sizes. */
uint64_t atsc_chunk_sizes(uint64_t len, uint64_t *out, uint64_t cap);
/* OptimizerPlan::clean_data, optimizer/mod.rs:64-71 (drops NaN and +-Inf). Returns kept count. */
uint64_t atsc_clean_data(const double *in, uint64_t n, double *out);
/* utils::next_size, utils/mod.rs:32-38 */
uint64_t atsc_next_size(uint64_t n);
/* CompressorHeader::to_bytes + frame count varint (header.rs:60-67, data.rs:83):
 * writes "BRRO" u32le(1) u8(n_frames mod 256) varint(n_frames) into out (>= 18 bytes);
 * returns bytes written. */
uint64_t atsc_bro_prefix(uint64_t n_frames, uint8_t *out);
/* CompressedStream::from_bytes front half (data.rs:89-97, header.rs:69-84): validates magic
 * and version; returns the offset of the first frame record (after the count varint) and
 * the frame count, or ATSC_E_FORMAT / ATSC_E_VERSION. */
int atsc_bro_open(const uint8_t *bro, uint64_t len, uint64_t *body_off, uint64_t *n_frames);
/* CompressedStream::from_bytes as a dry run (data.rs:89-103): header, version, frame count and the
 * walk over every frame record (frame/mod.rs:25-33) without decoding a payload.  ATSC_E_FORMAT where
 * the reference's bincode decode `.unwrap()` panics (data.rs:98): a truncated or inflated length, a
 * count the bytes cannot hold, an unknown compressor id.  Outputs may be NULL. */
int atsc_bro_scan(const uint8_t *bro, uint64_t len, uint64_t *n_frames, uint64_t *n_samples);

/* ------------------------------------------------------------------------------------------
 * csv-compressor front end: host-only, no GPU context needed.
 * ------------------------------------------------------------------------------------------ */
/* VSRI, the timestamp index (vsri/src/lib.rs): continuous segments y = m*x + b of equally spaced
 * points, [m, x0, y0, count] each (lib.rs:100-106).  Arithmetic is Rust's release-mode i32
 * (wrapping).  Look-ups return 1 = Some(*out), 0 = None, ATSC_E_INVALID where the reference
 * panics (division by zero on a one-point segment, lib.rs:311). */
typedef struct atsc_vsri atsc_vsri;
atsc_vsri *atsc_vsri_new(void);                                      /* Vsri::new, lib.rs:110-119 */
void atsc_vsri_free(atsc_vsri *v);
int atsc_vsri_load(const char *path, atsc_vsri **out);               /* Vsri::load, lib.rs:447-486 */
int atsc_vsri_flush_to(const atsc_vsri *v, const char *path);        /* Vsri::flush_to, lib.rs:424-443 */
/* Vsri::update_for_point (lib.rs:236-273); ATSC_E_INVALID = Error::UpdateIndexForPointError */
int atsc_vsri_update_for_point(atsc_vsri *v, int32_t y);
int32_t atsc_vsri_min(const atsc_vsri *v);                           /* lib.rs:276-278 */
int32_t atsc_vsri_max(const atsc_vsri *v);                           /* lib.rs:281-283 */
uint64_t atsc_vsri_segment_count(const atsc_vsri *v);
int atsc_vsri_segment(const atsc_vsri *v, uint64_t i, int32_t out[4]);
int32_t atsc_vsri_get_sample_count(const atsc_vsri *v);              /* lib.rs:355-358 */
int atsc_vsri_get_sample(const atsc_vsri *v, int32_t y, int32_t *out);           /* lib.rs:301-317 */
int atsc_vsri_get_next_sample(const atsc_vsri *v, int32_t y, int32_t *out);      /* lib.rs:154-169 */
int atsc_vsri_get_previous_sample(const atsc_vsri *v, int32_t y, int32_t *out);  /* lib.rs:175-193 */
int atsc_vsri_get_this_or_next(const atsc_vsri *v, int32_t y, int32_t *out);     /* lib.rs:137-141 */
int atsc_vsri_get_this_or_previous(const atsc_vsri *v, int32_t y, int32_t *out); /* lib.rs:144-148 */
int atsc_vsri_get_time(const atsc_vsri *v, int32_t x, int32_t *out);             /* lib.rs:320-341 */
int atsc_vsri_is_empty(const atsc_vsri *v, int32_t t0, int32_t t1);  /* lib.rs:198-232; 1 / 0 */
int atsc_vsri_get_all_timestamps(const atsc_vsri *v, int32_t **out, uint64_t *n); /* lib.rs:344-353; atsc_free */
/* vsri::day_elapsed_seconds (lib.rs:49-57); ATSC_E_INVALID outside chrono's DateTime range */
int atsc_day_elapsed_seconds(int64_t timestamp_sec, int32_t *out);
/* csv-compressor/src/csv.rs:41-56: `timestamp,value` files (i64, f64; csv crate reader / writer,
 * ryu float formatting).  Arrays from the reader are released with atsc_free. */
int atsc_samples_csv_read(const char *path, int64_t **ts, double **val, uint64_t *n);
int atsc_samples_csv_write(const char *path, const int64_t *ts, const double *val, uint64_t n);
/* Metric::append_samples (metric.rs:53-65), index side: ts_ms[i] / 1000 -> seconds since midnight
 * -> update_for_point.  On failure *failed_at (may be NULL) is the offending sample. */
int atsc_metric_index_samples(atsc_vsri *index, const int64_t *ts_ms, uint64_t n, uint64_t *failed_at);
/* Metric::get_samples (metric.rs:83-97): out[i] = index.get_time(i); ATSC_E_INVALID where that is
 * None (an unwrap panic in the reference). */
int atsc_metric_sample_times(const atsc_vsri *index, uint64_t n, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif
