/*
 * atsc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single-threaded) of the instaclustr/atsc per-frame
 * compressor path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (atsc_amd/) never
 * links, imports or calls it.
 *
 * Parity status: PINNED by the reference's own known-answer tests (see
 * tests/test_oracle_kat.py).  The reference is Rust and cannot be built in
 * this image (no cargo/rustc), so there is no oracle/_ref build.
 * Third-party arithmetic restated here (crates absent from /root/reference):
 *   rustfft 6.2.0 (Cargo.lock:1020), splines 4.3.1 (Cargo.lock:1107),
 *   inverse_distance_weight 0.1.1 (Cargo.lock:632), bincode 2.0.0-rc.3
 *   (Cargo.lock:143).  FFT bins are pinned to f32 rounding only (rustfft's
 *   butterfly order is CPU dependent); everything else is pinned bit-exactly.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef ATSC_ORACLE_H
#define ATSC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compressor wire ids -- atsc/src/compressor/mod.rs:35-44 */
enum {
    ORC_NOOP = 0,
    ORC_FFT = 1,
    ORC_IDW = 2,
    ORC_CONSTANT = 3,
    ORC_POLYNOMIAL = 4,
    ORC_AUTO = 5,
    ORC_RLE = 6
};

/* Bitdepth wire ids -- atsc/src/optimizer/utils.rs:21-26 */
enum { ORC_BD_F64 = 0, ORC_BD_I32 = 1, ORC_BD_I16 = 2, ORC_BD_U8 = 3 };

typedef struct {
    double max;
    uint64_t max_loc;
    double min;
    uint64_t min_loc;
    double mean;
    int32_t bitdepth;
    int32_t fractional;
} orc_stats;

/* growable byte buffer handed back to python (free with orc_free) */
typedef struct {
    uint8_t *ptr;
    size_t len;
    size_t cap;
} orc_buf;

void orc_free(void *p);

/* ---- helpers (atsc/src/utils/mod.rs, atsc/src/utils/error.rs, optimizer) ---- */
size_t orc_next_size(size_t n);                 /* utils/mod.rs:32-38 */
int orc_is_decomposable(size_t n);              /* utils/mod.rs:41-49 */
size_t orc_prev_power_of_two(size_t n);         /* utils/mod.rs:24-29 */
double orc_round_f64(double x, uint32_t d);     /* utils/mod.rs:61-64 */
double orc_round_and_limit_f64(double x, double mn, double mx, uint32_t d); /* :66-74 */
double orc_error_mape(const double *orig, const double *gen, size_t n); /* utils/error.rs:104-116 */
void orc_stats_new(const double *x, size_t n, orc_stats *out); /* optimizer/utils.rs:39-89 */
void orc_split_n(double x, int64_t *ip, double *frac);         /* optimizer/utils.rs:115-160 */
/* optimizer/mod.rs:78-98; returns number of chunks written (<= cap) */
size_t orc_chunk_sizes(size_t len, size_t *out, size_t cap);
/* optimizer/mod.rs:64-71; out must hold n doubles; returns kept count */
size_t orc_clean_data(const double *x, size_t n, double *out);
/* fft.rs:184-204; out must hold orc_next_size(n) doubles */
size_t orc_gibbs_sizing(const double *x, size_t n, double *out);

/* ---- single codecs: unbounded entry points (compressor/mod.rs:63-74) ---- */
/* Each returns a malloc'd payload in *out. */
int orc_noop(const double *x, size_t n, orc_buf *out);               /* noop.rs:72-77 */
int orc_constant(const double *x, size_t n, orc_buf *out);           /* constant.rs:135-139 */
int orc_rle(const double *x, size_t n, orc_buf *out);                /* rle.rs:239-243 */
int orc_fft(const double *x, size_t n, orc_buf *out);                /* fft.rs:466-484 */
int orc_fft_set(const double *x, size_t n, size_t freqs, orc_buf *out); /* fft.rs:526-544 */
int orc_polynomial(const double *x, size_t n, int idw, orc_buf *out);   /* polynomial.rs:407-413 */

/* ---- bounded entry points (compressor/mod.rs:94-107) ---- */
/* err receives CompressorResult.error; iterations (may be NULL) the ladder trips */
int orc_fft_allowed_error(const double *x, size_t n, double max_err, orc_buf *out,
                          double *err, int *iterations);             /* fft.rs:494-512 */
int orc_polynomial_allowed_error(const double *x, size_t n, double max_err, int idw,
                                 orc_buf *out, double *err, int *iterations); /* polynomial.rs:415-425 */

/* Compressor::compress / compress_bounded / get_compress_bounded_results.
 * bounded=0 -> compress (mod.rs:63-74), bounded=1 -> get_compress_bounded_results
 * (mod.rs:94-107; compress_bounded :76-92 yields the same bytes). */
int orc_compress(int compressor, const double *x, size_t n, int bounded, double max_err,
                 orc_buf *out, double *err);
/* Compressor::decompress -- compressor/mod.rs:109-119.  out: malloc'd doubles, *out_n count */
int orc_decompress(int compressor, const uint8_t *data, size_t len, size_t samples,
                   double **out, size_t *out_n);

/* CompressorFrame::compress_best -- frame/mod.rs:71-149 */
int orc_compress_best(const double *x, size_t n, float max_error, int level,
                      orc_buf *out, int *chosen, double *err);

/* ---- stream level (data.rs, header.rs, main.rs) ---- */
/* CompressedStream built from chunks with compress_chunk_with (bounded=0, data.rs:47-53)
 * or compress_chunk_bounded_with (bounded=1, data.rs:56-76), then to_bytes (data.rs:79-85).
 * chunk_off has n_chunks+1 prefix offsets into x. */
int orc_stream_compress(const double *x, const uint64_t *chunk_off, size_t n_chunks,
                        int compressor, int bounded, float max_error, int level,
                        orc_buf *out, uint8_t *chosen /* n_chunks or NULL */,
                        double *errs /* n_chunks or NULL */);
/* main.rs:130-165 compress_data: plan (clean + chunk) + stream; cli_error is the u8 -e value */
int orc_compress_data(const double *x, size_t n, int compressor, uint8_t cli_error, int level,
                      orc_buf *out);
/* main.rs:168-172 decompress_data (from_bytes + decompress).  returns <0 on bad magic/version */
/* test support: pop order of the reference's BinaryHeap over bins with these f32 norms (n <= 65536) */
int orc_heap_order(const float *norms, size_t n, size_t k, uint32_t *out);
int orc_decompress_data(const uint8_t *bro, size_t len, double **out, size_t *out_n);

#ifdef __cplusplus
}
#endif
#endif
