// csv-compressor -- command line front end over libatsc_hip.so with the reference's flags, file
// naming and exit behaviour (csv-compressor/src/main.rs:31-232): `timestamp,value` CSV in,
// .bro (+ .vsri index, + .wavbro samples) out; `-u` turns .bro + .vsri back into .wbro + .csv.
// Compression and decompression run on the GPU; the index and the text formats are host code.
//
//   csv-compressor [-o OUT] [-u] [--no-compression] [--output-vsri] [--output-wavbrro] [--output-csv]
//                  [--compressor auto|noop|fft|constant|polynomial|idw] [-e 0..50] [-c 0..6] <INPUT>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/atsc_hip.h"

namespace {

struct Args {
    std::string input, output;
    bool has_output = false, uncompress = false, no_compression = false;
    bool output_vsri = false, output_wavbrro = false, output_csv = false;
    int compressor = ATSC_AUTO;  // default_value = "auto" (main.rs:66)
    int error = 5;               // default_value_t = 5 (main.rs:73)
    int level = 0;
};

constexpr int PANIC = 101;  // exit status of a Rust panic: every failure below is an expect()/panic!()

void usage()
{
    fprintf(stderr,
            "A Time-Series compressor utilizes Brro Compressor for CSV format\n\n"
            "Usage: csv-compressor [OPTIONS] <INPUT>\n\nOptions:\n"
            "  -o, --output <OUTPUT>          where the result will be stored\n"
            "  -u                             uncompress the input\n"
            "      --no-compression           do not write the .bro\n"
            "      --output-vsri              write the generated VSRI index\n"
            "      --output-wavbrro           write the generated WavBrro\n"
            "      --output-csv               (accepted; the reference never reads it)\n"
            "      --compressor <COMPRESSOR>  auto, noop, fft, constant, polynomial, idw [default: auto]\n"
            "  -e, --error <ERROR>            maximum allowed error in %% (0..50) [default: 5]\n"
            "  -c, --compression-selection-sample-level <0..6>  [default: 0]\n"
            "  -h, --help    -V, --version\n");
}

bool parse_compressor(const std::string &v, int &out)  // main.rs:85-94: no rle here
{
    static const struct { const char *n; int id; } T[] = {
        {"auto", ATSC_AUTO}, {"noop", ATSC_NOOP}, {"fft", ATSC_FFT}, {"constant", ATSC_CONSTANT},
        {"polynomial", ATSC_POLYNOMIAL}, {"idw", ATSC_IDW}};
    for (auto &t : T)
        if (v == t.n) { out = t.id; return true; }
    return false;
}
bool parse_int(const std::string &v, int lo, int hi, int &out)
{
    if (v.empty()) return false;
    char *end = nullptr;
    long x = strtol(v.c_str(), &end, 10);
    if (*end || x < lo || x > hi) return false;
    out = (int)x;
    return true;
}

std::string with_ext(const std::string &path, const char *ext)  // PathBuf::set_extension
{
    const size_t slash = path.find_last_of('/');
    const size_t dot = path.find_last_of('.');
    std::string base = (dot != std::string::npos && (slash == std::string::npos || dot > slash + 1)) ? path.substr(0, dot) : path;
    return base + "." + ext;
}

int die(const char *what, int rc = 0, const char *detail = "")
{
    fprintf(stderr, "thread 'main' panicked: %s%s%s%s\n", what, rc ? ": " : "", rc ? atsc_strerror(rc) : "", detail);
    return PANIC;
}

int uncompress(const Args &a, const std::string &output_base)  // main.rs:139-173
{
    uint8_t *bro = nullptr;
    uint64_t len = 0;
    int rc = atsc_bro_read_file(a.input.c_str(), &bro, &len);
    if (rc) return die("failed to read bro file", rc);
    if (!bro) return 0;  // not a BRO file: nothing happens
    atsc_ctx *ctx = nullptr;
    rc = atsc_ctx_create(&ctx, 0);
    if (rc) { atsc_free(bro); return die("no GPU context", rc); }
    double *data = nullptr;
    uint64_t n = 0;
    rc = atsc_decompress_data(ctx, bro, len, &data, &n);
    atsc_free(bro);
    if (rc) { int e = die("decompress", rc, atsc_ctx_last_error(ctx)); atsc_ctx_destroy(ctx); return e; }
    atsc_ctx_destroy(ctx);
    atsc_vsri *index = nullptr;
    rc = atsc_vsri_load(with_ext(a.input, "vsri").c_str(), &index);
    if (rc) { atsc_free(data); return die("failed to read vsri", rc); }
    const std::string wbro_path = with_ext(output_base, "wbro");
    rc = atsc_wbro_write(wbro_path.c_str(), data, n);
    if (rc) { atsc_free(data); atsc_vsri_free(index); return die("writing wavbrro", rc); }
    std::vector<int64_t> ts(n ? n : 1);
    rc = atsc_metric_sample_times(index, n, ts.data());  // Metric::get_samples: get_time(i).unwrap()
    atsc_vsri_free(index);
    if (rc) { atsc_free(data); return die("called `Option::unwrap()` on a `None` value (index has no time for a sample)"); }
    rc = atsc_samples_csv_write(with_ext(wbro_path, "csv").c_str(), ts.data(), data, n);
    atsc_free(data);
    if (rc) return die("failed to write samples to file", rc);
    return 0;
}

int compress(const Args &a, const std::string &output_base)  // main.rs:174-207
{
    int64_t *ts = nullptr;
    double *vals = nullptr;
    uint64_t n = 0;
    int rc = atsc_samples_csv_read(a.input.c_str(), &ts, &vals, &n);
    if (rc) return die("failed to read samples from file", rc);
    atsc_vsri *index = atsc_vsri_new();
    if (!index) { atsc_free(ts); atsc_free(vals); return die("out of memory"); }
    uint64_t bad = 0;
    rc = atsc_metric_index_samples(index, ts, n, &bad);
    atsc_free(ts);
    if (rc) {
        fprintf(stderr, "updating for point failed, sample: %llu\n", (unsigned long long)bad);
        atsc_free(vals);
        atsc_vsri_free(index);
        return die("failed to create metric from samples");
    }
    int status = 0;
    if (a.output_wavbrro && atsc_wbro_write(with_ext(output_base, "wavbro").c_str(), vals, n)) status = die("writing wavbrro");
    if (!status && a.output_vsri && (rc = atsc_vsri_flush_to(index, with_ext(output_base, "vsri").c_str())))
        status = die("failed to flush vsri to the file", rc);
    atsc_vsri_free(index);
    if (!status && !a.no_compression) {
        atsc_ctx *ctx = nullptr;
        rc = atsc_ctx_create(&ctx, 0);
        if (rc) { atsc_free(vals); return die("no GPU context", rc); }
        uint8_t *bro = nullptr;
        uint64_t len = 0;
        rc = atsc_compress_data(ctx, vals, n, a.compressor, (uint8_t)a.error, a.level, &bro, &len);
        if (rc) {
            status = die("compress", rc, atsc_ctx_last_error(ctx));
        } else {
            FILE *f = fopen(with_ext(output_base, "bro").c_str(), "wb");
            if (!f || fwrite(bro, 1, len, f) != len) status = die("failed to write compressed data");
            if (f) fclose(f);
            atsc_free(bro);
        }
        atsc_ctx_destroy(ctx);
    }
    atsc_free(vals);
    return status;
}

}  // namespace

int main(int argc, char **argv)
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i], v;
        auto value = [&](const char *name) -> bool {
            const std::string pre = std::string(name) + "=";
            if (s.rfind(pre, 0) == 0) { v = s.substr(pre.size()); return true; }
            if (s == name && i + 1 < argc) { v = argv[++i]; return true; }
            return false;
        };
        if (s == "-h" || s == "--help") { usage(); return 0; }
        if (s == "-V" || s == "--version") { printf("csv-compressor 0.7.2 (%s)\n", atsc_version()); return 0; }
        if (s == "-u") a.uncompress = true;
        else if (s == "--no-compression") a.no_compression = true;
        else if (s == "--output-vsri") a.output_vsri = true;
        else if (s == "--output-wavbrro") a.output_wavbrro = true;
        else if (s == "--output-csv") a.output_csv = true;
        else if (value("--output") || value("-o")) { a.output = v; a.has_output = true; }
        else if (value("--compressor")) { if (!parse_compressor(v, a.compressor)) { fprintf(stderr, "error: invalid value '%s' for '--compressor'\n", v.c_str()); return 2; } }
        else if (value("--error") || value("-e")) { if (!parse_int(v, 0, 50, a.error)) { fprintf(stderr, "error: invalid value '%s' for '--error': not in 0..=50\n", v.c_str()); return 2; } }
        else if (value("--compression-selection-sample-level") || value("-c")) { if (!parse_int(v, 0, 6, a.level)) { fprintf(stderr, "error: invalid value '%s' for '-c': not in 0..=6\n", v.c_str()); return 2; } }
        else if (!s.empty() && s[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", s.c_str()); usage(); return 2; }
        else a.input = s;
    }
    if (a.input.empty()) { usage(); return 2; }
    struct stat st;
    if (stat(a.input.c_str(), &st) != 0) return die("Failed to retrieve metadata of the input");  // main.rs:226-229
    if (!S_ISREG(st.st_mode)) return die("Input is not a file");                                 // main.rs:219-221
    const std::string output_base = a.has_output ? a.output : a.input;                            // main.rs:133-137
    return a.uncompress ? uncompress(a, output_base) : compress(a, output_base);
}
