// atsc -- command line front end over libatsc_hip.so with the reference's flags and file naming
// (atsc/src/main.rs:29-127,176-243).  Every frame is compressed / decompressed on the GPU.
//
//   atsc [--compressor auto|noop|fft|constant|polynomial|idw|rle] [-e 0..50] [-u]
//        [-c 0..6] [--verbose] [--csv] [--no-header] [--fields=TIME,VALUE] <file-or-directory>
#include <dirent.h>
#include <sys/stat.h>

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/atsc_hip.h"

namespace {

struct Args {
    std::string input;
    int compressor = ATSC_AUTO;  // default_value = "auto" (main.rs:180)
    int error = 3;               // default_value_t = 3 (main.rs:187)
    bool uncompress = false;
    int level = 0;
    bool verbose = false, csv = false, no_header = false;
    std::string fields = "time,value";  // main.rs:218
};

void usage()
{
    fprintf(stderr,
            "A Time-Series compressor\n\nUsage: atsc [OPTIONS] <INPUT>\n\nOptions:\n"
            "      --compressor <COMPRESSOR>  auto, noop, fft, constant, polynomial, idw, rle [default: auto]\n"
            "  -e, --error <ERROR>            maximum allowed error in %% (0..50) [default: 3]\n"
            "  -u                             uncompress the input file/directory\n"
            "  -c, --compression-selection-sample-level <0..6>  [default: 0]\n"
            "      --verbose                  dump every sample\n"
            "      --csv                      input is a CSV file\n"
            "      --no-header                the CSV has no header\n"
            "      --fields <TIME,VALUE>      CSV field names [default: time,value]\n"
            "  -h, --help    -V, --version\n");
}

bool parse_compressor(const std::string &v, int &out)
{
    static const struct { const char *n; int id; } T[] = {
        {"auto", ATSC_AUTO}, {"noop", ATSC_NOOP}, {"fft", ATSC_FFT}, {"constant", ATSC_CONSTANT},
        {"polynomial", ATSC_POLYNOMIAL}, {"idw", ATSC_IDW}, {"rle", ATSC_RLE}};
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

// Rust `{:?}` of an f64: shortest round-trip digits, ".0" appended to integers
std::string debug_f64(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    std::string s(buf, r.ptr);
    if (s.find('e') != std::string::npos) {
        // Rust prints 1e16 as 1e16 and small/large values in exponent form as well
        return s;
    }
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}
void dump(const char *tag, const double *d, uint64_t n)
{
    printf("%s=[", tag);
    for (uint64_t i = 0; i < n; ++i) printf("%s%s", i ? ", " : "", debug_f64(d[i]).c_str());
    printf("]\n");
}

std::string with_ext(const std::string &path, const char *ext)  // PathBuf::set_extension
{
    const size_t slash = path.find_last_of('/');
    const size_t dot = path.find_last_of('.');
    std::string base = (dot != std::string::npos && (slash == std::string::npos || dot > slash + 1)) ? path.substr(0, dot) : path;
    return base + "." + ext;
}

int process_single_file(atsc_ctx *ctx, const std::string &path, const Args &a)
{
    if (a.uncompress) {  // main.rs:72-83
        uint8_t *bro = nullptr;
        uint64_t len = 0;
        int rc = atsc_bro_read_file(path.c_str(), &bro, &len);
        if (rc) return rc;
        if (!bro) return ATSC_OK;  // not a BRO file: skipped silently
        double *out = nullptr;
        uint64_t n = 0;
        rc = atsc_decompress_data(ctx, bro, len, &out, &n);
        atsc_free(bro);
        if (rc) return rc;
        if (a.verbose) dump("Output", out, n);
        rc = atsc_wbro_write(with_ext(path, "wbro").c_str(), out, n);
        atsc_free(out);
        return rc;
    }
    double *data = nullptr;
    uint64_t n = 0;
    int rc;
    if (a.csv) {  // main.rs:84-100
        const size_t comma = a.fields.find(',');
        const std::string tf = a.fields.substr(0, comma);
        const std::string vf = comma == std::string::npos ? std::string() : a.fields.substr(comma + 1);
        rc = atsc_csv_read(path.c_str(), a.no_header ? 0 : 1, tf.c_str(), vf.c_str(), &data, &n);
    } else {
        rc = atsc_wbro_read(path.c_str(), &data, &n);  // main.rs:112
    }
    if (rc) return rc;
    if (a.verbose) dump("Input", data, n);
    uint8_t *bro = nullptr;
    uint64_t len = 0;
    rc = atsc_compress_data(ctx, data, n, a.compressor, (uint8_t)a.error, a.level, &bro, &len);
    atsc_free(data);
    if (rc) return rc;
    FILE *f = fopen(with_ext(path, "bro").c_str(), "wb");  // main.rs:121-124
    if (!f) { atsc_free(bro); return ATSC_E_IO; }
    const size_t w = fwrite(bro, 1, len, f);
    fclose(f);
    atsc_free(bro);
    return w == len ? ATSC_OK : ATSC_E_IO;
}

bool has_ext(const std::string &p, const char *ext)
{
    const size_t dot = p.find_last_of('.');
    return dot != std::string::npos && p.substr(dot + 1) == ext;
}

// main.rs:50-68 walks read_dir while it writes into the same directory and calls
// process_single_file twice per entry; here the listing is taken once and every input file
// (.wbro / .csv when compressing, anything BRO-tagged when uncompressing) is handled once.
int process_directory(atsc_ctx *ctx, const Args &a)
{
    std::vector<std::string> files;
    DIR *d = opendir(a.input.c_str());
    if (!d) return ATSC_E_IO;
    while (dirent *e = readdir(d)) {
        std::string p = a.input + "/" + e->d_name;
        struct stat st;
        if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode)) files.push_back(p);
    }
    closedir(d);
    int last = ATSC_OK;
    for (const std::string &p : files) {
        if (!a.uncompress && !(a.csv ? has_ext(p, "csv") : has_ext(p, "wbro"))) continue;
        int rc = process_single_file(ctx, p, a);
        if (rc) {
            fprintf(stderr, "[ERROR] %s File: %s\n", atsc_strerror(rc), p.c_str());
            last = rc;
        }
    }
    return last;
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
        if (s == "-V" || s == "--version") { printf("atsc 0.7.2 (%s)\n", atsc_version()); return 0; }
        if (s == "-u") a.uncompress = true;
        else if (s == "--verbose") a.verbose = true;
        else if (s == "--csv") a.csv = true;
        else if (s == "--no-header") a.no_header = true;
        else if (value("--compressor")) { if (!parse_compressor(v, a.compressor)) { fprintf(stderr, "error: invalid value '%s' for '--compressor'\n", v.c_str()); return 2; } }
        else if (value("--error") || value("-e")) { if (!parse_int(v, 0, 50, a.error)) { fprintf(stderr, "error: invalid value '%s' for '--error': not in 0..=50\n", v.c_str()); return 2; } }
        else if (value("--compression-selection-sample-level") || value("-c")) { if (!parse_int(v, 0, 6, a.level)) { fprintf(stderr, "error: invalid value '%s' for '-c': not in 0..=6\n", v.c_str()); return 2; } }
        else if (value("--fields")) a.fields = v;
        else if (!s.empty() && s[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", s.c_str()); usage(); return 2; }
        else a.input = s;
    }
    if (a.input.empty()) { usage(); return 2; }
    struct stat st;
    if (stat(a.input.c_str(), &st) != 0) { fprintf(stderr, "[ERROR] %s: No such file or directory\n", a.input.c_str()); return 1; }
    atsc_ctx *ctx = nullptr;
    int rc = atsc_ctx_create(&ctx, 0);
    if (rc) { fprintf(stderr, "[ERROR] %s\n", atsc_strerror(rc)); return 1; }
    if (S_ISREG(st.st_mode)) rc = process_single_file(ctx, a.input, a);
    else if (S_ISDIR(st.st_mode)) rc = process_directory(ctx, a);
    else { fprintf(stderr, "[ERROR] The provided path is neither a file nor a directory.\n"); rc = ATSC_E_IO; }
    if (rc) fprintf(stderr, "[ERROR] %s (%s)\n", atsc_strerror(rc), atsc_ctx_last_error(ctx));
    atsc_ctx_destroy(ctx);
    return rc ? 1 : 0;  // main.rs:239-242
}
