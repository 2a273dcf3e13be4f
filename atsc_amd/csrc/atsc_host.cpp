// atsc_host.cpp -- host side of libatsc_hip.so: context, plans, launch orchestration and the
// format helpers that sit either side of the GPU path.  There is NO CPU compression path in
// this library: without a HIP device atsc_ctx_create fails with ATSC_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "../../include/atsc_hip.h"
#include "atsc_internal.h"

namespace atsc {
hipError_t launch_compress_class(int cls, uint32_t count, uint32_t lds, const double *samples,
                                 const DevFrame *frames, const uint32_t *ids, const DevPlan *plans,
                                 const float2 *twpool, const KParams &prm, uint8_t *slots,
                                 DevResult *res, atsc_frame_diag *diag, const UniArgs &uni, hipStream_t s,
                                 hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t launch_compress_large(uint32_t count, const double *samples, const DevFrame *frames,
                                 const uint32_t *ids, const DevPlan *plans, const float2 *twpool,
                                 const KParams &prm, uint8_t *slots, DevResult *res, atsc_frame_diag *diag,
                                 unsigned char *ws, uint64_t ws_stride, uint32_t ws_slots, hipStream_t s,
                                 const LargePre *pre = nullptr);
uint64_t large_ws_bytes(uint32_t n, uint32_t L, uint32_t kcap);
hipError_t launch_decompress_large(uint32_t count, const struct DevDFrame *frames, const uint32_t *ids,
                                   const DevPlan *plans, const float2 *twpool, const uint8_t *body,
                                   double *out, int *status, unsigned char *ws, uint64_t ws_stride,
                                   uint32_t ws_slots, int tiled, int sparse, hipStream_t s, const LargePre *pre = nullptr,
                                   uint32_t sp_tiles = 0);
hipError_t launch_order_by_cost(const uint32_t *ids_src, uint32_t *ids_dst, const uint32_t *cost, uint8_t *bkt,
                                uint32_t *hist_cursor, const uint32_t *class_first,
                                const uint32_t *class_count, int n_classes, hipStream_t s);
hipError_t launch_pack(const DevFrame *frames, const DevResult *res, uint64_t n_frames,
                       uint32_t *local, uint64_t *blocksum, const uint8_t *slots, uint8_t *body,
                       uint64_t body_cap, uint64_t *rec_off, uint8_t *chosen, double *err,
                       const uint32_t *big_ids, uint32_t n_big, hipStream_t s, uint64_t *chain = nullptr);
hipError_t launch_nonfinite_flag(const double *x, uint64_t n, uint32_t *flag, hipStream_t s);
hipError_t launch_copy_words(void *dst, const void *src_host_mapped, uint64_t n_words, hipStream_t s);
hipError_t launch_decompress(const struct DevDFrame *frames, uint64_t n_frames, const uint32_t *ids,
                             int cls, uint32_t count, uint32_t lds, const DevPlan *plans,
                             const float2 *twpool, const uint8_t *body, double *out, int *status,
                             hipStream_t s);
}  // namespace atsc

using namespace atsc;

static const int N_CLASSES = 7;       // 0..5: LDS-resident frame kernels, 6: large frames (atsc_large.hip)
static const int CLASS_LARGE = 6;
static const uint32_t MAX_FRAME_TIER_M = 4096;   // longest frame of the LDS-resident kernels
static const uint32_t MAX_FRAME = 131072;        // MAX_FRAME_SIZE of the reference chunker (optimizer/mod.rs:27)
static const uint32_t LARGE_WS_SLOTS = 256;      // large frames in flight (one workgroup + workspace each) ...
// ... of the longest kind; a batch of shorter large frames gets as many slots as the same memory holds (a launch of 256
// 8192-sample frames is 2 M samples: every grid of the large tier would be latency-bound on it)
static uint32_t large_ws_slots(uint64_t ws_stride)
{
    uint64_t budget = 1600ull << 20;
    if (const char *e = getenv("ATSC_LARGE_WS_MB")) budget = (uint64_t)std::max(1, atoi(e)) << 20;
    const uint64_t fit = ws_stride ? budget / ws_stride : LARGE_WS_SLOTS;
    return (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(LARGE_WS_SLOTS, fit));
}

struct atsc_ctx {
    int device = 0;
    std::string last_error;
    // diagnostics of the last compress call
    atsc_frame_diag *d_diag = nullptr;
    uint64_t diag_cap = 0;
    uint64_t diag_n = 0;
    hipStream_t diag_stream = nullptr;
    bool want_diag = false;
    // Device memory pool.  The host-pointer entry points build a plan and five buffers per call and
    // drop them at the end; hipMalloc / hipFree of hundreds of megabytes cost milliseconds each, so
    // freed blocks are kept (up to POOL_MAX_BYTES) and handed out again when the size fits.
    std::vector<std::pair<void *, size_t>> pool_free_list;
    std::map<void *, size_t> pool_live;
    size_t pool_held = 0;
    // Streams of the context's own (created on first use; few, because the runtime maps streams onto a handful of
    // hardware queues).  Two uses:
    //  * pipelined calls (atsc_compress_plan_dev_pipelined): consecutive batches go round-robin over the chains of a
    //    plan, chain c on chain_streams[c] -- a dependent launch starts 6-10 us after its predecessor ends on this system
    //    (tools/gap_probe.hip), and a frame kernel's freed wave slots refill slowly from a single queue; several queues
    //    feeding the same CUs hide both (what bench.py --chains did from outside in round 2);
    //  * the large tier: see large groups below.
    // resident launches (k_compress_resident): one set of frame counters per stream that has carried one (launches
    // on a stream are sequential, and a launch leaves its counters zero)
    uint32_t *d_queues = nullptr;
    hipStream_t q_stream[16] = {};
    uint32_t q_used = 0;
    hipStream_t chain_streams[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t pack_streams[4] = {nullptr, nullptr, nullptr, nullptr};  // a chain's packing: beside its next batch's codecs
    int n_chains = 2;                   // atsc_ctx_set_chains / ATSC_CHAINS (1..4)
    // The large tier's kernel chain is bound by latency, not by throughput (a chain of launches, several of them one
    // workgroup per frame): the large frames of a call are dealt over LARGE_GROUPS contiguous groups, each on a stream
    // of the context's own, forked from and joined to the caller's stream by events -- the groups' chains overlap.
    // (the groups run on chain_streams[])
    bool adaptive_order = false;        // pipelined calls start a class's costliest frames first (atsc_ctx_set_adaptive_order)
    int debug_stop = 0;  // ATSC_DEBUG_STOP: phase-timing aid for tools/, never set in production
    // optional timing of the dominant k_compress launch (HIP events on the launch stream)
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    // Host-pointer entry points (atsc_compress_frames ...): plans kept by frame layout (a service compresses
    // the same layout batch after batch; building a plan walks every frame and uploads its tables), and a
    // stream of their own so that the blocking host-to-device copy of one part of a batch does not order
    // itself behind the kernels of the part before it (the legacy default stream would).
    struct CachedPlan {
        uint64_t hash = 0, stamp = 0;
        std::vector<uint32_t> lens;
        atsc_plan *plan = nullptr;
    };
    std::vector<CachedPlan> plan_cache;
    uint64_t plan_stamp = 0;
    hipStream_t work_stream = nullptr;
    hipStream_t copy_stream = nullptr;             // host-to-device copies of the host-pointer entry points
    hipStream_t d2h_stream = nullptr;              // ... and the records' way back, part by part (registered memory)
    std::vector<hipEvent_t> ev_parts;              // "part g's records are packed"
    unsigned char *h_stage = nullptr;              // page-locked staging for tables a kernel copies up (h2d_small)
    size_t h_stage_cap = 0, h_stage_used = 0;
    hipEvent_t ev_copy[2] = {nullptr, nullptr};    // "part g's samples are on the device"
};

struct PlanTables {
    std::vector<DevPlan> plans;   // host copy
    std::vector<float2> twpool;   // host copy
    std::map<uint32_t, uint32_t> by_n;
    DevPlan *d_plans = nullptr;
    float2 *d_tw = nullptr;
};

// A second view of the same batch with other per-frame transform parameters:
//  * trial launch of the sample-level selector: the first COMPRESSION_SPEED[level] samples of every
//    frame that is at least that long (frame/mod.rs:89-111)
//  * FFT::compress (unbounded): no Gibbs padding, transform length = frame length (fft.rs:366-388)
struct SubPlan {
    bool large_tiled = false;
    PlanTables tabs;
    DevFrame *d_frames = nullptr;
    uint32_t *d_ids = nullptr;
    DevResult *d_res = nullptr;
    std::vector<uint32_t> class_count, class_lds, class_first;
    uint32_t count = 0, min_n = 0;
};

struct atsc_plan {
    atsc_ctx *ctx = nullptr;
    uint64_t n_frames = 0, n_samples = 0, body_bound = 0, slot_bytes = 0;
    PlanTables tabs;
    std::vector<uint32_t> class_count, class_lds, class_first;
    std::vector<UniArgs> class_uni;  // per class: by-value launch arguments when the class is uniform
    std::vector<DevFrame> h_frames;  // host copy (trial plans are derived from it)
    mutable SubPlan *trials[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    mutable SubPlan *nopad = nullptr;
    DevFrame *d_frames = nullptr;
    uint32_t *d_ids = nullptr;
    DevResult *d_res = nullptr;
    uint8_t *d_slots = nullptr;
    uint32_t *d_local = nullptr;
    uint64_t *d_blocksum = nullptr;
    unsigned char *d_ws = nullptr;   // workspace of the large-frame kernel
    uint64_t ws_stride = 0;
    uint32_t ws_slots = 0;
    bool large_tiled = false;        // form of the large tier's in-kernel transforms
    LargePre large_pre{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // batched pre-pass of the large tier (tiles1 == 0: off)
    // atsc_compress_plan_dev_pipelined: consecutive calls go round-robin over up to four chains.  A chain is a stream
    // of the context's (atsc_ctx::chain_streams) plus everything a batch in flight owns: scratch set (payload slots,
    // results, scan scratch), large-tier workspace, cost records and the launch order derived from them.  Chain 0's
    // scratch set is the plan's own (plain calls use it).
    struct Scratch {
        DevResult *d_res = nullptr;
        uint8_t *d_slots = nullptr;
        uint32_t *d_local = nullptr;
        uint64_t *d_blocksum = nullptr;
    };
    struct Chain {
        Scratch S;
        unsigned char *d_ws = nullptr;      // large-tier workspace set
        hipEvent_t ev_fork = nullptr;       // the caller's stream at the call
        hipEvent_t ev_lfork = nullptr;      // large tier: this call's stream where the groups fork
        hipEvent_t ev_codec = nullptr;      // every kernel that reads d_samples is done (atsc_plan_input_release)
        hipEvent_t ev_done = nullptr;       // records packed (atsc_plan_join)
        hipEvent_t ev_group[4] = {nullptr, nullptr, nullptr, nullptr};  // large tier: group chains done
        bool pending = false;               // ev_done recorded and not yet known to have passed
        // scheduling hint: clocks per frame in this chain's last batch and the launch order derived from them
        uint32_t *d_cost = nullptr;
        uint8_t *d_bucket = nullptr;
        uint32_t *d_hist = nullptr;         // histogram + cursors of k_cost_hist / k_cost_scatter
        uint32_t *d_ids_adapt = nullptr;
        bool adapt_valid = false;
        bool ready = false;
    };
    mutable Chain chains[8];  // two sets per chain stream: set q runs on stream q % (number of chains)
    mutable uint32_t turn = 0;
    mutable bool single_set = false;  // memory allowed no second scratch set: pipelined calls reuse set 0
    mutable uint32_t chain_cap = 0;  // most chains this plan's pipelined calls use (0: not decided yet; plan_chains)
    uint64_t slots_bytes = 0;
};

struct atsc_dplan {
    atsc_ctx *ctx = nullptr;
    uint64_t n_frames = 0, n_samples = 0;
    PlanTables tabs;
    std::vector<uint32_t> class_count, class_lds, class_first;
    DevDFrame *d_frames = nullptr;
    uint32_t *d_ids = nullptr;
    int *d_status = nullptr;
    unsigned char *d_ws = nullptr;
    uint64_t ws_stride = 0;
    uint32_t ws_slots = 0;
    bool large_tiled = false;
    LargePre large_pre{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // batched inverse transform of the large FFT frames (tiles1 == 0: off)
    uint32_t large_sp_tiles = 0;        // tiles per frame of the sparse inverse's (tile, frame) grid (0: off)
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
static int fail(atsc_ctx *ctx, int rc, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return rc;
}
#define HIPCHK(ctx, call)                                                    \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return fail((ctx), ATSC_E_HIP, #call, e__);   \
    } while (0)

static const size_t POOL_MAX_BYTES = 8ull << 30;
static hipError_t pool_alloc(atsc_ctx *ctx, void **out, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    // smallest kept block that fits without wasting more than half of itself (or 4 MB)
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->pool_free_list.size(); ++i) {
        const size_t sz = ctx->pool_free_list[i].second;
        if (sz >= bytes && sz <= std::max(2 * bytes, bytes + (4u << 20)) &&
            (best == (size_t)-1 || sz < ctx->pool_free_list[best].second))
            best = i;
    }
    if (best != (size_t)-1) {
        *out = ctx->pool_free_list[best].first;
        ctx->pool_live[*out] = ctx->pool_free_list[best].second;
        ctx->pool_held -= ctx->pool_free_list[best].second;
        ctx->pool_free_list.erase(ctx->pool_free_list.begin() + (long)best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !ctx->pool_free_list.empty()) {  // give the kept blocks back and retry
        for (auto &b : ctx->pool_free_list) (void)hipFree(b.first);
        ctx->pool_free_list.clear();
        ctx->pool_held = 0;
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) ctx->pool_live[*out] = bytes;
    return e;
}
// The caller guarantees that no kernel still uses the block (plan destruction synchronises the device
// once, as hipFree would for every block; the host-pointer entry points have synchronised already).
static void pool_free(atsc_ctx *ctx, void *p)
{
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) { (void)hipFree(p); return; }
    const size_t sz = it->second;
    ctx->pool_live.erase(it);
    if (ctx->pool_held + sz > POOL_MAX_BYTES || ctx->pool_free_list.size() >= 256) { (void)hipFree(p); return; }
    ctx->pool_free_list.emplace_back(p, sz);
    ctx->pool_held += sz;
}

static bool is_decomposable(uint64_t n)
{
    if (n == 0) return false;
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    return n == 1;
}
static uint32_t pow2_ge(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
static uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

extern "C" uint64_t atsc_next_size(uint64_t n)
{
    n += 1;
    while (!is_decomposable(n)) n += 1;
    return n;
}

extern "C" uint64_t atsc_payload_bound_bytes(uint64_t n)
{
    // worst case over codecs: RLE with every value distinct  2 + varint(D) + n*(8 + 1 + varint(idx));
    // Noop 1 + varint(n) + 9n ; Polynomial store-all 2 + varint(n) + 8n + 17
    // (run indices >= 65536 take 5 varint bytes)
    return 32 + (n > 65535 ? 17 : 14) * n;
}

// Form of the transforms that run inside the large-frame kernels: two LDS-tiled passes when the
// batch has more large frames than the GPU has CUs to give them (fewer bytes through L2: 15.7 vs
// 14.3 Gsamples/s at 256+ frames), stage by stage when every frame has a CU to itself and latency is
// what counts (6.8 vs 6.0 Gsamples/s at 80 frames).  ATSC_LARGE_FFT=tiled|stages overrides.
static bool choose_large_tiled(uint32_t n_large_frames)
{
    bool tiled = n_large_frames > 128;
    if (const char *e = getenv("ATSC_LARGE_FFT")) tiled = strcmp(e, "tiled") == 0;
    return tiled;
}
// Inverse transforms of the large tier run from the sparse list of admitted bins (sparse_inverse,
// atsc_large.hip); ATSC_LARGE_DENSE=1 keeps the dense transforms through the workspace (A/B runs).
static bool large_sparse() { return getenv("ATSC_LARGE_DENSE") == nullptr; }
// Grid extents of the batched pre-pass (forward transform, untangle, norms of every large frame over
// the whole GPU before the per-frame kernel); all zero when a large frame length has no M1 x M2 split.
static LargePre large_pre_extents(const std::vector<DevPlan> &plans, const std::vector<uint32_t> &large_plan_ids)
{
    LargePre pre{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool first_large = true, rows_ok = true;
    if (getenv("ATSC_LARGE_NO_PREPASS")) return pre;
    pre.cols243 = getenv("ATSC_LARGE_OLD_COLS") ? 0u : 1u;
    for (uint32_t pi : large_plan_ids) {
        const DevPlan &p = plans[pi];
        if (!p.f4_m1) return LargePre{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        pre.tiles1 = std::max(pre.tiles1, (p.f4_m2 + 15) / 16);
        pre.tiles2 = std::max(pre.tiles2, (p.f4_m1 + 15) / 16);
        pre.chunks = std::max(pre.chunks, (p.bins + 255) / 256);
        pre.m1_max = std::max(pre.m1_max, p.f4_m1);
        pre.m2_max = std::max(pre.m2_max, p.f4_m2);
        if (p.sp_mf) pre.sp_tiles = std::max(pre.sp_tiles, (p.sp_md + 7) / 8);
        pre.chunks_n = std::max(pre.chunks_n, (p.n + 4095) / 4096);
        if (p.f4_m1 != 243) pre.cols243 = 0;
        // rows9p: the set of P (as a bit mask) over the large frames when EVERY one of them has M = 243 x 9 P, else 0
        uint32_t rp = 0;
        if (p.f4_m1 == 243 && p.f4_m2 % 9 == 0 && !getenv("ATSC_LARGE_OLD_ROWS")) {
            const uint32_t P9 = p.f4_m2 / 9;
            if (P9 >= 2 && P9 <= 32 && (P9 & (P9 - 1)) == 0) rp = P9;
        }
        pre.rows9p = (rp && (first_large || pre.rows9p)) ? (pre.rows9p | rp) : 0u;
        if (!rp) rows_ok = false;
        first_large = false;
    }
    if (!rows_ok) pre.rows9p = 0;
    if (getenv("ATSC_LARGE_NO_TRIP_TILES")) pre.sp_tiles = 0;
    return pre;
}

static int class_of(uint32_t n, uint32_t L)
{
    if (L <= 128) return 0;                               // <1,2>
    if (L <= 320) return 1;                               // <1,5>
    if (L <= 576) return 2;                               // <1,9>
    if (L <= 1280) return 3;                              // <4,5>
    if (L <= 2304) return 4;                              // <4,9>
    if (L <= 5120 && n <= MAX_FRAME_TIER_M) return 5;     // <16,5>
    if (n <= MAX_FRAME) return CLASS_LARGE;
    return -1;
}

// (cos, sin)(2 pi t / L), t < L, rounded from f64.  A 131072-sample frame's table is 139968 libm calls
// twice over -- milliseconds, per plan and per decode plan -- so the tables are kept for the life of the
// process (up to TW_CACHE_MAX entries in all; beyond that they are recomputed).
static std::mutex g_tw_mu;
static std::map<uint32_t, std::vector<float2>> g_tw_cache;
static size_t g_tw_held = 0;
static void twiddle_table(uint32_t L, float2 *out)
{
    std::mutex &mu = g_tw_mu;
    std::map<uint32_t, std::vector<float2>> &cache = g_tw_cache;
    size_t &held = g_tw_held;
    const size_t TW_CACHE_MAX = 16u << 20;  // 16 M entries = 128 MB
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(L);
        if (it != cache.end()) {
            memcpy(out, it->second.data(), (size_t)L * sizeof(float2));
            return;
        }
    }
    for (uint32_t t = 0; t < L; ++t) {
        const double a = 2.0 * 3.14159265358979323846 * (double)t / (double)L;
        out[t] = make_float2((float)cos(a), (float)sin(a));
    }
    if (L >= 512) {
        std::lock_guard<std::mutex> g(mu);
        if (held + L <= TW_CACHE_MAX && !cache.count(L)) {
            cache[L].assign(out, out + L);
            held += L;
        }
    }
}

// Fills the per-length table entry (see DevPlan) and appends the twiddle table of L if new.
static int build_plan_entry(uint32_t n, PlanTables &T, std::map<uint32_t, uint64_t> &tw_by_L,
                            bool nopad = false, bool dec = false)
{
    DevPlan p;
    memset(&p, 0, sizeof(p));
    p.n = n;
    if (n >= 128 && !nopad) {  // fft.rs:305-309
        p.L = (uint32_t)atsc_next_size(n);
        p.pre = (p.L - n) / 2;
        p.direct = 0;
    } else {
        // bounded path below 128 samples, or FFT::compress (fft.rs:366-388) which transforms the
        // frame as it is: any length; 2^a 3^b lengths >= 128 still go through the Stockham stages
        p.L = n;
        p.pre = 0;
        p.direct = (n < 128 || !is_decomposable(n)) ? 1 : 0;
    }
    p.bins = p.L / 2 + 1;
    p.mf = (3 >= n / 100) ? 3 : n / 100;
    p.dk1 = std::max(p.mf / 2, 1u);
    p.dk2 = std::max(p.mf / 10, 1u);
    p.kcap = std::min(p.bins, p.mf + 17 * p.dk1 + 5 * p.dk2);
    p.half = (!p.direct && p.L % 2 == 0) ? 1 : 0;
    p.M = p.half ? p.L / 2 : p.L;
    p.sc = p.L / p.M;
    if (!p.direct) {
        uint32_t l = p.M, s = 0, st = 1;
        auto push = [&](uint32_t r) {
            p.radix[s] = r;
            p.stmagic[s] = st >= 2 ? (uint32_t)(0x100000000ull / st) + 1u : 0u;  // st == 1: p = t
            st *= r;
            ++s;
        };
        while (l % 4 == 0 && s < 14) { push(4); l /= 4; }
        while (l % 2 == 0 && s < 14) { push(2); l /= 2; }
        while (l % 3 == 0 && s < 14) { push(3); l /= 3; }
        if (l != 1) return ATSC_E_INVALID;
        p.nstages = s;
    }
    {
        const uint32_t base = (3 >= n / 100) ? 3 : n / 100;
        const uint32_t dj1 = std::max(n / 10, 1u), dj2 = std::max(n / 100, 1u);
        uint32_t jump = 0;
        p.inv_n = 1.0 / (double)n;
        for (uint32_t t = 0; t < 23; ++t) {
            const uint32_t pts = base + jump;
            const uint32_t step = std::max(n / pts, 1u);
            const uint32_t cnt = (n + step - 1) / step;
            const uint32_t K = cnt + (((cnt - 1) * step != n - 1) ? 1u : 0u);
            p.pstep[t] = step;
            p.pK[t] = K;
            p.pmagic[t] = step >= 2 ? (uint32_t)(0x100000000ull / step) + 1u : 0u;
            const int64_t gap = (int64_t)(n - 1) - ((int64_t)K - 2) * (int64_t)step;
            p.pgap[t] = (K >= 2 && gap > 0) ? (uint32_t)gap : 1u;
            p.pry[t] = 1.0 / (double)step;
            p.pryL[t] = 1.0 / (double)p.pgap[t];
            if (t + 1 <= 17) jump += dj1;
            else if (t + 1 <= 22) jump += dj2;
        }
    }
    p.f4_m1 = p.f4_m2 = 0;
    if (!p.direct && p.M >= 64) {
        uint32_t d = (uint32_t)std::sqrt((double)p.M);
        while (d > 1 && p.M % d != 0) --d;
        if (d > 1 && p.M / d <= 480) { p.f4_m1 = d; p.f4_m2 = p.M / d; }
        // M = 243 x 9 P, P = 2 .. 32 a power of two -- every power-of-two frame length from 8192 to 131072 samples (the
        // reference chunker's chunks, L = 2^a 3^7): the split the register-resident column / row transforms and the
        // large tier's grid path are written for (k_large_cols243, k_large_rows9p<P>, atsc_large_fast.h)
        if (p.half && p.M % (243u * 9u) == 0) {
            const uint32_t P9 = p.M / (243u * 9u);
            if (P9 >= 2 && P9 <= 32 && (P9 & (P9 - 1)) == 0 && !getenv("ATSC_LARGE_SQRT_SPLIT")) { p.f4_m1 = 243; p.f4_m2 = 9 * P9; }
        }
    }
    p.sp_mf = p.sp_md = 0;
    if (!p.direct && p.M >= 1024) {
        uint32_t d = std::min(p.M, 992u);
        while (d > 1 && p.M % d != 0) --d;
        if (d >= 64 && p.M / d <= 256) { p.sp_mf = d; p.sp_md = p.M / d; }
    }
    p.p2bins = pow2_ge(p.bins);
    p.p2n = pow2_ge(n);
    p.magicL = p.L >= 2 ? (uint32_t)(0x100000000ull / p.L) + 1u : 0u;
    // AB: FFT ping-pong (8 B per complex point, +8 so the second buffer can hold bins = M+1 values),
    // and at least 8n+64 bytes for the RLE run records / hash table and the spline tables.
    p.ab_half = align16(p.direct ? 8 * p.bins : 8 * p.M + 8);
    p.ab_bytes = std::max(2 * p.ab_half, align16(8 * n + 64));
    uint32_t o = 0;
    if (dec) {
        p.o_red = o; o += 384;
        p.o_xs = o; o += align16(8 * n);
        p.o_tw = o; o += align16(8 * std::max(p.L, n));
        p.o_ab = o; o += p.ab_bytes;
        p.o_sel = o; o += align16(12 * std::max(p.kcap, 1u));
        p.o_aux = o; o += align16(4 * (n + 2));
    } else {
        const int c = class_of(n, p.L);
        const EncLds e = enc_lds(n, p.L, p.direct ? p.bins : p.M, p.direct != 0, p.kcap, c >= 0 && c <= 2);
        p.o_red = e.o_red; p.o_xs = e.o_xs; p.o_tw = e.o_tw; p.o_aux = e.o_aux; p.o_ab = e.o_ab;
        p.ab_half = e.ab_half; p.ab_bytes = e.ab_bytes; p.o_sel = e.o_sel; p.o_hist = e.o_hist;
        o = e.total;
    }
    p.lds_bytes = o;
    auto it = tw_by_L.find(p.L);
    if (it == tw_by_L.end()) {
        const uint64_t off = T.twpool.size();
        tw_by_L[p.L] = off;
        p.tw_off = off;
        T.twpool.resize(off + p.L);
        twiddle_table(p.L, T.twpool.data() + off);
    } else {
        p.tw_off = it->second;
    }
    T.by_n[n] = (uint32_t)T.plans.size();
    T.plans.push_back(p);
    return ATSC_OK;
}

// A table's way to the device.  up == nullptr: a synchronous copy.  Otherwise the bytes are staged in the context's
// page-locked buffer and a kernel on `up` copies them (launch_copy_words): for a plan built while a large transfer to the
// host is in flight, whose copy-engine queue a synchronous hipMemcpy would wait behind (0.5 ms behind 42 MB).  The staging
// buffer is handed out front to back; the caller resets h_stage_used when no such kernel can be pending any more.
static hipError_t h2d_small(atsc_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t up)
{
    if (bytes == 0) return hipSuccess;
    if (up && (bytes & 3u) == 0 && ctx->h_stage && ctx->h_stage_used + bytes + 256 <= ctx->h_stage_cap) {  // (the last 256 bytes: status words)
        unsigned char *st = ctx->h_stage + ctx->h_stage_used;
        memcpy(st, src, bytes);
        ctx->h_stage_used += (bytes + 255) & ~(size_t)255;
        return launch_copy_words(dst, st, bytes / 4, up);
    }
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}
static int upload_tables(atsc_ctx *ctx, PlanTables &T, hipStream_t up = nullptr)
{
    HIPCHK(ctx, pool_alloc(ctx, (void **)&T.d_plans, std::max<size_t>(1, T.plans.size()) * sizeof(DevPlan)));
    HIPCHK(ctx, h2d_small(ctx, T.d_plans, T.plans.data(), T.plans.size() * sizeof(DevPlan), up));
    HIPCHK(ctx, pool_alloc(ctx, (void **)&T.d_tw, std::max<size_t>(1, T.twpool.size()) * sizeof(float2)));
    HIPCHK(ctx, h2d_small(ctx, T.d_tw, T.twpool.data(), T.twpool.size() * sizeof(float2), up));
    return ATSC_OK;
}
static void free_tables(atsc_ctx *ctx, PlanTables &T)
{
    pool_free(ctx, T.d_plans);
    pool_free(ctx, T.d_tw);
    T.d_plans = nullptr;
    T.d_tw = nullptr;
}

// ------------------------------------------------------------------------------------------
// public: misc
// ------------------------------------------------------------------------------------------
extern "C" const char *atsc_version(void) { return "atsc-mi355x 0.1 (gfx950)"; }

extern "C" const char *atsc_strerror(int rc)
{
    switch (rc) {
    case ATSC_OK: return "ok";
    case ATSC_E_INVALID: return "invalid argument";
    case ATSC_E_NOMEM: return "out of memory";
    case ATSC_E_UNSUPPORTED:
        return "not implemented on the GPU path (frames longer than 131072 samples)";
    case ATSC_E_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case ATSC_E_HIP: return "HIP runtime error";
    case ATSC_E_CAPACITY: return "output buffer too small";
    case ATSC_E_FORMAT: return "malformed input bytes";
    case ATSC_E_VERSION: return "BRO version is newer than this library";
    case ATSC_E_IO: return "i/o error";
    default: return "unknown error";
    }
}

extern "C" int atsc_ctx_create(atsc_ctx **out, int device)
{
    ATSC_API_BEGIN
    if (!out) return ATSC_E_INVALID;
    *out = nullptr;
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) return ATSC_E_NO_DEVICE;
    if (device < 0 || device >= cnt) return ATSC_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return ATSC_E_NO_DEVICE;
    atsc_ctx *c = new (std::nothrow) atsc_ctx();
    if (!c) return ATSC_E_NOMEM;
    c->device = device;
    c->want_diag = getenv("ATSC_DIAG") != nullptr;
    if (const char *ds = getenv("ATSC_DEBUG_STOP")) c->debug_stop = atoi(ds);
    if (getenv("ATSC_NO_ADAPTIVE_ORDER")) c->adaptive_order = false;
    if (getenv("ATSC_ADAPTIVE_ORDER")) c->adaptive_order = true;
    // Two chains by default; four where the process runs with eight or more hardware queues (GPU_MAX_HW_QUEUES, read by
    // the HIP runtime at start-up: four by default): with four queues, four chains plus the caller's stream share
    // queues and gain nothing over two (106 us per step either way), with eight they are worth 3 % (97 -> 100 Gsamples/s,
    // three runs each)
    if (const char *hq = getenv("GPU_MAX_HW_QUEUES")) c->n_chains = atoi(hq) >= 8 ? 4 : 2;
    if (const char *ch = getenv("ATSC_CHAINS")) c->n_chains = std::min(4, std::max(1, atoi(ch)));
    *out = c;
    return ATSC_OK;
    ATSC_API_END
}
extern "C" void atsc_ctx_destroy(atsc_ctx *ctx)
{
    if (!ctx) return;
    for (auto &cp : ctx->plan_cache) atsc_plan_destroy(cp.plan);
    ctx->plan_cache.clear();
    if (ctx->work_stream) (void)hipStreamDestroy(ctx->work_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    for (auto &ev : ctx->ev_parts) (void)hipEventDestroy(ev);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (auto &ev : ctx->ev_copy)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &pr : ctx->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (ctx->d_diag) (void)hipFree(ctx->d_diag);
    if (ctx->d_queues) (void)hipFree(ctx->d_queues);
    for (auto &cs : ctx->chain_streams)
        if (cs) (void)hipStreamDestroy(cs);
    for (auto &cs : ctx->pack_streams)
        if (cs) (void)hipStreamDestroy(cs);
    for (auto &b : ctx->pool_free_list) (void)hipFree(b.first);
    // blocks still held by plans that outlive their context (a contract violation) are left alone: their
    // owners would otherwise free them a second time
    delete ctx;
}
// Gives back what the context keeps for the next call: the device blocks of its pool that no plan holds, and the
// plans it caches by frame layout for the host-pointer entry points (their scratch sets, workspaces and tables).
extern "C" int atsc_ctx_trim(atsc_ctx *ctx)
{
    ATSC_API_BEGIN
    if (!ctx) return ATSC_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto &cp : ctx->plan_cache) atsc_plan_destroy(cp.plan);  // (synchronises the device)
    ctx->plan_cache.clear();
    HIPCHK(ctx, hipDeviceSynchronize());
    for (auto &b : ctx->pool_free_list) (void)hipFree(b.first);
    ctx->pool_free_list.clear();
    ctx->pool_held = 0;
    return ATSC_OK;
    ATSC_API_END
}
// Process-wide caches: the host block kept for the next decoded result (atsc_free) and the twiddle tables.
extern "C" void atsc_release_caches(void)
{
    atsc::big_trim();
    std::lock_guard<std::mutex> g(g_tw_mu);
    g_tw_cache.clear();
    g_tw_held = 0;
}
// Page-locks caller memory for the host-pointer entry points (hipHostRegister): the host-to-device copies of
// atsc_compress_frames / atsc_compress_data then run at the link's rate and asynchronously.
extern "C" int atsc_host_register(void *p, uint64_t bytes)
{
    if (!p || !bytes) return ATSC_E_INVALID;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? ATSC_OK : ATSC_E_HIP;
}
extern "C" int atsc_host_unregister(void *p)
{
    if (!p) return ATSC_E_INVALID;
    return hipHostUnregister(p) == hipSuccess ? ATSC_OK : ATSC_E_HIP;
}
extern "C" const char *atsc_ctx_last_error(const atsc_ctx *ctx)
{
    return ctx ? ctx->last_error.c_str() : "";
}
// Timing of the dominant compress kernel: one event pair per atsc_compress_plan_dev call around
// the launch of the frame class that holds the most frames.
extern "C" int atsc_ctx_set_profiling(atsc_ctx *ctx, int on)
{
    ATSC_API_BEGIN
    if (!ctx) return ATSC_E_INVALID;
    ctx->profiling = on != 0;
    ctx->ev_used = 0;
    // the event pairs of the first timed launches exist before the first of them (creating an event takes tens of
    // microseconds now and then: not inside a region the caller is timing)
    // (device-scope release: the events only carry timestamps, and a system-scope release at the end of every timed
    // dispatch is an L2 write-back between consecutive launches)
    while (on && ctx->ev_pool.size() < 64) {
        hipEvent_t a, b;
        HIPCHK(ctx, hipEventCreateWithFlags(&a, hipEventReleaseToDevice));
        HIPCHK(ctx, hipEventCreateWithFlags(&b, hipEventReleaseToDevice));
        ctx->ev_pool.emplace_back(a, b);
    }
    return ATSC_OK;
    ATSC_API_END
}
extern "C" int atsc_ctx_profile_read(atsc_ctx *ctx, double *total_ms, uint64_t *launches)
{
    ATSC_API_BEGIN
    if (!ctx || !total_ms || !launches) return ATSC_E_INVALID;
    double tot = 0.0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_pool[i].second));
        float ms = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = ctx->ev_used;
    ctx->ev_used = 0;
    return ATSC_OK;
    ATSC_API_END
}
// Diagnostics are opt-in (they add 40 B/frame of HBM writes): enabled by this call.
extern "C" int atsc_ctx_enable_diag(atsc_ctx *ctx, int on)
{
    if (!ctx) return ATSC_E_INVALID;
    ctx->want_diag = on != 0;
    return ATSC_OK;
}
extern "C" int atsc_ctx_last_diag(atsc_ctx *ctx, atsc_frame_diag *out, uint64_t n_frames)
{
    ATSC_API_BEGIN
    if (!ctx || !out) return ATSC_E_INVALID;
    if (!ctx->d_diag || ctx->diag_n != n_frames) return fail(ctx, ATSC_E_INVALID, "no diagnostics recorded");
    HIPCHK(ctx, hipStreamSynchronize(ctx->diag_stream));
    HIPCHK(ctx, hipMemcpy(out, ctx->d_diag, n_frames * sizeof(atsc_frame_diag), hipMemcpyDeviceToHost));
    return ATSC_OK;
    ATSC_API_END
}

// ------------------------------------------------------------------------------------------
// compress plan
// ------------------------------------------------------------------------------------------
static void free_sub(atsc_ctx *ctx, SubPlan *t)
{
    if (!t) return;
    free_tables(ctx, t->tabs);
    pool_free(ctx, t->d_frames);
    pool_free(ctx, t->d_ids);
    pool_free(ctx, t->d_res);
    delete t;
}

extern "C" void atsc_plan_destroy(atsc_plan *p)
{
    if (!p) return;
    (void)hipDeviceSynchronize();  // as hipFree would: nothing in flight may still use the plan's blocks
    for (int i = 0; i < 7; ++i) free_sub(p->ctx, p->trials[i]);
    free_sub(p->ctx, p->nopad);
    free_tables(p->ctx, p->tabs);
    pool_free(p->ctx, p->d_frames);
    pool_free(p->ctx, p->d_ids);
    pool_free(p->ctx, p->d_res);
    pool_free(p->ctx, p->d_slots);
    pool_free(p->ctx, p->d_local);
    pool_free(p->ctx, p->d_blocksum);
    pool_free(p->ctx, p->d_ws);
    for (int c = 0; c < 8; ++c) {
        atsc_plan::Chain &ch = p->chains[c];
        if (c > 0) {  // chain 0 borrows the plan's own scratch set and workspace
            pool_free(p->ctx, ch.S.d_res);
            pool_free(p->ctx, ch.S.d_slots);
            pool_free(p->ctx, ch.S.d_local);
            pool_free(p->ctx, ch.S.d_blocksum);
            pool_free(p->ctx, ch.d_ws);
        }
        pool_free(p->ctx, ch.d_cost);
        pool_free(p->ctx, ch.d_bucket);
        pool_free(p->ctx, ch.d_hist);
        pool_free(p->ctx, ch.d_ids_adapt);
        if (ch.ev_fork) (void)hipEventDestroy(ch.ev_fork);
        if (ch.ev_lfork) (void)hipEventDestroy(ch.ev_lfork);
        if (ch.ev_codec) (void)hipEventDestroy(ch.ev_codec);
        if (ch.ev_done) (void)hipEventDestroy(ch.ev_done);
        for (int g = 0; g < 4; ++g)
            if (ch.ev_group[g]) (void)hipEventDestroy(ch.ev_group[g]);
    }
    delete p;
}

extern "C" int atsc_plan_create(atsc_ctx *ctx, const uint64_t *frame_off, uint64_t n_frames,
                                atsc_plan **out)
{
    ATSC_API_BEGIN
    if (!ctx || !frame_off || !out || n_frames == 0) return fail(ctx, ATSC_E_INVALID, "plan_create: bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    atsc_plan *p = new (std::nothrow) atsc_plan();
    if (!p) return ATSC_E_NOMEM;
    p->ctx = ctx;
    p->n_frames = n_frames;
    std::vector<DevFrame> frames(n_frames);
    std::vector<int> cls(n_frames);
    std::map<uint32_t, uint64_t> tw_by_L;
    p->class_count.assign(N_CLASSES, 0);
    p->class_lds.assign(N_CLASSES, 0);
    p->class_first.assign(N_CLASSES, 0);
    uint64_t slot = 0, bound = 0;
    for (uint64_t f = 0; f < n_frames; ++f) {
        if (frame_off[f + 1] <= frame_off[f]) {
            atsc_plan_destroy(p);
            return fail(ctx, ATSC_E_INVALID, "plan_create: empty or unordered frame");
        }
        const uint64_t n64 = frame_off[f + 1] - frame_off[f];
        if (n64 > MAX_FRAME) {
            atsc_plan_destroy(p);
            return fail(ctx, ATSC_E_UNSUPPORTED, "plan_create: frame longer than 131072 samples");
        }
        const uint32_t n = (uint32_t)n64;
        auto it = p->tabs.by_n.find(n);
        uint32_t pi;
        if (it == p->tabs.by_n.end()) {
            int rc = build_plan_entry(n, p->tabs, tw_by_L);
            if (rc) { atsc_plan_destroy(p); return fail(ctx, rc, "plan_create: plan entry"); }
            pi = p->tabs.by_n[n];
        } else {
            pi = it->second;
        }
        const DevPlan &dp = p->tabs.plans[pi];
        const int c = class_of(n, dp.L);
        if (c < 0) { atsc_plan_destroy(p); return fail(ctx, ATSC_E_UNSUPPORTED, "plan_create: frame class"); }
        cls[f] = c;
        p->class_count[c]++;
        p->class_lds[c] = std::max(p->class_lds[c], dp.lds_bytes);
        if (c == CLASS_LARGE) p->ws_stride = std::max(p->ws_stride, large_ws_bytes(n, dp.L, dp.kcap));
        frames[f].sample_off = frame_off[f];
        frames[f].slot_off = slot;
        frames[f].n = n;
        frames[f].plan = pi;
        const uint64_t pb = atsc_payload_bound_bytes(n);
        slot += (pb + 15) & ~15ull;
        bound += pb + 16;
    }
    p->n_samples = frame_off[n_frames] - frame_off[0];
    p->h_frames = frames;
    p->slot_bytes = slot;
    p->body_bound = bound;
    // frame ids grouped by class
    std::vector<uint32_t> ids(n_frames);
    {
        uint32_t acc = 0;
        for (int c = 0; c < N_CLASSES; ++c) { p->class_first[c] = acc; acc += p->class_count[c]; }
        std::vector<uint32_t> cur(p->class_first);
        for (uint64_t f = 0; f < n_frames; ++f) ids[cur[cls[f]]++] = (uint32_t)f;
    }
    // uniform classes: one frame length, frames and slots in arithmetic progression
    p->class_uni.assign(N_CLASSES, UniArgs());
    for (int c = 0; c < N_CLASSES; ++c) {
        UniArgs &u = p->class_uni[c];
        memset(&u, 0, sizeof(u));
        const uint32_t cnt = p->class_count[c];
        if (!cnt) continue;
        const uint32_t *cid = ids.data() + p->class_first[c];
        const DevFrame &f0 = frames[cid[0]];
        const uint64_t stride = (atsc_payload_bound_bytes(f0.n) + 15) & ~15ull;
        bool ok = true;
        for (uint32_t i = 0; i < cnt && ok; ++i) {
            const DevFrame &f = frames[cid[i]];
            ok = cid[i] == cid[0] + i && f.plan == f0.plan &&
                 f.sample_off == f0.sample_off + (uint64_t)i * f0.n &&
                 f.slot_off == f0.slot_off + (uint64_t)i * stride;
        }
        if (ok && c != CLASS_LARGE && !getenv("ATSC_NO_UNIFORM")) {
            u.enabled = 1;
            u.fid0 = cid[0];
            u.sample_off0 = f0.sample_off;
            u.slot_off0 = f0.slot_off;
            u.slot_stride = stride;
            u.n = f0.n;
            u.plan = f0.plan;
        }
    }
    p->large_tiled = choose_large_tiled(p->class_count[CLASS_LARGE]);
    if (p->class_count[CLASS_LARGE]) {
        std::vector<uint32_t> lp;
        for (uint64_t f = 0; f < n_frames; ++f)
            if (cls[f] == CLASS_LARGE) lp.push_back(frames[f].plan);
        std::sort(lp.begin(), lp.end());
        lp.erase(std::unique(lp.begin(), lp.end()), lp.end());
        p->large_pre = large_pre_extents(p->tabs.plans, lp);
        p->large_pre.even_off = 1;
        for (uint64_t f = 0; f < n_frames; ++f)
            if (cls[f] == CLASS_LARGE && (frames[f].sample_off & 1ull)) p->large_pre.even_off = 0;
    }
    int rc = upload_tables(ctx, p->tabs);
    if (rc) { atsc_plan_destroy(p); return rc; }
    const uint32_t nb = (uint32_t)((n_frames + 1023) / 1024);
#define PCHK(call)                                                                     \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) { atsc_plan_destroy(p); return fail(ctx, ATSC_E_HIP, #call, e__); } \
    } while (0)
    PCHK(pool_alloc(ctx, (void **)&p->d_frames, n_frames * sizeof(DevFrame)));
    PCHK(hipMemcpy(p->d_frames, frames.data(), n_frames * sizeof(DevFrame), hipMemcpyHostToDevice));
    PCHK(pool_alloc(ctx, (void **)&p->d_ids, n_frames * sizeof(uint32_t)));
    PCHK(hipMemcpy(p->d_ids, ids.data(), n_frames * sizeof(uint32_t), hipMemcpyHostToDevice));
    PCHK(pool_alloc(ctx, (void **)&p->d_res, n_frames * sizeof(DevResult)));
    p->slots_bytes = std::max<uint64_t>(slot, 16);
    PCHK(pool_alloc(ctx, (void **)&p->d_slots, p->slots_bytes));
    PCHK(pool_alloc(ctx, (void **)&p->d_local, n_frames * sizeof(uint32_t)));
    PCHK(pool_alloc(ctx, (void **)&p->d_blocksum, (nb + 1) * sizeof(uint64_t)));
    if (p->class_count[CLASS_LARGE]) {
        // (a multiple of 4: the groups of a call get equal shares of the slots)
        p->ws_slots = (std::min<uint32_t>(p->class_count[CLASS_LARGE], large_ws_slots(p->ws_stride)) + 3u) & ~3u;
        PCHK(pool_alloc(ctx, (void **)&p->d_ws, p->ws_stride * p->ws_slots));
    }
#undef PCHK
    *out = p;
    return ATSC_OK;
    ATSC_API_END
}
extern "C" uint64_t atsc_plan_n_frames(const atsc_plan *p) { return p ? p->n_frames : 0; }
extern "C" uint64_t atsc_plan_n_samples(const atsc_plan *p) { return p ? p->n_samples : 0; }
extern "C" uint64_t atsc_plan_body_bound(const atsc_plan *p) { return p ? p->body_bound : 0; }

static const uint32_t COMPRESSION_SPEED[7] = {0x7fffffffu, 4096, 2048, 1024, 512, 256, 128};  // frame/mod.rs:22

// Builds a SubPlan over the frames with n >= min_n; frame length override `force_n` (0 = keep).
static int build_sub(atsc_ctx *ctx, const atsc_plan *plan, uint32_t min_n, uint32_t force_n, bool nopad,
                     bool want_res, SubPlan **out)
{
    *out = nullptr;
    SubPlan *t = new (std::nothrow) SubPlan();
    if (!t) return ATSC_E_NOMEM;
    t->min_n = min_n;
    t->class_count.assign(N_CLASSES, 0);
    t->class_lds.assign(N_CLASSES, 0);
    t->class_first.assign(N_CLASSES, 0);
    std::vector<DevFrame> fr(plan->h_frames);
    std::vector<uint32_t> sel;
    std::vector<int> cls;
    std::map<uint32_t, uint64_t> tw_by_L;
    for (size_t f = 0; f < fr.size(); ++f) {
        if (fr[f].n < min_n) continue;
        const uint32_t n = force_n ? force_n : fr[f].n;
        auto it = t->tabs.by_n.find(n);
        uint32_t pi;
        if (it == t->tabs.by_n.end()) {
            int rc = build_plan_entry(n, t->tabs, tw_by_L, nopad);
            if (rc) { free_sub(ctx, t); return fail(ctx, rc, "sub plan entry"); }
            pi = t->tabs.by_n[n];
        } else {
            pi = it->second;
        }
        const DevPlan &dp = t->tabs.plans[pi];
        const int c = class_of(n, dp.L);
        if (c < 0) { free_sub(ctx, t); return fail(ctx, ATSC_E_UNSUPPORTED, "sub plan: frame class"); }
        fr[f].n = n;
        fr[f].plan = pi;
        sel.push_back((uint32_t)f);
        cls.push_back(c);
        t->class_count[c]++;
        t->class_lds[c] = std::max(t->class_lds[c], dp.lds_bytes);
    }
    t->count = (uint32_t)sel.size();
    if (t->count) {
        std::vector<uint32_t> ids(sel.size());
        uint32_t acc = 0;
        for (int c = 0; c < N_CLASSES; ++c) { t->class_first[c] = acc; acc += t->class_count[c]; }
        std::vector<uint32_t> cur(t->class_first);
        for (size_t i = 0; i < sel.size(); ++i) ids[cur[cls[i]]++] = sel[i];
        t->large_tiled = choose_large_tiled(t->class_count[CLASS_LARGE]);
        int rc = upload_tables(ctx, t->tabs);
        if (rc) { free_sub(ctx, t); return rc; }
#define TCHK(call)                                                                      \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) { free_sub(ctx, t); return fail(ctx, ATSC_E_HIP, #call, e__); } \
    } while (0)
        TCHK(pool_alloc(ctx, (void **)&t->d_frames, fr.size() * sizeof(DevFrame)));
        TCHK(hipMemcpy(t->d_frames, fr.data(), fr.size() * sizeof(DevFrame), hipMemcpyHostToDevice));
        TCHK(pool_alloc(ctx, (void **)&t->d_ids, ids.size() * sizeof(uint32_t)));
        TCHK(hipMemcpy(t->d_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (want_res) TCHK(pool_alloc(ctx, (void **)&t->d_res, fr.size() * sizeof(DevResult)));
#undef TCHK
    }
    *out = t;
    return ATSC_OK;
}

static int launch_sub(atsc_ctx *ctx, const atsc_plan *plan, const SubPlan *t, const double *d_samples,
                      const KParams &prm, uint8_t *d_slots, DevResult *res, atsc_frame_diag *diag,
                      hipStream_t s, unsigned char *d_ws)
{
    UniArgs nouni;
    memset(&nouni, 0, sizeof(nouni));
    for (int c = 0; c < N_CLASSES; ++c) {
        if (!t->class_count[c]) continue;
        hipError_t e;
        if (c == CLASS_LARGE) {
            KParams lp = prm;
            lp.large_tiled = t->large_tiled ? 1u : 0u;
            lp.prefft = 0;
            e = launch_compress_large(t->class_count[c], d_samples, t->d_frames, t->d_ids + t->class_first[c],
                                      t->tabs.d_plans, t->tabs.d_tw, lp, d_slots, res, diag,
                                      d_ws, plan->ws_stride, plan->ws_slots, s);
        }
        else
            e = launch_compress_class(c, t->class_count[c], t->class_lds[c], d_samples, t->d_frames,
                                      t->d_ids + t->class_first[c], t->tabs.d_plans, t->tabs.d_tw,
                                      prm, d_slots, res, diag, nouni, s);
        if (e != hipSuccess) return fail(ctx, ATSC_E_HIP, "launch k_compress (sub plan)", e);
    }
    return ATSC_OK;
}

// Streams and events a call needs from chain c; full: also the chain's scratch set, workspace and cost records
// (pipelined calls; chain 0 borrows the plan's own scratch set).
static int ensure_chain(atsc_ctx *ctx, const atsc_plan *plan, uint32_t c, bool full, uint32_t q)
{
    atsc_plan::Chain &ch = plan->chains[q];  // set q (scratch, events), run on chain stream c
    if (!ctx->chain_streams[c]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->chain_streams[c], hipStreamNonBlocking));
    // (the pack stream is made when a call first packs on it: a process has a handful of hardware queues, the
    // runtime deals streams over them, and a stream nobody uses still takes its turn in that deal -- made up front
    // beside chain 0 it took four chains from 101 to 86 Gsamples/s)
    if (!ch.ev_fork) {
        HIPCHK(ctx, hipEventCreateWithFlags(&ch.ev_fork, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ch.ev_lfork, hipEventDisableTiming));
        // (ev_codec rides on the last k_compress dispatch as its stop event -- hipExtLaunchKernel -- when that is possible:
        // a recorded marker between two codec launches opens a bubble of 10-20 us on the stream)
        HIPCHK(ctx, hipEventCreateWithFlags(&ch.ev_codec, hipEventReleaseToDevice));
        HIPCHK(ctx, hipEventCreateWithFlags(&ch.ev_done, hipEventDisableTiming));
        for (int g = 0; g < 4; ++g) HIPCHK(ctx, hipEventCreateWithFlags(&ch.ev_group[g], hipEventDisableTiming));
    }
    if (q == 0 && !ch.S.d_res) {
        ch.S.d_res = plan->d_res; ch.S.d_slots = plan->d_slots; ch.S.d_local = plan->d_local; ch.S.d_blocksum = plan->d_blocksum;
        ch.d_ws = plan->d_ws;
    }
    if (!full || ch.ready) return ATSC_OK;
    // A set that cannot be built completely is taken apart again (its blocks would otherwise stay in pool_live with
    // the pointers overwritten by the next attempt) and the caller goes on with fewer chains.
    hipError_t e = hipSuccess;
    // (test aid: ATSC_DEBUG_FAIL_SET=q lets the allocation of set q and of every later set fail half-way)
    static const int fail_from = getenv("ATSC_DEBUG_FAIL_SET") ? atoi(getenv("ATSC_DEBUG_FAIL_SET")) : -1;
    int taken = 0;
    auto take = [&](void **dst, size_t bytes) {
        if (e == hipSuccess && fail_from >= 0 && (int)q >= fail_from && ++taken == 3) e = hipErrorOutOfMemory;
        if (e == hipSuccess) e = pool_alloc(ctx, dst, bytes);
    };
    if (q > 0) {
        const uint32_t nb = (uint32_t)((plan->n_frames + 1023) / 1024);
        take((void **)&ch.S.d_res, plan->n_frames * sizeof(DevResult));
        take((void **)&ch.S.d_slots, plan->slots_bytes);
        take((void **)&ch.S.d_local, plan->n_frames * sizeof(uint32_t));
        take((void **)&ch.S.d_blocksum, (nb + 1) * sizeof(uint64_t));
        if (plan->ws_slots) take((void **)&ch.d_ws, plan->ws_stride * plan->ws_slots);
    }
    take((void **)&ch.d_cost, plan->n_frames * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(ch.d_cost, 0, plan->n_frames * sizeof(uint32_t));
    take((void **)&ch.d_bucket, plan->n_frames);
    take((void **)&ch.d_hist, 2 * 8 * 64 * sizeof(uint32_t));
    take((void **)&ch.d_ids_adapt, plan->n_frames * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (q > 0) {
            pool_free(ctx, ch.S.d_res); pool_free(ctx, ch.S.d_slots); pool_free(ctx, ch.S.d_local);
            pool_free(ctx, ch.S.d_blocksum); pool_free(ctx, ch.d_ws);
            ch.S = atsc_plan::Scratch();
            ch.d_ws = nullptr;
        }
        pool_free(ctx, ch.d_cost); pool_free(ctx, ch.d_bucket); pool_free(ctx, ch.d_hist); pool_free(ctx, ch.d_ids_adapt);
        ch.d_cost = nullptr; ch.d_bucket = nullptr; ch.d_hist = nullptr; ch.d_ids_adapt = nullptr;
        return fail(ctx, ATSC_E_NOMEM, "pipelined call: scratch set", e);
    }
    ch.ready = true;
    return ATSC_OK;
}
// chains a plan's pipelined calls rotate over: the context's setting, fewer when the scratch sets behind them (two per
// chain; chain 0's first is the plan's own) would take more than half of the device memory that is free when the plan
// first asks, and fewer again after a set could not be allocated (plan->chain_cap).  INTEGRATION.md has the footprint.
static uint32_t plan_chains(const atsc_ctx *ctx, const atsc_plan *plan)
{
    if (!plan->chain_cap) {
        uint32_t cap = 4;  // (atsc_ctx_set_chains takes 1..4)
        size_t free_b = 0, total_b = 0;
        uint64_t budget = 48ull << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>(budget, free_b / 2);
        else (void)hipGetLastError();
        const uint64_t per_set = plan->slots_bytes + plan->ws_stride * plan->ws_slots + plan->n_frames * 48;
        while (cap > 1 && per_set * (2 * cap - 1) > budget) --cap;
        plan->chain_cap = cap;
    }
    return std::min<uint32_t>((uint32_t)ctx->n_chains, plan->chain_cap);
}

// Shared body of atsc_compress_plan_dev (pipelined == false: everything on `stream`, scratch set 0)
// and atsc_compress_plan_dev_pipelined (the call's kernels, packing included, on the next chain's stream,
// forked from `stream` by an event).
static int compress_impl(atsc_ctx *ctx, const atsc_plan *plan, const double *d_samples,
                         int compressor, int bounded, float max_error, int sample_level,
                         uint8_t *d_body, uint64_t body_cap, uint64_t *d_rec_off,
                         uint8_t *d_chosen, double *d_err, void *stream, bool pipelined, uint64_t *d_chain = nullptr)
{
    if (!ctx || !plan || !d_samples || !d_body || !d_rec_off) return fail(ctx, ATSC_E_INVALID, "compress: null argument");
    if (sample_level < 0 || sample_level > 6) return fail(ctx, ATSC_E_INVALID, "compress: sample level");
    switch (compressor) {
    case ATSC_AUTO:
        if (!bounded) return fail(ctx, ATSC_E_INVALID, "compress: Auto needs the bounded path (compressor/mod.rs:72 todo!())");
        break;
    case ATSC_FFT:
    case ATSC_POLYNOMIAL:
    case ATSC_IDW:
    case ATSC_NOOP:
    case ATSC_CONSTANT:
    case ATSC_RLE:
        break;
    default:
        return fail(ctx, ATSC_E_INVALID, "compress: unknown compressor id");
    }
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t s = caller;  // the stream this call's kernels run on
    uint32_t ci = 0, qi = 0; // chain (stream), set (scratch + events)
    if (pipelined) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        uint32_t nch = plan_chains(ctx, plan);
        // (every set is built by the first pipelined call: an allocation of hundreds of megabytes is milliseconds, not
        // something to meet in the middle of a stream of batches.  A set that does not fit halves the ambition: the
        // sets built so far stay -- their batches may be in flight -- and the rotation goes over fewer of them; one
        // chain's first set is the plan's own, so the call itself only fails when not even the cost records fit)
        for (uint32_t q = 0; q < (plan->single_set ? 1u : 2 * nch); ++q) {
            int rc = ensure_chain(ctx, plan, q % nch, true, q);
            if (!rc) continue;
            if (q == 0) return rc;  // (set 0 is the plan's own scratch: only its cost records were asked for)
            // sets 0 .. q-1 exist: the most chains whose two sets each are among them; one set alone means every call
            // waits for its predecessor's packing
            ctx->last_error.clear();
            nch = std::max<uint32_t>(1, q / 2);
            plan->chain_cap = nch;
            if (q == 1) plan->single_set = true;
            break;
        }
        qi = plan->turn % (plan->single_set ? 1 : 2 * nch);  // two sets per chain: a chain's stream always holds a queued batch
        ci = qi % nch;
        plan->turn++;
        atsc_plan::Chain &ch = plan->chains[qi];
        // The chain's previous batch owns this scratch set until its records are packed.  Its kernels precede this
        // call's on the chain's stream, but the large tier's groups run on the other chains' streams as well; waiting
        // on the host keeps that simple and bounds the host's run-ahead to one batch per chain.
        if (ch.pending) HIPCHK(ctx, hipEventSynchronize(ch.ev_done));
        ch.pending = false;
        s = ctx->chain_streams[ci];
        // Work the caller enqueued on `stream` before this call (the copy that brought d_samples, say) precedes the
        // call's kernels.  An event record plus a cross-stream wait cost ~6 us of queue time per batch on this system,
        // so they are only spent when `stream` still has work in flight.
        const hipError_t busy = hipStreamQuery(caller);
        if (busy != hipSuccess && busy != hipErrorNotReady)  // an invalid handle, a capturing stream, an earlier fault
            return fail(ctx, ATSC_E_HIP, "pipelined call: hipStreamQuery(stream)", busy);
        if (busy == hipErrorNotReady) {
            (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: the launchers read the last error
            HIPCHK(ctx, hipEventRecord(ch.ev_fork, caller));
            HIPCHK(ctx, hipStreamWaitEvent(s, ch.ev_fork, 0));
        }
    } else {
        // a plain call after pipelined ones: the plan's own scratch set is chain 0's, and the caller expects stream
        // order with everything enqueued before
        for (int i = 0; i < 8; ++i)
            if (plan->chains[i].pending) {
                HIPCHK(ctx, hipStreamWaitEvent(s, plan->chains[i].ev_done, 0));
                plan->chains[i].pending = false;
            }
    }
    atsc_plan::Chain &CH = plan->chains[qi];
    atsc_plan::Scratch S;
    S.d_res = plan->d_res; S.d_slots = plan->d_slots; S.d_local = plan->d_local; S.d_blocksum = plan->d_blocksum;
    if (pipelined) S = CH.S;
    uint32_t large_groups = 0;  // the large tier ran as this many group chains on the context's streams (CH.ev_group[])
    const bool adapt = pipelined && ctx->adaptive_order;
    bool want_order = false;  // set once the main launches (which record the costs) are enqueued
    const uint32_t *ids_main = (adapt && CH.adapt_valid) ? CH.d_ids_adapt : plan->d_ids;
    // Where the records are packed.  One chain: on a stream of its own, beside the chain's next codecs.  Two chains
    // and more: on the chain's stream, behind its codecs -- the other chains' codecs run beside it, and every stream
    // saved matters: the runtime maps streams onto a few hardware queues (four by default), and a chain that shares
    // its queue with another chain's packing waits behind that packing's wait for *its* codecs.  Measured on the
    // 10.5 M-sample batch, two chains: 115.5 us per step with packing streams, 106.4 without (104.8 with
    // GPU_MAX_HW_QUEUES=8 and packing streams, i.e. no sharing); tools/chain_stamp_probe.py shows the stalls.
    static const bool pack_same_env = getenv("ATSC_PACK_SAME_STREAM") != nullptr, pack_apart_env = getenv("ATSC_PACK_APART") != nullptr;
    // (a context that has run three or four chains and is then set to one keeps packing on the chain's stream: a pack
    // stream created behind four chain streams was seen to serialise with chain 0 -- 176 instead of 113 us per step, and
    // no cost order -- whatever GPU_MAX_HW_QUEUES said)
    const bool pack_apart = pipelined && !pack_same_env &&
                            (pack_apart_env || (plan_chains(ctx, plan) == 1 && (ctx->pack_streams[0] || !ctx->chain_streams[2])));
    bool codec_attached = false;  // ev_codec is the stop event of this call's last codec dispatch
    auto pack = [&]() -> int {
        for (uint32_t g = 0; g < large_groups; ++g)
            if (ctx->chain_streams[g] != s) HIPCHK(ctx, hipStreamWaitEvent(s, CH.ev_group[g], 0));
        hipStream_t ps = s;
        if (pipelined) {
            if (!codec_attached) HIPCHK(ctx, hipEventRecord(CH.ev_codec, s));  // nothing reads d_samples from here on
            if (pack_apart) {  // the packing runs beside the chain's next codecs
                if (!ctx->pack_streams[ci]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->pack_streams[ci], hipStreamNonBlocking));
                ps = ctx->pack_streams[ci];
                HIPCHK(ctx, hipStreamWaitEvent(ps, CH.ev_codec, 0));
            }
        }
        hipError_t e = launch_pack(plan->d_frames, S.d_res, plan->n_frames, S.d_local, S.d_blocksum,
                                   S.d_slots, d_body, body_cap, d_rec_off, d_chosen, d_err,
                                   plan->d_ids + plan->class_first[CLASS_LARGE], plan->class_count[CLASS_LARGE], ps, d_chain);
        if (e != hipSuccess) return fail(ctx, ATSC_E_HIP, "launch pack", e);
        if (pipelined) {
            if (adapt && want_order) {
                // launch order for this chain's next batch: costliest frames first
                e = launch_order_by_cost(plan->d_ids, CH.d_ids_adapt, CH.d_cost, CH.d_bucket,
                                         CH.d_hist, plan->class_first.data(), plan->class_count.data(),
                                         CLASS_LARGE, ps);
                if (e != hipSuccess) return fail(ctx, ATSC_E_HIP, "launch order_by_cost", e);
                CH.adapt_valid = true;
            }
            HIPCHK(ctx, hipEventRecord(CH.ev_done, ps));
            CH.pending = true;
        }
        return ATSC_OK;
    };
    KParams prm;
    prm.max_err = (double)max_error;  // frame/mod.rs:67,118: `max_error as f64`
    prm.poly_target = std::round(prm.max_err * 1000.0) / 1000.0;  // polynomial.rs:230
    {
        // round_f64(err, 4) = round(err * 1e4) / 1e4 is monotone in the integer q = round(err * 1e4),
        // so the loop tests of polynomial.rs:231,255 become integer threshold tests (no divide per trip)
        const double t = prm.poly_target;
        double q = std::floor(t * 10000.0) + 2.0;
        while (q / 10000.0 > t) q -= 1.0;
        prm.poly_q_hi = q;
        q = std::floor(t * 10000.0) - 2.0;
        while (q / 10000.0 < t) q += 1.0;
        prm.poly_q_lo = q;
        if (!(t == t)) { prm.poly_q_hi = INFINITY; prm.poly_q_lo = -INFINITY; }
    }
    {
        const double v = prm.max_err * 1000.0;  // fft.rs:334 `as i32` saturates, NaN -> 0
        prm.max_err_m = (v != v) ? 0 : v >= 2147483647.0 ? INT32_MAX : v <= -2147483648.0 ? INT32_MIN : (int32_t)v;
    }
    prm.mode = compressor;
    prm.bounded = bounded;
    prm.want_diag = ctx->want_diag;
    prm.debug_stop = ctx->debug_stop;
#ifdef ATSC_STAMPS
    prm.debug_stop = (int32_t)((qi & 3u) << 24);  // dev build: the frame-span stamps are kept per scratch set
#endif
    atsc_frame_diag *d_diag = nullptr;
    if (ctx->want_diag) {
        if (ctx->diag_cap < plan->n_frames) {
            if (ctx->d_diag) (void)hipFree(ctx->d_diag);
            ctx->d_diag = nullptr;
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_diag, plan->n_frames * sizeof(atsc_frame_diag)));
            ctx->diag_cap = plan->n_frames;
        }
        ctx->diag_n = plan->n_frames;
        ctx->diag_stream = s;
        d_diag = ctx->d_diag;
    }
    prm.trial = 0;
    prm.trial_min_n = 0;
    prm.trial_res = nullptr;
    prm.cost = nullptr;
    prm.large_tiled = 0;
    prm.sparse_inv = large_sparse() ? 1u : 0u;
    prm.prefft = 0;
    prm.prestats = 0;
    prm.fast_skip = 0;
    prm.tile_stats = 0;
    unsigned char *ws_set = pipelined ? CH.d_ws : plan->d_ws;  // large-tier workspace of this call
    if (pipelined && ((compressor == ATSC_AUTO && sample_level > 0) || (compressor == ATSC_FFT && !bounded))) {
        // the sub-plans below (trial prefixes, unpadded transforms) keep one result / table set per plan: such calls
        // do not overlap with the other chains' batches
        for (int i = 0; i < 8; ++i)
            if ((uint32_t)i != qi && plan->chains[i].pending) HIPCHK(ctx, hipStreamWaitEvent(s, plan->chains[i].ev_done, 0));
    }
    if (compressor == ATSC_AUTO && sample_level > 0) {
        if (!plan->trials[sample_level]) {
            SubPlan *t = nullptr;
            const uint32_t S = COMPRESSION_SPEED[sample_level];
            int rc = build_sub(ctx, plan, S, S, false, true, &t);
            if (rc) return rc;
            plan->trials[sample_level] = t;
        }
        const SubPlan *t = plan->trials[sample_level];
        if (t->count) {
            KParams tp = prm;
            tp.trial = 1;
            int rc = launch_sub(ctx, plan, t, d_samples, tp, S.d_slots, t->d_res, nullptr, s, ws_set);
            if (rc) return rc;
            prm.trial_res = t->d_res;
            prm.trial_min_n = t->min_n;
        }
    }
    if (compressor == ATSC_FFT && !bounded) {
        // Compressor::compress -> fft() (compressor/mod.rs:67, fft.rs:466-484): the frame is
        // transformed unpadded at its own length
        if (!plan->nopad) {
            SubPlan *t = nullptr;
            int rc = build_sub(ctx, plan, 0, 0, true, false, &t);
            if (rc) return rc;
            plan->nopad = t;
        }
        int rc = launch_sub(ctx, plan, plan->nopad, d_samples, prm, S.d_slots, S.d_res, d_diag, s, ws_set);
        if (rc) return rc;
        return pack();
    }
    int dominant = 0, last_c = 0;
    if (adapt) { prm.cost = CH.d_cost; want_order = true; }
    for (int c = 1; c < N_CLASSES; ++c)
        if (plan->class_count[c] > plan->class_count[dominant]) dominant = c;
    for (int c = 0; c < N_CLASSES; ++c)
        if (plan->class_count[c]) last_c = c;
    for (int c = 0; c < N_CLASSES; ++c) {
        if (!plan->class_count[c]) continue;
        const bool timed = ctx->profiling && c == dominant;
        // the small-frame classes hand the event pair to the dispatch (hipExtLaunchKernel); the
        // large tier is several launches and is bracketed by recorded events
        const bool bracket = timed && c == CLASS_LARGE;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (timed) {
            if (ctx->ev_used == ctx->ev_pool.size()) {
                hipEvent_t a, b;
                HIPCHK(ctx, hipEventCreateWithFlags(&a, hipEventReleaseToDevice));
                HIPCHK(ctx, hipEventCreateWithFlags(&b, hipEventReleaseToDevice));
                ctx->ev_pool.emplace_back(a, b);
            }
            ev0 = ctx->ev_pool[ctx->ev_used].first;
            ev1 = ctx->ev_pool[ctx->ev_used].second;
            if (bracket) HIPCHK(ctx, hipEventRecord(ev0, s));
        }
        if (pipelined && c == last_c && c != CLASS_LARGE && !ev1) {
            // the packing waits for this dispatch's own completion event: no marker on the codec stream
            ev1 = CH.ev_codec;
            codec_attached = true;
        }
        hipError_t e;
        if (c == CLASS_LARGE) {
            KParams lp = prm;
            lp.large_tiled = plan->large_tiled ? 1u : 0u;
            // the pre-pass transforms every large frame; a trial launch or a forced codec other than
            // FFT / Auto would not use its results
            const bool pre = plan->large_pre.tiles1 && (compressor == ATSC_AUTO || compressor == ATSC_FFT);
            const uint32_t cnt = plan->class_count[c];
            // groups: contiguous shares of the large frames, each with its share of the workspace slots, on the
            // context's chain streams (see atsc_ctx::chain_streams)
            uint32_t G = 1;  // (measured: 80 frames of 131072 samples 0.45 ms as one chain, 0.50 as two, 0.73 as four)
            if (const char *ge = getenv("ATSC_LARGE_GROUPS")) G = (uint32_t)std::min(4, std::max(1, atoi(ge)));
            if (ctx->debug_stop != 0 || ctx->want_diag) G = 1;  // the probes read one launch's output
            G = std::min(G, cnt);
            if (G <= 1) {
                e = launch_compress_large(cnt, d_samples, plan->d_frames, plan->d_ids + plan->class_first[c],
                                          plan->tabs.d_plans, plan->tabs.d_tw, lp, S.d_slots, S.d_res, d_diag, ws_set,
                                          plan->ws_stride, plan->ws_slots, s, pre ? &plan->large_pre : nullptr);
            } else {
                // group g runs on chain stream g, forked from this call's stream (which may be one of them)
                for (uint32_t g = 0; g < G; ++g) {
                    int rc = ensure_chain(ctx, plan, g, false, g);
                    if (rc) return rc;
                }
                {
                    int rc = ensure_chain(ctx, plan, ci, false, qi);
                    if (rc) return rc;
                }
                hipEvent_t fork = CH.ev_lfork;
                HIPCHK(ctx, hipEventRecord(fork, s));
                const uint32_t per = (cnt + G - 1) / G, slots_g = plan->ws_slots / G;
                e = hipSuccess;
                uint32_t used = 0;
                for (uint32_t g = 0; g < G; ++g) {
                    const uint32_t g0 = g * per, g1 = std::min(cnt, g0 + per);
                    if (g0 >= g1) break;
                    hipStream_t ls = ctx->chain_streams[g];
                    if (ls != s) HIPCHK(ctx, hipStreamWaitEvent(ls, fork, 0));
                    e = launch_compress_large(g1 - g0, d_samples, plan->d_frames, plan->d_ids + plan->class_first[c] + g0,
                                              plan->tabs.d_plans, plan->tabs.d_tw, lp, S.d_slots, S.d_res, d_diag,
                                              ws_set + (uint64_t)g * slots_g * plan->ws_stride, plan->ws_stride, slots_g, ls,
                                              pre ? &plan->large_pre : nullptr);
                    if (e != hipSuccess) break;
                    if (ls != s) HIPCHK(ctx, hipEventRecord(CH.ev_group[g], ls));
                    ++used;
                }
                large_groups = used;
                if (e == hipSuccess && bracket)  // the bracket's end event follows every group
                    for (uint32_t g = 0; g < used; ++g)
                        if (ctx->chain_streams[g] != s) HIPCHK(ctx, hipStreamWaitEvent(s, CH.ev_group[g], 0));
            }
        }
        else {
            UniArgs u = plan->class_uni[c];
            u.adaptive = (ids_main != plan->d_ids) ? 1u : 0u;
            u.count = plan->class_count[c];
            u.spread = 0;
            static const bool resident_on = getenv("ATSC_RESIDENT") != nullptr;
            if (resident_on && u.enabled && compressor == ATSC_AUTO && bounded && 0.0 <= prm.max_err && !d_diag &&
                (prm.debug_stop & 0xffffff) == 0 && !prm.trial && !prm.trial_res) {
                const uint32_t g = resident_grid(c, u.n, plan->class_lds[c]);
                if (g && u.count >= 2 * g) {
                    if (!ctx->d_queues) {
                        HIPCHK(ctx, hipMalloc((void **)&ctx->d_queues, 16 * RESIDENT_Q_WORDS * sizeof(uint32_t)));
                        HIPCHK(ctx, hipMemset(ctx->d_queues, 0, 16 * RESIDENT_Q_WORDS * sizeof(uint32_t)));
                    }
                    uint32_t qx = 0;
                    while (qx < ctx->q_used && ctx->q_stream[qx] != s) ++qx;
                    if (qx == ctx->q_used && qx < 16) { ctx->q_stream[qx] = s; ctx->q_used++; }
                    if (qx < 16) {
                        u.queue = ctx->d_queues + RESIDENT_Q_WORDS * qx;
                        u.q_grid = g;
                    }
                }
            }
            static const bool no_spread = getenv("ATSC_NO_SPREAD") != nullptr;
            if (u.enabled && !u.adaptive && !no_spread && u.count >= 4096) {
                // a stride near count / golden ratio, made coprime to count
                uint32_t sp = (uint32_t)((double)u.count * 0.6180339887) | 1u;
                auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
                while (gcd(sp, u.count) != 1) sp += 2;
                u.spread = sp % u.count;
            }
            e = launch_compress_class(c, plan->class_count[c], plan->class_lds[c], d_samples,
                                      plan->d_frames, ids_main + plan->class_first[c],
                                      plan->tabs.d_plans, plan->tabs.d_tw, prm, S.d_slots,
                                      S.d_res, d_diag, u, s, ev0, ev1);
        }
        if (e != hipSuccess) return fail(ctx, ATSC_E_HIP, "launch k_compress", e);
        if (bracket) HIPCHK(ctx, hipEventRecord(ev1, s));
        if (timed) ctx->ev_used++;
    }
    return pack();
}

extern "C" int atsc_compress_plan_dev(atsc_ctx *ctx, const atsc_plan *plan, const double *d_samples,
                                      int compressor, int bounded, float max_error, int sample_level,
                                      uint8_t *d_body, uint64_t body_cap, uint64_t *d_rec_off,
                                      uint8_t *d_chosen, double *d_err, void *stream)
{
    ATSC_API_BEGIN
    return compress_impl(ctx, plan, d_samples, compressor, bounded, max_error, sample_level, d_body,
                         body_cap, d_rec_off, d_chosen, d_err, stream, false);
    ATSC_API_END
}
extern "C" int atsc_compress_plan_dev_pipelined(atsc_ctx *ctx, const atsc_plan *plan,
                                                const double *d_samples, int compressor, int bounded,
                                                float max_error, int sample_level, uint8_t *d_body,
                                                uint64_t body_cap, uint64_t *d_rec_off,
                                                uint8_t *d_chosen, double *d_err, void *stream)
{
    ATSC_API_BEGIN
    return compress_impl(ctx, plan, d_samples, compressor, bounded, max_error, sample_level, d_body,
                         body_cap, d_rec_off, d_chosen, d_err, stream, true);
    ATSC_API_END
}
extern "C" int atsc_ctx_set_adaptive_order(atsc_ctx *ctx, int on)
{
    if (!ctx) return ATSC_E_INVALID;
    ctx->adaptive_order = on != 0;
    return ATSC_OK;
}
extern "C" int atsc_plan_join(atsc_ctx *ctx, const atsc_plan *plan, void *stream)
{
    if (!ctx || !plan) return fail(ctx, ATSC_E_INVALID, "plan_join: null argument");
    for (int i = 0; i < 8; ++i)
        if (plan->chains[i].pending) HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, plan->chains[i].ev_done, 0));
    return ATSC_OK;
}
extern "C" int atsc_plan_input_release(atsc_ctx *ctx, const atsc_plan *plan, void *stream)
{
    if (!ctx || !plan) return fail(ctx, ATSC_E_INVALID, "plan_input_release: null argument");
    for (int i = 0; i < 8; ++i)
        if (plan->chains[i].pending) HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, plan->chains[i].ev_codec, 0));
    return ATSC_OK;
}
extern "C" int atsc_ctx_set_chains(atsc_ctx *ctx, int n)
{
    if (!ctx || n < 1 || n > 4) return ATSC_E_INVALID;
    ctx->n_chains = n;
    return ATSC_OK;
}

// Plan for frames [f0, f1) of a caller's offset array, from the context's cache when the layout was seen before.
static const size_t PLAN_CACHE_MAX = 4;
static int cached_plan(atsc_ctx *ctx, const uint64_t *frame_off, uint64_t f0, uint64_t f1, atsc_plan **out)
{
    const uint64_t nf = f1 - f0;
    uint64_t h = 0xcbf29ce484222325ull ^ nf;
    for (uint64_t f = f0; f < f1; ++f) {
        const uint64_t len = frame_off[f + 1] - frame_off[f];
        if (frame_off[f + 1] <= frame_off[f]) return fail(ctx, ATSC_E_INVALID, "compress_frames: empty or unordered frame");
        if (len > MAX_FRAME) return fail(ctx, ATSC_E_UNSUPPORTED, "compress_frames: frame longer than 131072 samples");
        h = (h ^ len) * 0x100000001b3ull;
    }
    for (auto &cp : ctx->plan_cache) {
        if (cp.hash != h || cp.lens.size() != nf) continue;
        bool same = true;
        for (uint64_t f = 0; f < nf && same; ++f) same = cp.lens[f] == (uint32_t)(frame_off[f0 + f + 1] - frame_off[f0 + f]);
        if (!same) continue;
        cp.stamp = ++ctx->plan_stamp;
        *out = cp.plan;
        return ATSC_OK;
    }
    std::vector<uint64_t> rel(nf + 1);
    for (uint64_t i = 0; i <= nf; ++i) rel[i] = frame_off[f0 + i] - frame_off[f0];
    atsc_plan *plan = nullptr;
    int rc = atsc_plan_create(ctx, rel.data(), nf, &plan);
    if (rc) return rc;
    if (ctx->plan_cache.size() >= PLAN_CACHE_MAX) {  // drop the layout used longest ago
        size_t victim = 0;
        for (size_t i = 1; i < ctx->plan_cache.size(); ++i)
            if (ctx->plan_cache[i].stamp < ctx->plan_cache[victim].stamp) victim = i;
        atsc_plan_destroy(ctx->plan_cache[victim].plan);
        ctx->plan_cache.erase(ctx->plan_cache.begin() + (long)victim);
    }
    atsc_ctx::CachedPlan cp;
    cp.hash = h;
    cp.stamp = ++ctx->plan_stamp;
    cp.lens.resize(nf);
    for (uint64_t f = 0; f < nf; ++f) cp.lens[f] = (uint32_t)(rel[f + 1] - rel[f]);
    cp.plan = plan;
    ctx->plan_cache.push_back(std::move(cp));
    *out = plan;
    return ATSC_OK;
}

// Host-pointer compress.  The time goes into the host-to-device copy of the samples (PCIe: 84 MB take
// 1.5 ms, the kernels 0.2 ms), so a batch of equal-length small frames is cut into parts: while the blocking
// copy of part i + 1 runs, the kernels of part i do (on the context's own stream); what is left behind the
// last copy is one part's kernels and the copy of the records.  The parts share one cached plan (same
// layout), run in stream order, and their records are laid end to end: the bytes are those of one call.
// nonfinite (optional): receives 1 when a sample is NaN or infinite (checked on the device, atsc_compress_data)
// out_alloc (optional, then `body` is ignored): the records go to a malloc'd block of head_room + *body_len bytes,
// behind head_room bytes left for the caller -- sized once the length is known.  (Handing in a worst-case buffer
// of hundreds of megabytes and shrinking it afterwards is expensive beyond the allocation: the runtime pins
// the pages a device-to-host copy lands in, and unmapping the rest of such a block stalled the NEXT
// host-to-device copy by 28 ms in a process that also runs PyTorch.)
static int compress_frames_impl(atsc_ctx *ctx, const double *samples, const uint64_t *frame_off,
                                uint64_t n_frames, int compressor, int bounded, float max_error,
                                int sample_level, uint8_t *body, uint64_t body_cap,
                                uint64_t *body_len, uint64_t *rec_off, uint8_t *chosen,
                                double *err, int *nonfinite, uint64_t head_room = 0, uint8_t **out_alloc = nullptr)
{
    if (!ctx || !samples || !frame_off || (!body && !out_alloc) || !body_len) return fail(ctx, ATSC_E_INVALID, "compress_frames: null argument");
    if (out_alloc) *out_alloc = nullptr;
    if (n_frames == 0) return fail(ctx, ATSC_E_INVALID, "compress_frames: no frames");
    static const bool trace = getenv("ATSC_TRACE_HOST") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[compress] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->work_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->work_stream, hipStreamNonBlocking));
    if (!ctx->copy_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &ev : ctx->ev_copy) HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const uint64_t ns = frame_off[n_frames] - frame_off[0];
    // parts: only for uniform frames of the LDS-resident tiers (the large tier runs best with every frame in one launch)
    const uint64_t fl = frame_off[1] - frame_off[0];
    bool uniform = fl >= 1 && fl <= MAX_FRAME;
    for (uint64_t f = 1; f < n_frames && uniform; ++f) uniform = (frame_off[f + 1] - frame_off[f]) == fl;
    uint64_t parts = 1;
    if (uniform) {
        // (large frames run best many to a launch: two parts, and only when each still has 32 frames)
        parts = fl <= MAX_FRAME_TIER_M ? std::min<uint64_t>(8, std::max<uint64_t>(1, ns >> 21)) : (n_frames >= 64 ? 2 : 1);
        if (const char *e = getenv("ATSC_HOST_PARTS")) parts = std::max(1, atoi(e));
        if (ctx->want_diag) parts = 1;  // atsc_ctx_last_diag reports one launch
        parts = std::min(parts, n_frames);
    }
    const uint64_t per = (n_frames + parts - 1) / parts;
    parts = (n_frames + per - 1) / per;
    struct Part { uint64_t f0, f1; atsc_plan *plan; };
    std::vector<Part> pt(parts);
    uint64_t bound = 0;
    for (uint64_t g = 0; g < parts; ++g) {
        pt[g].f0 = g * per;
        pt[g].f1 = std::min(n_frames, (g + 1) * per);
        int rc = cached_plan(ctx, frame_off, pt[g].f0, pt[g].f1, &pt[g].plan);
        if (rc) return rc;
        bound += atsc_plan_body_bound(pt[g].plan);
    }
    lap("plans");
    int rc = ATSC_OK;
    uint64_t piece = 2u << 20;  // samples per copy call
    bool by_parts = false, parts_overflow = false;
    if (!out_alloc && parts > 1 && body) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, samples) == hipSuccess && at.type == hipMemoryTypeHost &&
            hipPointerGetAttributes(&at, body) == hipSuccess && at.type == hipMemoryTypeHost)
            by_parts = getenv("ATSC_NO_D2H_PARTS") == nullptr;
        else
            (void)hipGetLastError();  // (unregistered memory is an answer, not an error)
    }
    if (by_parts) {
        if (!ctx->d2h_stream && hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking) != hipSuccess) by_parts = false;
        while (by_parts && ctx->ev_parts.size() < parts) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { by_parts = false; break; }
            ctx->ev_parts.push_back(ev);
        }
    }
    if (const char *e2 = getenv("ATSC_H2D_PIECE_MB")) piece = (uint64_t)std::max(1, atoi(e2)) << 17;
    double *d_x = nullptr, *d_err = nullptr;
    uint8_t *d_body = nullptr, *d_ch = nullptr;
    uint64_t *d_off = nullptr;  // part g's n_g + 1 offsets start at f0_g + g; then parts + 1 chain words:
                                // chain[g] = where part g's records start in d_body (k_pack_emit)
    std::vector<uint64_t> h_off;
    uint64_t total = 0;
    hipError_t e = hipSuccess;
    hipStream_t ws = ctx->work_stream;
#define FCHK(call)                                            \
    do {                                                      \
        e = (call);                                           \
        if (e != hipSuccess) { rc = fail(ctx, ATSC_E_HIP, #call, e); goto done; } \
    } while (0)
    FCHK(pool_alloc(ctx, (void **)&d_x, ns * sizeof(double)));
    FCHK(pool_alloc(ctx, (void **)&d_body, std::max<uint64_t>(bound, 16)));
    FCHK(pool_alloc(ctx, (void **)&d_off, (n_frames + 2 * parts + 2) * sizeof(uint64_t)));  // ... then the non-finite flag
    FCHK(pool_alloc(ctx, (void **)&d_ch, n_frames));
    FCHK(pool_alloc(ctx, (void **)&d_err, n_frames * sizeof(double)));
    lap("alloc");
    FCHK(hipMemsetAsync(d_off + n_frames + parts, 0, sizeof(uint64_t), ws));
    FCHK(hipMemsetAsync(d_off + n_frames + 2 * parts + 1, 0, sizeof(uint64_t), ws));
    for (uint64_t g = 0; g < parts; ++g) {
        const uint64_t s0 = frame_off[pt[g].f0] - frame_off[0], s1 = frame_off[pt[g].f1] - frame_off[0];
        // (pieces of at most 16 MB: one pageable copy of 40 MB and more has been seen to take 10-15 ms -- the
        // runtime pins such a source on the fly -- where the same bytes in smaller calls go at the link's rate)
        // The copies go to a stream of their own and the kernels wait for an event behind them: the order does not
        // rest on a blocking copy having landed when it returns.  From pageable memory the call still returns only
        // once the bytes are staged, i.e. part g + 1 is being copied while part g's kernels run; from memory the
        // caller registered (atsc_host_register) the copies are true DMA transfers and the host runs ahead.
        for (uint64_t c0 = s0; c0 < s1; c0 += piece) {
            const uint64_t c1 = std::min<uint64_t>(s1, c0 + piece);
            FCHK(hipMemcpyAsync(d_x + c0, samples + frame_off[0] + c0, (c1 - c0) * sizeof(double), hipMemcpyHostToDevice,
                                ctx->copy_stream));
        }
        FCHK(hipEventRecord(ctx->ev_copy[g & 1], ctx->copy_stream));
        FCHK(hipStreamWaitEvent(ws, ctx->ev_copy[g & 1], 0));
        lap("  part copy");
        if (nonfinite) FCHK(launch_nonfinite_flag(d_x + s0, s1 - s0, (uint32_t *)(d_off + n_frames + 2 * parts + 1), ws));
        rc = compress_impl(ctx, pt[g].plan, d_x + s0, compressor, bounded, max_error, sample_level, d_body, bound,
                           d_off + pt[g].f0 + g, d_ch + pt[g].f0, d_err + pt[g].f0, ws, false,
                           d_off + n_frames + parts + g);
        if (rc) goto done;
        if (by_parts) FCHK(hipEventRecord(ctx->ev_parts[g], ws));
        lap("  part launches");
    }
    lap("h2d + launches");
    if (by_parts) {
        // Registered caller memory: the uploads above are in flight and the host is free, so the records of part g go
        // back as soon as part g is packed -- the link carries both directions at once -- and what trails the last
        // upload is the last part's kernels and the last part's records (a fifth of them).
        uint64_t prev_end = 0;
        for (uint64_t g = 0; g < parts; ++g) {
            FCHK(hipEventSynchronize(ctx->ev_parts[g]));
            uint64_t end_g = 0;
            FCHK(hipMemcpy(&end_g, d_off + n_frames + parts + g + 1, sizeof(uint64_t), hipMemcpyDeviceToHost));
            if (end_g > body_cap || end_g < prev_end) { parts_overflow = true; break; }
            if (end_g > prev_end)
                FCHK(hipMemcpyAsync(body + prev_end, d_body + prev_end, end_g - prev_end, hipMemcpyDeviceToHost, ctx->d2h_stream));
            prev_end = end_g;
        }
        FCHK(hipStreamSynchronize(ctx->d2h_stream));
        lap("records by parts");
    }
    FCHK(hipStreamSynchronize(ws));
    lap("kernel tail");
    {
        uint64_t tail[2] = {0, 0};  // records' end, non-finite flag
        FCHK(hipMemcpy(tail, d_off + n_frames + 2 * parts, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        total = tail[0];
        if (nonfinite) *nonfinite = (tail[1] & 0xffffffffull) != 0 ? 1 : 0;
    }
    *body_len = total;
    if (out_alloc) {
        body = (uint8_t *)malloc(head_room + total + 1);
        if (!body) { rc = fail(ctx, ATSC_E_NOMEM, "compress_frames: output"); goto done; }
        *out_alloc = body;
        body += head_room;
    } else if (total > body_cap) {
        rc = fail(ctx, ATSC_E_CAPACITY, "compress_frames: body_cap");
        goto done;
    }
    if (total && !(by_parts && !parts_overflow)) FCHK(hipMemcpy(body, d_body, total, hipMemcpyDeviceToHost));
    if (rec_off) {
        h_off.resize(n_frames + parts);
        FCHK(hipMemcpy(h_off.data(), d_off, (n_frames + parts) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (uint64_t g = 0; g < parts; ++g)
            for (uint64_t f = pt[g].f0; f < pt[g].f1; ++f) rec_off[f] = h_off[f + g];
        rec_off[n_frames] = total;
    }
    lap("d2h records");
    if (chosen) FCHK(hipMemcpy(chosen, d_ch, n_frames, hipMemcpyDeviceToHost));
    if (err) FCHK(hipMemcpy(err, d_err, n_frames * sizeof(double), hipMemcpyDeviceToHost));
#undef FCHK
done:
    if (rc) {  // nothing may still run on the blocks that go back to the pool
        (void)hipStreamSynchronize(ws);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        if (by_parts && ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
    }
    if (rc && out_alloc && *out_alloc) { free(*out_alloc); *out_alloc = nullptr; }
    pool_free(ctx, d_x);
    pool_free(ctx, d_body);
    pool_free(ctx, d_off);
    pool_free(ctx, d_ch);
    pool_free(ctx, d_err);
    return rc;
}
extern "C" int atsc_compress_frames(atsc_ctx *ctx, const double *samples, const uint64_t *frame_off,
                                    uint64_t n_frames, int compressor, int bounded, float max_error,
                                    int sample_level, uint8_t *body, uint64_t body_cap,
                                    uint64_t *body_len, uint64_t *rec_off, uint8_t *chosen,
                                    double *err)
{
    ATSC_API_BEGIN
    return compress_frames_impl(ctx, samples, frame_off, n_frames, compressor, bounded, max_error, sample_level, body,
                                body_cap, body_len, rec_off, chosen, err, nullptr);
    ATSC_API_END
}
// atsc_compress_data's call (atsc_stream.cpp; declared in atsc_internal.h): the same, plus the device-side check
// for samples that OptimizerPlan::clean_data would have dropped
extern "C" int atsc_internal_compress_frames_scan(atsc_ctx *ctx, const double *samples, const uint64_t *frame_off,
                                                  uint64_t n_frames, int compressor, int bounded, float max_error,
                                                  int sample_level, uint64_t head_room, uint8_t **out,
                                                  uint64_t *body_len, uint64_t *rec_off, int *nonfinite)
{
    ATSC_API_BEGIN
    if (!out) return ATSC_E_INVALID;
    return compress_frames_impl(ctx, samples, frame_off, n_frames, compressor, bounded, max_error, sample_level, nullptr,
                                0, body_len, rec_off, nullptr, nullptr, nonfinite, head_room, out);
    ATSC_API_END
}

// Frames are independent (main.rs:146-163), so a batch shards over devices by contiguous frame ranges:
// the same split for every host language (atsc_amd/parallel.py::shard_range is this function).
extern "C" void atsc_shard_range(uint64_t n_units, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end)
{
    if (world == 0) world = 1;
    const uint64_t base = n_units / world, rem = n_units % world;
    const uint64_t b = (uint64_t)rank * base + std::min<uint64_t>(rank, rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

// The same with rank 0 carrying root_weight_milli / 1000 times a peer's share (every peer's records cross one link to
// the root, the root's stay where they are: atsc_amd/parallel.py::shard_range_weighted is this function).
extern "C" void atsc_shard_range_weighted(uint64_t n_units, uint32_t rank, uint32_t world, uint32_t root_weight_milli,
                                          uint64_t *begin, uint64_t *end)
{
    if (world == 0) world = 1;
    const uint64_t w0 = root_weight_milli ? root_weight_milli : 1;
    const uint64_t total = w0 + 1000ull * (world - 1);
    auto cut = [&](uint64_t r) -> uint64_t {
        if (r == 0) return 0;
        if (r >= world) return n_units;
        return (uint64_t)(((unsigned __int128)n_units * (w0 + 1000ull * (r - 1))) / total);
    };
    if (begin) *begin = cut(rank);
    if (end) *end = cut((uint64_t)rank + 1);
}

// One process, several GPUs: shard r = atsc_shard_range(n_frames, r, n_ctx) goes through ctxs[r] on a
// host thread of its own (a context is used by one thread at a time); the shards' records are laid end to
// end in frame order.  The result is byte for byte what one context returns for the whole batch.
extern "C" int atsc_compress_frames_sharded(atsc_ctx *const *ctxs, uint32_t n_ctx, const double *samples,
                                            const uint64_t *frame_off, uint64_t n_frames, int compressor,
                                            int bounded, float max_error, int sample_level, uint8_t *body,
                                            uint64_t body_cap, uint64_t *body_len, uint64_t *rec_off,
                                            uint8_t *chosen, double *err)
{
    ATSC_API_BEGIN
    if (!ctxs || n_ctx == 0 || !samples || !frame_off || !body || !body_len || n_frames == 0) return ATSC_E_INVALID;
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (!ctxs[r]) return ATSC_E_INVALID;
    if (n_ctx == 1)
        return atsc_compress_frames(ctxs[0], samples, frame_off, n_frames, compressor, bounded, max_error,
                                    sample_level, body, body_cap, body_len, rec_off, chosen, err);
    struct Shard {
        uint64_t b = 0, e = 0, len = 0;
        std::unique_ptr<uint8_t[]> buf;
        std::vector<uint64_t> off;
        int rc = ATSC_OK;
    };
    std::vector<Shard> sh(n_ctx);
    for (uint32_t r = 0; r < n_ctx; ++r) {
        atsc_shard_range(n_frames, r, n_ctx, &sh[r].b, &sh[r].e);
        const uint64_t nf = sh[r].e - sh[r].b;
        if (!nf) continue;
        uint64_t cap = 0;
        for (uint64_t f = sh[r].b; f < sh[r].e; ++f)
            cap += atsc_payload_bound_bytes(frame_off[f + 1] - frame_off[f]) + 16;
        sh[r].buf.reset(new uint8_t[cap ? cap : 1]);  // worst case; only the bytes produced are touched
        sh[r].len = cap;
        sh[r].off.resize(nf + 1);
    }
    {
        std::vector<std::thread> th;
        struct JoinAll {
            std::vector<std::thread> &t;
            ~JoinAll() { for (auto &x : t) if (x.joinable()) x.join(); }
        } join_all{th};
        for (uint32_t r = 0; r < n_ctx; ++r) {
            Shard *S = &sh[r];
            if (S->e == S->b) continue;
            atsc_ctx *c = ctxs[r];
            th.emplace_back([=] {
                uint64_t blen = 0;
                S->rc = atsc_compress_frames(c, samples, frame_off + S->b, S->e - S->b, compressor, bounded,
                                             max_error, sample_level, S->buf.get(), S->len, &blen, S->off.data(),
                                             chosen ? chosen + S->b : nullptr, err ? err + S->b : nullptr);
                S->len = blen;
            });
        }
    }
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_ctx; ++r) {
        if (sh[r].e == sh[r].b) continue;
        if (sh[r].rc) return sh[r].rc;
        total += sh[r].len;
    }
    *body_len = total;
    if (total > body_cap) return ATSC_E_CAPACITY;
    uint64_t pos = 0;
    for (uint32_t r = 0; r < n_ctx; ++r) {
        if (sh[r].e == sh[r].b) continue;
        memcpy(body + pos, sh[r].buf.get(), sh[r].len);
        if (rec_off)
            for (uint64_t f = sh[r].b; f < sh[r].e; ++f) rec_off[f] = pos + sh[r].off[f - sh[r].b];
        pos += sh[r].len;
    }
    if (rec_off) rec_off[n_frames] = total;
    return ATSC_OK;
    ATSC_API_END
}

// ------------------------------------------------------------------------------------------
// host-side format helpers
// ------------------------------------------------------------------------------------------
static uint64_t prev_power_of_two(uint64_t n)
{
    const uint64_t v = n | 1;
    const int hi = 63 - __builtin_clzll(v);
    return (1ull << hi) & n;
}
extern "C" uint64_t atsc_chunk_sizes(uint64_t len, uint64_t *out, uint64_t cap)
{
    uint64_t k = 0;
    while (len > 0) {
        uint64_t sz;
        if (len >= 131072) sz = 131072;
        else if (len <= 512) sz = len;
        else sz = prev_power_of_two(len);
        if (out && k < cap) out[k] = sz;
        ++k;
        len -= sz;
    }
    return k;
}
extern "C" uint64_t atsc_clean_data(const double *in, uint64_t n, double *out)
{
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (!(std::isnan(in[i]) || std::isinf(in[i]))) out[k++] = in[i];
    return k;
}
static uint32_t host_put_varint(uint8_t *p, uint64_t v)
{
    if (v < 251) { p[0] = (uint8_t)v; return 1; }
    uint32_t nb;
    if (v < (1ull << 16)) { p[0] = 251; nb = 2; }
    else if (v < (1ull << 32)) { p[0] = 252; nb = 4; }
    else { p[0] = 253; nb = 8; }
    for (uint32_t i = 0; i < nb; ++i) p[1 + i] = (uint8_t)(v >> (8 * i));
    return nb + 1;
}
static bool host_get_varint(const uint8_t *b, uint64_t len, uint64_t &pos, uint64_t &v)
{
    return host_varint(b, len, pos, v);
}
extern "C" uint64_t atsc_bro_prefix(uint64_t n_frames, uint8_t *out)
{
    memcpy(out, "BRRO", 4);
    const uint32_t ver = 1;
    memcpy(out + 4, &ver, 4);
    out[8] = (uint8_t)(n_frames & 0xff);  // u8 += 1 per frame, wraps (header.rs:52-54)
    return 9 + host_put_varint(out + 9, n_frames);
}
extern "C" int atsc_bro_open(const uint8_t *bro, uint64_t len, uint64_t *body_off, uint64_t *n_frames)
{
    if (!bro || len < 9) return ATSC_E_FORMAT;
    if (memcmp(bro, "BRRO", 4) != 0) return ATSC_E_FORMAT;
    uint32_t ver;
    memcpy(&ver, bro + 4, 4);
    if (ver > 1) return ATSC_E_VERSION;
    uint64_t pos = 9, nf = 0;
    if (!host_get_varint(bro, len, pos, nf)) return ATSC_E_FORMAT;
    if (body_off) *body_off = pos;
    if (n_frames) *n_frames = nf;
    return ATSC_OK;
}

// ------------------------------------------------------------------------------------------
// decompress plan
// ------------------------------------------------------------------------------------------
extern "C" void atsc_dplan_destroy(atsc_dplan *p)
{
    if (!p) return;
    (void)hipDeviceSynchronize();
    free_tables(p->ctx, p->tabs);
    pool_free(p->ctx, p->d_frames);
    pool_free(p->ctx, p->d_ids);
    pool_free(p->ctx, p->d_status);
    pool_free(p->ctx, p->d_ws);
    delete p;
}
extern "C" uint64_t atsc_dplan_n_frames(const atsc_dplan *p) { return p ? p->n_frames : 0; }
extern "C" uint64_t atsc_dplan_n_samples(const atsc_dplan *p) { return p ? p->n_samples : 0; }

// Host half of atsc_dplan_create: walks the untrusted record bytes and builds the per-frame table and the
// per-length tables.  No HIP call in here (the sanitizer build of tests/asan drives it without a GPU
// through atsc_internal_dplan_parse).
struct DPlanHost {
    std::vector<DevDFrame> frames;
    std::vector<int> cls;
    PlanTables tabs;
    std::vector<uint32_t> class_count, class_lds;
    uint64_t ws_stride = 0, n_samples = 0;
};
// begin / soft_limit / end_pos: a stream without a count in front can be walked in pieces -- the records from byte
// `begin` up to the first record boundary at or behind `soft_limit` (*end_pos: where that is)
static int dplan_parse(const uint8_t *body, uint64_t body_len, int has_count, DPlanHost &H, const char **why,
                       uint64_t begin = 0, uint64_t soft_limit = ~0ull, uint64_t *end_pos = nullptr)
{
    uint64_t pos = begin, declared = 0;
    *why = "";
    if (has_count && (begin != 0 || soft_limit != ~0ull)) { *why = "dplan_create: a counted stream is walked whole"; return ATSC_E_INVALID; }
    if (has_count && !host_get_varint(body, body_len, pos, declared)) { *why = "dplan_create: frame count"; return ATSC_E_FORMAT; }
    // untrusted bytes: a declared count can be anything; every record takes at least 4 bytes
    if (has_count && declared > body_len / 4) { *why = "dplan_create: frame count exceeds the bytes present"; return ATSC_E_FORMAT; }
    std::map<uint32_t, uint64_t> tw_by_L;
    H.class_count.assign(N_CLASSES, 0);
    H.class_lds.assign(N_CLASSES, 0);
    uint64_t out_off = 0;
    uint32_t last_n = 0, last_pi = 0;
    if (has_count) { H.frames.reserve(declared); H.cls.reserve(declared); }
    else {
        // no count in front (atsc_decompress_frames): room for a record per 64 bytes, so that the walk of a typical
        // stream does not stop to move its tables (it doubles from there if the records are shorter)
        // (of the range this call walks, not of the whole body: the parts form walks a sixth at a time)
        const uint64_t walk_end = std::min(soft_limit, body_len);
        const uint64_t guess = std::min<uint64_t>((walk_end > begin ? walk_end - begin : 0) / 64 + 16, 1ull << 22);
        H.frames.reserve(guess);
        H.cls.reserve(guess);
    }
    int last_c = -1;
    while (has_count ? H.frames.size() < declared : (pos < body_len && pos < soft_limit)) {
        HostRecord hr;
        if (!host_next_record(body, body_len, pos, hr)) { *why = "dplan_create: truncated frame record"; return ATSC_E_FORMAT; }
        const uint64_t sc = hr.sample_count, tag = hr.tag, dl = hr.payload_len;
        const uint64_t pay = hr.payload_off;
        if (tag > 6 || tag == ATSC_AUTO) { *why = "dplan_create: compressor id"; return ATSC_E_FORMAT; }
        uint64_t nout = sc;
        if (tag == ATSC_NOOP) {
            // noop_to_data returns the stored vector whatever sample_count says (noop.rs:79-83)
            // (every stored value takes at least one byte: a count above the payload length is forged)
            uint64_t q = pay + 1, cnt = 0;
            if (dl < 2 || !host_get_varint(body, pay + dl, q, cnt) || cnt > dl) { *why = "dplan_create: noop payload"; return ATSC_E_FORMAT; }
            nout = cnt;
        }
        if (nout == 0) { *why = "dplan_create: frame sample count"; return ATSC_E_FORMAT; }
        if (nout > MAX_FRAME) { *why = "dplan_create: frame longer than 131072 samples"; return ATSC_E_UNSUPPORTED; }
        const uint32_t n = (uint32_t)nout;
        uint32_t pi;
        const bool reused = (n == last_n);
        if (reused) {  // streams are runs of equal frame lengths: skip the table lookup
            pi = last_pi;
        } else {
            auto it = H.tabs.by_n.find(n);
            if (it == H.tabs.by_n.end()) {
                int rc = build_plan_entry(n, H.tabs, tw_by_L, false, true);
                if (rc) { *why = "dplan_create: plan entry"; return rc; }
                pi = H.tabs.by_n[n];
            } else {
                pi = it->second;
            }
            last_n = n;
            last_pi = pi;
        }
        const DevPlan &dp = H.tabs.plans[pi];
        const int c = (reused && last_c >= 0) ? last_c : class_of(n, dp.L);
        if (c < 0) { *why = "dplan_create: frame class"; return ATSC_E_UNSUPPORTED; }
        last_c = c;
        DevDFrame d;
        d.payload_off = pay;
        d.out_off = out_off;
        d.payload_len = (uint32_t)dl;
        d.n = n;
        d.tag = (uint32_t)tag;
        d.plan = pi;
        H.frames.push_back(d);
        H.cls.push_back(c);
        H.class_count[c]++;
        H.class_lds[c] = std::max(H.class_lds[c], dp.lds_bytes);
        if (c == CLASS_LARGE) H.ws_stride = std::max(H.ws_stride, large_ws_bytes(n, dp.L, dp.kcap));
        out_off += n;
    }
    if (H.frames.empty()) { *why = "dplan_create: no frames"; return ATSC_E_FORMAT; }
    H.n_samples = out_off;
    if (end_pos) *end_pos = pos;
    return ATSC_OK;
}
// test hook of the sanitizer build (declared in atsc_internal.h, not part of the public ABI)
extern "C" int atsc_internal_dplan_parse(const uint8_t *body, uint64_t body_len, int has_count,
                                         uint64_t *n_frames, uint64_t *n_samples)
{
    ATSC_API_BEGIN
    if (!body) return ATSC_E_INVALID;
    DPlanHost H;
    const char *why;
    int rc = dplan_parse(body, body_len, has_count, H, &why);
    if (rc) return rc;
    if (n_frames) *n_frames = H.frames.size();
    if (n_samples) *n_samples = H.n_samples;
    return ATSC_OK;
    ATSC_API_END
}

static int dplan_create_range(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count, uint64_t begin,
                              uint64_t soft_limit, uint64_t *end_pos, atsc_dplan **out, hipStream_t up = nullptr);
extern "C" int atsc_dplan_create(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len,
                                 int has_count, atsc_dplan **out)
{
    return dplan_create_range(ctx, body, body_len, has_count, 0, ~0ull, nullptr, out);
}
// up: the stream a kernel copies the plan's tables up on instead of synchronous copies (h2d_small); the plan may then
// only be used on that stream (or behind it)
static int dplan_create_range(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count, uint64_t begin,
                              uint64_t soft_limit, uint64_t *end_pos, atsc_dplan **out, hipStream_t up)
{
    ATSC_API_BEGIN
    if (!ctx || !body || !out) return fail(ctx, ATSC_E_INVALID, "dplan_create: null argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    static const bool trace = getenv("ATSC_TRACE_HOST") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[dplan]      %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    DPlanHost H;
    {
        const char *why;
        int rc = dplan_parse(body, body_len, has_count, H, &why, begin, soft_limit, end_pos);
        if (rc) return fail(ctx, rc, why);
    }
    lap("record walk");
    atsc_dplan *p = new (std::nothrow) atsc_dplan();
    if (!p) return ATSC_E_NOMEM;
    p->ctx = ctx;
    p->tabs = std::move(H.tabs);
    p->class_count = H.class_count;
    p->class_lds = H.class_lds;
    p->class_first.assign(N_CLASSES, 0);
    p->ws_stride = H.ws_stride;
    std::vector<DevDFrame> &frames = H.frames;
    std::vector<int> &cls = H.cls;
    const uint64_t out_off = H.n_samples;
    p->n_frames = frames.size();
    p->n_samples = out_off;
    std::vector<uint32_t> ids(frames.size());
    {
        uint32_t acc = 0;
        for (int c = 0; c < N_CLASSES; ++c) { p->class_first[c] = acc; acc += p->class_count[c]; }
        std::vector<uint32_t> cur(p->class_first);
        for (size_t f = 0; f < frames.size(); ++f) ids[cur[cls[f]]++] = (uint32_t)f;
    }
    p->large_tiled = choose_large_tiled(p->class_count[CLASS_LARGE]);
    if (p->class_count[CLASS_LARGE]) {
        std::vector<uint32_t> lp;
        for (size_t f = 0; f < frames.size(); ++f)
            if (cls[f] == CLASS_LARGE) lp.push_back(frames[f].plan);
        std::sort(lp.begin(), lp.end());
        lp.erase(std::unique(lp.begin(), lp.end()), lp.end());
        p->large_pre = large_pre_extents(p->tabs.plans, lp);
        // grid extent of k_decompress_large_tiles: the most tiles (8 output columns each) a large frame has
        for (uint32_t pi : lp)
            if (p->tabs.plans[pi].sp_mf) p->large_sp_tiles = std::max(p->large_sp_tiles, (p->tabs.plans[pi].sp_md + 7) / 8);
        if (getenv("ATSC_LARGE_DECODE_ONE_KERNEL")) p->large_sp_tiles = 0;
        // When every large FFT frame is one the decoder's grid path takes (k_large_dparse: a power-of-two chunk whose
        // payload fits its LDS window and holds at most 1344 entries -- what this library's first ladder trip stores),
        // the general decoder behind it has no FFT frame to bucket, and the launch of its tile grid (4.7 us of nothing)
        // is left out; should k_large_dparse leave such a frame alone after all (a malformed or flat payload), the
        // general kernel then transforms it by itself.
        if (p->large_sp_tiles && !getenv("ATSC_LARGE_NO_FAST")) {
            bool all_fast = true;
            for (size_t f = 0; f < frames.size() && all_fast; ++f) {
                if (cls[f] != CLASS_LARGE || frames[f].tag != ATSC_FFT) continue;
                const DevPlan &q = p->tabs.plans[frames[f].plan];
                const uint32_t p9 = q.f4_m2 / 9u;
                const bool geo = q.f4_m1 == 243 && q.half && p9 * 9u == q.f4_m2 && p9 >= 2 && p9 <= 32 && (p9 & (p9 - 1)) == 0 &&
                                 q.mf <= 1344 && !(q.pre & 1u) && !(frames[f].n & 1u);
                all_fast = geo && frames[f].payload_len + 4u <= 16384u && frames[f].payload_len <= 11u * 1344u + 12u;
            }
            if (all_fast) p->large_sp_tiles = 0;
        }
    }
    lap("class lists");
    int rc = upload_tables(ctx, p->tabs, up);
    if (rc) { atsc_dplan_destroy(p); return rc; }
    lap("tables up");
#define PCHK(call)                                                                      \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) { atsc_dplan_destroy(p); return fail(ctx, ATSC_E_HIP, #call, e__); } \
    } while (0)
    PCHK(pool_alloc(ctx, (void **)&p->d_frames, frames.size() * sizeof(DevDFrame)));
    PCHK(h2d_small(ctx, p->d_frames, frames.data(), frames.size() * sizeof(DevDFrame), up));
    PCHK(pool_alloc(ctx, (void **)&p->d_ids, ids.size() * sizeof(uint32_t)));
    PCHK(h2d_small(ctx, p->d_ids, ids.data(), ids.size() * sizeof(uint32_t), up));
    PCHK(pool_alloc(ctx, (void **)&p->d_status, sizeof(int)));
    if (up) PCHK(hipMemsetAsync(p->d_status, 0, sizeof(int), up));
    else PCHK(hipMemset(p->d_status, 0, sizeof(int)));
    if (p->class_count[CLASS_LARGE]) {
        p->ws_slots = std::min<uint32_t>(p->class_count[CLASS_LARGE], large_ws_slots(p->ws_stride));
        PCHK(pool_alloc(ctx, (void **)&p->d_ws, p->ws_stride * p->ws_slots));
    }
#undef PCHK
    lap("frames + ids up");
    *out = p;
    return ATSC_OK;
    ATSC_API_END
}

extern "C" int atsc_decompress_plan_dev(atsc_ctx *ctx, const atsc_dplan *dp, const uint8_t *d_body,
                                        double *d_out, void *stream)
{
    ATSC_API_BEGIN
    if (!ctx || !dp || !d_body || !d_out) return fail(ctx, ATSC_E_INVALID, "decompress: null argument");
    hipStream_t s = (hipStream_t)stream;
    for (int c = 0; c < N_CLASSES; ++c) {
        if (!dp->class_count[c]) continue;
        hipError_t e;
        if (c == CLASS_LARGE)
            e = launch_decompress_large(dp->class_count[c], dp->d_frames, dp->d_ids + dp->class_first[c],
                                        dp->tabs.d_plans, dp->tabs.d_tw, d_body, d_out, dp->d_status,
                                        dp->d_ws, dp->ws_stride, dp->ws_slots, dp->large_tiled ? 1 : 0,
                                        large_sparse() ? 1 : 0, s,
                                        dp->large_pre.tiles1 ? &dp->large_pre : nullptr, dp->large_sp_tiles);
        else
            // (a class that holds every frame is walked in index order: no id lookup in front of the frame record)
            e = launch_decompress(dp->d_frames, dp->n_frames,
                                  dp->class_count[c] == dp->n_frames ? nullptr : dp->d_ids + dp->class_first[c], c,
                                  dp->class_count[c], dp->class_lds[c], dp->tabs.d_plans,
                                  dp->tabs.d_tw, d_body, d_out, dp->d_status, s);
        if (e != hipSuccess) return fail(ctx, ATSC_E_HIP, "launch k_decompress", e);
    }
    return ATSC_OK;
    ATSC_API_END
}

// atsc_decompress_frames into memory the caller registered (atsc_host_register), for a stream without a count in
// front: in parts (six; ATSC_DECODE_PARTS).  The samples' way back is the call (84 MB: 1.58 ms at the link's 53 GB/s) and
// the host's walk over the record headers is the largest part of the rest (0.15-0.35 ms for 40960 records: sequential,
// each header locates the next); with a page-locked destination a part's samples are a DMA transfer the host does not
// wait behind, so the next part is walked, uploaded and decoded meanwhile.  Two things keep the copy engine fed: the first
// part is short (a sixteenth of the records: the engine starts 0.1 ms into the call), and a part's tables go up by a
// kernel on the work stream from page-locked staging (h2d_small) -- a synchronous hipMemcpy issued while samples are on
// their way back waited 0.5 ms behind them.  The parts' status words follow their samples into the same staging buffer.
// d_body: the records, on their way to the device (ev_copy[0]).  Returns -1 when the form does not apply (the caller goes
// on with one plan), else the call's result; on an error the caller's buffer may hold the first parts' samples.
static int decompress_frames_halves(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, const uint8_t *d_body,
                                    double *out, uint64_t out_cap, uint64_t *out_n)
{
    static const bool off = getenv("ATSC_NO_DECODE_HALVES") != nullptr;
    if (off || body_len < (1u << 20)) return -1;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, out) != hipSuccess || at.type != hipMemoryTypeHost) {
            (void)hipGetLastError();  // (pageable memory is an answer, not an error)
            return -1;
        }
    }
    if (!ctx->work_stream && hipStreamCreateWithFlags(&ctx->work_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!ctx->d2h_stream && hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    static const int n_parts_env = getenv("ATSC_DECODE_PARTS") ? atoi(getenv("ATSC_DECODE_PARTS")) : 0;
    constexpr int MAXP = 8;
    const int NP = n_parts_env >= 2 ? std::min(n_parts_env, MAXP) : 6;
    while (ctx->ev_parts.size() < (size_t)NP) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return -1;
        ctx->ev_parts.push_back(ev);
    }
    hipStream_t ws = ctx->work_stream, ds = ctx->d2h_stream;
    static const bool no_kernel_up = getenv("ATSC_NO_KERNEL_UPLOAD") != nullptr;
    if (!ctx->h_stage && !no_kernel_up) {  // the parts' tables: 32-byte frame records, ids, plans, twiddles
        const size_t cap = 8u << 20;
        if (hipHostMalloc((void **)&ctx->h_stage, cap, hipHostMallocDefault) == hipSuccess) ctx->h_stage_cap = cap;
        else { ctx->h_stage = nullptr; (void)hipGetLastError(); }
    }
    ctx->h_stage_used = 0;  // (every earlier call ended with its streams drained)
    hipStream_t up = no_kernel_up ? nullptr : ws;
    volatile int *h_status = ctx->h_stage ? (volatile int *)(ctx->h_stage + ctx->h_stage_cap - 256) : nullptr;
    atsc_dplan *dp[MAXP] = {};
    double *d_o[MAXP] = {};
    uint64_t n[MAXP] = {}, done_n = 0, pos = 0;
    // the first part: if one record holds everything behind it there is nothing to split
    // a short first part -- the copy engine starts after its walk, upload and decode -- and even ones behind it (each is
    // walked while its predecessor's samples travel: the walk is four to five times faster than the link)
    static const int first_div = getenv("ATSC_DECODE_FIRST") ? std::max(2, atoi(getenv("ATSC_DECODE_FIRST"))) : 16;
    const uint64_t lim0 = body_len / (uint64_t)std::max(first_div, NP);
    int rc = dplan_create_range(ctx, body, body_len, 0, 0, lim0, &pos, &dp[0], up);
    if (rc) return rc;
    if (pos >= body_len) {
        (void)hipStreamSynchronize(ws);  // (its tables may still be on their way up)
        atsc_dplan_destroy(dp[0]);
        return -1;
    }
    hipError_t e = hipSuccess;
#define HCHK(call)                                            \
    do {                                                      \
        e = (call);                                           \
        if (e != hipSuccess) { rc = fail(ctx, ATSC_E_HIP, #call, e); goto done; } \
    } while (0)
    HCHK(hipStreamWaitEvent(ws, ctx->ev_copy[0], 0));  // the decoders run behind the records' copy
    for (int h = 0; h < NP; ++h) {
        if (h >= 1) {
            if (pos >= body_len) break;
            // (its tables go up by a kernel on ws: the earlier parts' samples occupy the copy engine)
            const uint64_t lim = h == NP - 1 ? ~0ull : lim0 + (body_len - lim0) / (uint64_t)(NP - 1) * (uint64_t)h;
            // a record longer than a part's stride (131072-sample Noop / RLE / deep FFT records) can end behind this
            // part's limit: the part is then empty and the next one starts where the walk stands
            if (pos >= lim) continue;
            rc = dplan_create_range(ctx, body, body_len, 0, pos, lim, h == NP - 1 ? nullptr : &pos, &dp[h], up);
            if (rc) goto done;
            if (h == NP - 1) pos = body_len;
        }
        n[h] = dp[h]->n_samples;
        if (done_n + n[h] > out_cap) { rc = fail(ctx, ATSC_E_CAPACITY, "decompress_frames: out_cap"); goto done; }
        HCHK(pool_alloc(ctx, (void **)&d_o[h], std::max<uint64_t>(n[h], 1) * sizeof(double)));
        rc = atsc_decompress_plan_dev(ctx, dp[h], d_body, d_o[h], ws);
        if (rc) goto done;
        HCHK(hipEventRecord(ctx->ev_parts[h], ws));
        HCHK(hipStreamWaitEvent(ds, ctx->ev_parts[h], 0));
        if (n[h]) HCHK(hipMemcpyAsync(out + done_n, d_o[h], n[h] * sizeof(double), hipMemcpyDeviceToHost, ds));
        // the part's status word rides behind its samples into the page-locked buffer's tail
        if (h_status) { h_status[h] = -1; HCHK(hipMemcpyAsync((void *)&h_status[h], dp[h]->d_status, sizeof(int), hipMemcpyDeviceToHost, ds)); }
        done_n += n[h];
    }
    HCHK(hipStreamSynchronize(ds));
    HCHK(hipStreamSynchronize(ws));
    for (int h = 0; h < NP; ++h) {
        if (!dp[h]) continue;
        int status = 0;
        if (h_status) status = h_status[h];
        else HCHK(hipMemcpy(&status, dp[h]->d_status, sizeof(int), hipMemcpyDeviceToHost));
        if (status) { rc = fail(ctx, ATSC_E_FORMAT, "decompress_frames: malformed payload"); goto done; }
    }
    *out_n = done_n;
#undef HCHK
done:
    if (rc) {
        (void)hipStreamSynchronize(ws);
        (void)hipStreamSynchronize(ds);
        *out_n = 0;  // (the caller's buffer may hold the first parts' samples: they are not a result)
    }
    for (int h = 0; h < MAXP; ++h) {
        pool_free(ctx, d_o[h]);
        if (dp[h]) atsc_dplan_destroy(dp[h]);
    }
    return rc;
}

// out == nullptr: *out_alloc receives a malloc'd buffer of exactly the decoded length (atsc_free)
static int decompress_frames_impl(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len, int has_count,
                                  double *out, uint64_t out_cap, double **out_alloc, uint64_t *out_n)
{
    static const bool trace = getenv("ATSC_TRACE_HOST") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[decompress] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // The records start towards the device before the host walks them: from memory the caller registered
    // (atsc_host_register) the copy runs beside the walk; from pageable memory the call returns once the bytes are staged,
    // which is what the blocking copy behind the walk cost as well.
    if (!ctx || !body) return fail(ctx, ATSC_E_INVALID, "decompress_frames: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &ev : ctx->ev_copy) HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    uint8_t *d_body = nullptr;
    {
        hipError_t e0 = pool_alloc(ctx, (void **)&d_body, std::max<uint64_t>(body_len, 16));
        if (e0 != hipSuccess) return fail(ctx, ATSC_E_HIP, "decompress_frames: record buffer", e0);
        e0 = hipMemcpyAsync(d_body, body, body_len, hipMemcpyHostToDevice, ctx->copy_stream);
        if (e0 == hipSuccess) e0 = hipEventRecord(ctx->ev_copy[0], ctx->copy_stream);
        if (e0 != hipSuccess) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            pool_free(ctx, d_body);
            return fail(ctx, ATSC_E_HIP, "decompress_frames: record copy", e0);
        }
    }
    lap("h2d records (enqueue)");
    if (out && !has_count) {
        const int hrc = decompress_frames_halves(ctx, body, body_len, d_body, out, out_cap, out_n);
        if (hrc != -1) {
            lap("parts");
            (void)hipStreamSynchronize(ctx->copy_stream);
            pool_free(ctx, d_body);
            return hrc;
        }
    }
    atsc_dplan *dp = nullptr;
    int rc = atsc_dplan_create(ctx, body, body_len, has_count, &dp);
    lap("dplan_create");
    if (rc) {
        (void)hipStreamSynchronize(ctx->copy_stream);  // the copy still reads the caller's bytes
        pool_free(ctx, d_body);
        return rc;
    }
    *out_n = dp->n_samples;
    double *host = out;
    if (!out) {
        // (big_alloc: the block the caller released last is handed out again, resident pages and all)
        host = (double *)big_alloc((dp->n_samples ? dp->n_samples : 1) * sizeof(double));
        if (!host) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            pool_free(ctx, d_body);
            atsc_dplan_destroy(dp);
            return fail(ctx, ATSC_E_NOMEM, "decompress_frames: output");
        }
    } else if (dp->n_samples > out_cap) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        pool_free(ctx, d_body);
        atsc_dplan_destroy(dp);
        return fail(ctx, ATSC_E_CAPACITY, "decompress_frames: out_cap");
    }
    double *d_out = nullptr;
    int status = 0;
    hipError_t e = hipSuccess;
#define FCHK(call)                                            \
    do {                                                      \
        e = (call);                                           \
        if (e != hipSuccess) { rc = fail(ctx, ATSC_E_HIP, #call, e); goto done; } \
    } while (0)
    FCHK(pool_alloc(ctx, (void **)&d_out, dp->n_samples * sizeof(double)));
    lap("alloc");
    FCHK(hipStreamWaitEvent(nullptr, ctx->ev_copy[0], 0));  // the decoders run behind the records' copy
    rc = atsc_decompress_plan_dev(ctx, dp, d_body, d_out, nullptr);
    if (rc) goto done;
    lap("launches");
    // (the status word first: a malformed payload leaves the caller's buffer untouched)
    FCHK(hipMemcpy(&status, dp->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status) { rc = fail(ctx, ATSC_E_FORMAT, "decompress_frames: malformed payload"); goto done; }
    lap("kernels + status");
    FCHK(hipMemcpy(host, d_out, dp->n_samples * sizeof(double), hipMemcpyDeviceToHost));
    lap("d2h samples");
#undef FCHK
done:
    (void)hipStreamSynchronize(ctx->copy_stream);
    pool_free(ctx, d_body);
    pool_free(ctx, d_out);
    atsc_dplan_destroy(dp);
    lap("free + destroy");
    if (!out) {
        if (rc) atsc_free(host);
        else *out_alloc = host;
    }
    return rc;
}
extern "C" int atsc_decompress_frames(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len,
                                      int has_count, double *out, uint64_t out_cap, uint64_t *out_n)
{
    ATSC_API_BEGIN
    if (!ctx || !body || !out || !out_n) return fail(ctx, ATSC_E_INVALID, "decompress_frames: null argument");
    const int rc = decompress_frames_impl(ctx, body, body_len, has_count, out, out_cap, nullptr, out_n);
    if (rc) *out_n = 0;  // whatever `out` holds by now is not a result
    return rc;
    ATSC_API_END
}
extern "C" int atsc_decompress_frames_alloc(atsc_ctx *ctx, const uint8_t *body, uint64_t body_len,
                                            int has_count, double **out, uint64_t *out_n)
{
    ATSC_API_BEGIN
    if (!ctx || !body || !out || !out_n) return fail(ctx, ATSC_E_INVALID, "decompress_frames_alloc: null argument");
    *out = nullptr;
    const int rc = decompress_frames_impl(ctx, body, body_len, has_count, nullptr, 0, out, out_n);
    if (rc) *out_n = 0;
    return rc;
    ATSC_API_END
}
