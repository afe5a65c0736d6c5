// Error handling, allocation helpers, model layout, aggregate-launch profiling.
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include "gm_internal.h"

static thread_local char g_err[512] = "";

void gm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gm_last_error(void) { return g_err; }
extern "C" int gm_version(void) { return 100; }

// The stream-ordered pool returns freed memory to the OS at every synchronisation point by default (release
// threshold 0), so each batch build would re-map its arrays: keep freed blocks in the pool instead.
static void pool_keep_memory() {
    static thread_local int done_for = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev == done_for) { (void)hipGetLastError(); return; }
    done_for = dev;
    hipMemPool_t pool;
    uint64_t keep = UINT64_MAX;
    if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess || hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess)
        (void)hipGetLastError();
}

// Slab sizes are rounded up to eight steps per octave: the slabs of a meta-batch follow its row / edge counts, which differ by a percent from one
// meta-batch to the next; quantised sizes repeat.
static size_t pool_size_class(size_t bytes) {
    if (bytes < ((size_t)1 << 20)) return bytes;
    size_t step = (size_t)1 << 17;                       // 1 MiB / 8
    while ((step << 4) <= bytes) step <<= 1;             // step = 2^(floor(log2 bytes) - 3)
    return (bytes + step - 1) / step * step;
}

int gm_dev_alloc(void** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    pool_keep_memory();
    const bool timing = gm_knob().timing != 0;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMallocAsync(p, bytes, s);
    if (timing) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > 50.0) fprintf(stderr, "[gm timing] hipMallocAsync(%zu MB) %.1f us\n", bytes >> 20, us);
    }
    if (e != hipSuccess) {
        if (gm_knob().timing) fprintf(stderr, "[gm timing] hipMallocAsync(%zu) failed: %s -- hipMalloc\n", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        gm_set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return GM_ENOMEM;
    }
    return GM_OK;
}

void gm_dev_free(void* p, hipStream_t s) {
    if (!p) return;
    const bool timing = gm_knob().timing != 0;
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipFreeAsync(p, s);
    if (timing) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > 50.0) fprintf(stderr, "[gm timing] hipFreeAsync %.1f us\n", us);
    }
    if (e != hipSuccess) {
        if (gm_knob().timing) fprintf(stderr, "[gm timing] hipFreeAsync failed -- hipFree\n");
        (void)hipGetLastError();
        (void)hipFree(p);
    }
}

// ---------------------------------------------------------------- slab cache
// The big allocations of a batch (gm_batch::slabs, the scratch of a receptive-field build) do not go back to HIP's stream-ordered pool when a batch is
// dropped.  With the release threshold raised the pool still hands ONE block of every build / drop cycle back to the OS inside hipFreeAsync and maps
// a new one in the next build: 0.7 - 1.7 ms of host time per meta-batch at the arxiv shape, in whichever thread dropped the batch (ROCm 7.2; found
// with GM_TIMING and tools/build_prof.py, reproduced in isolation by tools/pool_free_probe.py) -- the training thread's own loop, where a receptive-
// field meta-step takes 2.2 ms.  A released slab carries an event recorded on the releasing stream (behind the batch's consumers: batch_free);
// whoever takes it next waits for that event on ITS stream: the ordering the pool would give, without the unmapping.  The cache holds at most
// GM_SLAB_CACHE_MB (default 4096) per process; what it holds at exit is left to the driver (a static destructor would run after HIP's teardown).
struct SlabEntry { char* base; size_t cap; hipEvent_t ready; int dev; };
static std::mutex g_slab_mu;
static std::vector<SlabEntry> g_slab_cache;
static size_t g_slab_cached = 0;

int gm_slab_acquire(char** base, size_t* cap, size_t need, hipStream_t s) {
    need = pool_size_class(std::max<size_t>(need, 256));
    int dev = 0;
    GM_HIP(hipGetDevice(&dev));
    SlabEntry hit{nullptr, 0, nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(g_slab_mu);
        int best = -1;
        for (size_t k = 0; k < g_slab_cache.size(); ++k) {
            const SlabEntry& e = g_slab_cache[k];
            if (e.dev == dev && e.cap >= need && e.cap <= need + need / 4 && (best < 0 || e.cap < g_slab_cache[best].cap)) best = (int)k;
        }
        if (best >= 0) { hit = g_slab_cache[best]; g_slab_cache.erase(g_slab_cache.begin() + best); g_slab_cached -= hit.cap; }
    }
    if (hit.base) {
        const hipError_t e = hipStreamWaitEvent(s, hit.ready, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(hit.ready); }
        (void)hipEventDestroy(hit.ready);
        *base = hit.base; *cap = hit.cap;
        return GM_OK;
    }
    GM_TRY(gm_dev_alloc((void**)base, need, s));
    *cap = need;
    return GM_OK;
}

void gm_slab_release(char* base, size_t cap, hipStream_t s) {
    if (!base) return;
    static const size_t limit = (size_t)(getenv("GM_SLAB_CACHE_MB") ? std::max(0, atoi(getenv("GM_SLAB_CACHE_MB"))) : 4096) << 20;
    hipEvent_t ev = nullptr;
    int dev = 0;
    if (cap > limit || hipGetDevice(&dev) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, s) != hipSuccess) {
        (void)hipGetLastError();
        if (ev) (void)hipEventDestroy(ev);
        gm_dev_free(base, s);
        return;
    }
    std::vector<SlabEntry> evict;
    {
        std::lock_guard<std::mutex> lk(g_slab_mu);
        g_slab_cache.push_back(SlabEntry{base, cap, ev, dev});
        g_slab_cached += cap;
        while (g_slab_cached > limit || g_slab_cache.size() > 256) {        // oldest first
            evict.push_back(g_slab_cache.front());
            g_slab_cached -= g_slab_cache.front().cap;
            g_slab_cache.erase(g_slab_cache.begin());
        }
    }
    for (const SlabEntry& e : evict) {
        if (hipStreamWaitEvent(s, e.ready, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(e.ready); }
        (void)hipEventDestroy(e.ready);
        gm_dev_free(e.base, s);
    }
}

// ---------------------------------------------------------------- pinned staging pool (gm_stager)
struct StageChunk { char* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false, held = false; };
// ONE pool per device for the whole process, behind a mutex: Subgraphs.batches() starts a fresh prefetch thread per epoch, and a per-thread pool
// orphaned its pinned chunks (>= 4 MiB each) and events whenever such a thread ended.  The pools are deliberately leaked at process exit (a static
// destructor would run after the HIP runtime's own teardown); what they hold is bounded by the largest number of builds in flight at once.
struct StagePool {
    std::vector<StageChunk> chunks;
    int acquire(size_t bytes) {
        int best = -1;
        for (size_t k = 0; k < chunks.size(); ++k) {
            StageChunk& c = chunks[k];
            if (c.held) continue;
            if (c.pending) { if (hipEventQuery(c.ev) != hipSuccess) { (void)hipGetLastError(); continue; } c.pending = false; }
            if (c.cap >= bytes && (best < 0 || c.cap < chunks[best].cap)) best = (int)k;
        }
        if (best < 0) {
            StageChunk c;
            c.cap = std::max<size_t>(bytes, (size_t)4 << 20);       // one chunk carries a whole batch build (1 MiB chunks: four acquisitions -- each a walk over the pool with an event query per pending chunk -- per 32-task query batch)
            if (hipHostMalloc((void**)&c.p, c.cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
            if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(c.p); return -1; }
            chunks.push_back(c); best = (int)chunks.size() - 1;
        }
        chunks[best].held = true;
        return best;
    }
};
static std::mutex g_stage_mu;
static StagePool& stage_pool_locked() {                      // caller holds g_stage_mu
    static std::map<int, StagePool>* m = new std::map<int, StagePool>();      // per device, never destroyed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    return (*m)[dev];
}
void* gm_stager::take(size_t bytes) {
    bytes = (bytes + 63) / 64 * 64;
    if (bytes > left) {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        StagePool& pool = stage_pool_locked();
        const int k = pool.acquire(bytes);
        if (k < 0) return nullptr;
        used.push_back(k);
        cur = pool.chunks[k].p; left = pool.chunks[k].cap;
    }
    void* r = cur; cur += bytes; left -= bytes;
    return r;
}
int gm_stager::upload(void* dptr, const void* src, size_t bytes) {
    if (!bytes) return GM_OK;
    void* h = take(bytes);
    GM_REQUIRE(h, GM_ENOMEM, "pinned staging allocation of %zu bytes failed", bytes);
    memcpy(h, src, bytes);
    GM_HIP(hipMemcpyAsync(dptr, h, bytes, hipMemcpyHostToDevice, s));
    return GM_OK;
}
gm_stager::~gm_stager() {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    StagePool& pool = stage_pool_locked();
    for (int k : used) {
        StageChunk& c = pool.chunks[k];
        if (hipEventRecord(c.ev, s) == hipSuccess) c.pending = true;      // reusable once the stream has passed this point
        else { (void)hipGetLastError(); (void)hipStreamSynchronize(s); c.pending = false; }
        c.held = false;
    }
}

int gm_make_layout(const gm_model_t* m, gm_layout* L) {
    GM_REQUIRE(m && m->n_gcn >= 1 && m->n_gcn <= GM_MAX_GCN, GM_EINVAL, "model: n_gcn must be in [1,%d]", GM_MAX_GCN);
    GM_REQUIRE(m->n_out >= 1 && m->n_out <= 64, GM_ERANGE, "model: n_out=%d outside [1,64]", m->n_out);
    L->n_gcn = m->n_gcn;
    L->n_out = m->n_out;
    L->link = m->link_pred ? 1 : 0;
    int64_t off = 0;
    for (int l = 0; l <= m->n_gcn; ++l) {
        GM_REQUIRE(m->dims[l] >= 1 && m->dims[l] <= 2048, GM_ERANGE, "model: dims[%d]=%d outside [1,2048]", l, m->dims[l]);
        L->dims[l] = m->dims[l];
    }
    for (int l = 0; l < m->n_gcn; ++l) {
        L->w_off[l] = off; off += (int64_t)m->dims[l] * m->dims[l + 1];
        L->b_off[l] = off; off += m->dims[l + 1];
    }
    L->hc = m->dims[m->n_gcn] * (L->link ? 2 : 1);
    L->wl_off = off; off += (int64_t)m->n_out * L->hc;
    L->bl_off = off; off += m->n_out;
    L->P = off;
    return GM_OK;
}

extern "C" int64_t gm_model_param_count(const gm_model_t* m) {
    gm_layout L;
    if (gm_make_layout(m, &L) != GM_OK) return -1;
    return L.P;
}

// ---------------------------------------------------------------- launch profiling (bench.py roofline / MFMA utilisation)
// HIP events around every launch of a category, recorded on the stream the kernel is launched on.
struct ProfCat {
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int64_t> works;   // work of every timed launch, in launch order
    size_t used = 0;
    int64_t work = 0, launches = 0;
    bool open = false;
};
static thread_local int g_prof_on = 0;
static thread_local ProfCat g_prof[GM_PROF_CATS];

extern "C" void gm_profile_enable(int32_t on) {
    g_prof_on = on;
    gm_prof_reset();
    if (!on)        // the timing events go with the measurement (hundreds per profiled meta-step)
        for (int k = 0; k < GM_PROF_CATS; ++k) { for (hipEvent_t e : g_prof[k].ev) (void)hipEventDestroy(e); g_prof[k].ev.clear(); }
}
void gm_prof_reset(int n_cats) { for (int k = 0; k < n_cats && k < GM_PROF_CATS; ++k) { ProfCat& c = g_prof[k]; c.used = 0; c.work = 0; c.launches = 0; c.open = false; } }

void gm_prof_begin(int cat, hipStream_t s, int64_t work) {
    if (!g_prof_on) return;
    ProfCat& c = g_prof[cat];
    if (c.used + 2 > c.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        c.ev.push_back(a); c.ev.push_back(b);
    }
    (void)hipEventRecord(c.ev[c.used], s);
    c.work += work; c.launches += 1; c.open = true;
    if (c.works.size() < c.used / 2 + 1) c.works.resize(c.used / 2 + 1);
    c.works[c.used / 2] = work;
}
bool gm_prof_enabled() { return g_prof_on != 0; }
void gm_prof_reset_cat(int cat) { ProfCat& c = g_prof[cat]; c.used = 0; c.work = 0; c.launches = 0; c.open = false; }
void gm_prof_note(int cat, int64_t work) { if (g_prof_on) g_prof[cat].work += work; }
void gm_prof_end(int cat, hipStream_t s) {
    ProfCat& c = g_prof[cat];
    if (!g_prof_on || !c.open) return;
    (void)hipEventRecord(c.ev[c.used + 1], s);
    c.used += 2; c.open = false;
}

extern "C" int gm_profile_read(int32_t category, double* total_ms, int64_t* launches, int64_t* work) {
    GM_REQUIRE(category >= 0 && category < GM_PROF_CATS, GM_EINVAL, "profile_read: unknown category %d", category);
    ProfCat& c = g_prof[category];
    double tot = 0;
    for (size_t k = 0; k + 1 < c.used + 1 && k < c.used; k += 2) {
        GM_HIP(hipEventSynchronize(c.ev[k + 1]));
        float ms = 0;
        GM_HIP(hipEventElapsedTime(&ms, c.ev[k], c.ev[k + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = c.launches;
    if (work) *work = c.work;
    return GM_OK;
}
// Per-launch view of one category: ms[k], work[k] of the k-th timed launch since the last reset (up to cap); returns the number of launches.
extern "C" int gm_profile_read_launches(int32_t category, double* ms, int64_t* work, int32_t cap) {
    GM_REQUIRE(category >= 0 && category < GM_PROF_CATS && ms && work && cap >= 0, GM_EINVAL, "profile_read_launches: bad arguments");
    ProfCat& c = g_prof[category];
    int n = 0;
    for (size_t k = 0; k + 1 < c.used + 1 && k < c.used && n < cap; k += 2, ++n) {
        GM_HIP(hipEventSynchronize(c.ev[k + 1]));
        float t = 0;
        GM_HIP(hipEventElapsedTime(&t, c.ev[k], c.ev[k + 1]));
        ms[n] = t; work[n] = k / 2 < c.works.size() ? c.works[k / 2] : 0;
    }
    return n;
}
extern "C" int gm_profile_aggregate(double* total_ms, int64_t* launches, int64_t* algorithmic_bytes) {
    return gm_profile_read(GM_PROF_AGG, total_ms, launches, algorithmic_bytes);
}

#ifdef GM_PROBES
// Timeline probe for tools/ (probe build only): one thread writes the 100 MHz constant clock to *out when the stream reaches this point (stamps of
// different streams are comparable, HIP event times of different streams are not)
__global__ void k_debug_stamp(unsigned long long* out) { *out = wall_clock64(); }
extern "C" int gm_debug_stamp(void* out, void* stream) {
    GM_REQUIRE(out, GM_EINVAL, "debug_stamp: NULL argument");
    hipLaunchKernelGGL(k_debug_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
#endif

static gm_knobs g_knobs;
static std::once_flag g_knobs_once;
const gm_knobs& gm_knob() {
    gm_knobs& k = g_knobs;
    std::call_once(g_knobs_once, [] {
        gm_knobs& k = g_knobs;
        auto env = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        k.agg_min_waves = env("GM_AGG_MIN_WAVES", 0);                   // 0: by batch density (gm_agg_window)
        k.agg_min_win = std::max(0, env("GM_AGG_MIN_WIN", 0));          // 0: by batch density
        k.agg_sched = env("GM_AGG_SCHED", 1);
        k.agg_hub_part = env("GM_AGG_HUB_PART", 128);                  // 0: one block per hub row
        k.agg_unr = env("GM_AGG_UNR", 24);
        k.agg_nt = env("GM_AGG_NT", 1);
        k.agg_variant = env("GM_AGG_VARIANT", 0);                      // 1 = force the generic row-per-group kernel (debug)
        k.agg_edge_tables = env("GM_AGG_EDGE_TABLES", 1);
        k.heavy_deg = env("GM_HEAVY_DEG", 0);                          // 0 = by batch density (gm_heavy_deg_for)
        k.extract_global_bitmap = env("GM_EXTRACT_GLOBAL_BITMAP", 0);
        k.feat_pad = env("GM_FEAT_PAD", 1);
        k.timing = env("GM_TIMING", 0);
        const char* gm = getenv("GM_GEMM_MODE");
        k.gemm_mode = (gm && (!strcmp(gm, "split") || !strcmp(gm, "1"))) ? 1 : (gm && (!strcmp(gm, "f32") || !strcmp(gm, "0"))) ? 0 : -1;
        k.gemm_split_min_tiles = env("GM_GEMM_SPLIT_MIN_TILES", -1);
        k.gemm_split_grid = env("GM_GEMM_SPLIT_GRID", 0);
        k.gemm_fused_rounds = std::max(0, env("GM_GEMM_FUSED_ROUNDS", 0));          // 0: by launch size (gemm.hip)
        k.gemm_plain_rounds = std::max(1, env("GM_GEMM_PLAIN_ROUNDS", 1));
        k.centre_store = env("GM_CENTRE_STORE", 2);
        k.gemm_half_tiles = env("GM_GEMM_HALF_TILES", 1);
        k.gemm_bn = env("GM_GEMM_BN", 256);
        k.gemm_mid_tiles = env("GM_GEMM_MID_TILES", 1536);
        k.gemm_glds = env("GM_GEMM_GLDS", 1);
        k.gemm_nt = env("GM_GEMM_NT", 1);
        k.gemm_small = env("GM_GEMM_SMALL", 1);
        k.wgrad_split = env("GM_WGRAD_SPLIT", 1);
        k.dz_glds = env("GM_DZ_GLDS", 1);
        k.fuse_agg = env("GM_FUSE_AGG", 1);
        k.fuse_diff = env("GM_FUSE_DIFF", 2);
        k.extract_pref16 = env("GM_EXTRACT_PREF16", 1);
        k.head_stage = env("GM_HEAD_STAGE", 1);
        k.head_threads = env("GM_HEAD_THREADS", 0);
        k.query_streams = env("GM_QUERY_STREAMS", 0);
        k.agg_mid_list = env("GM_AGG_MID_LIST", 1);
        k.agg_mid_win = env("GM_AGG_MID_WIN", 0);
        k.side_stream_priority = env("GM_SIDE_STREAM_PRIORITY", 1);
        k.wgrad_round_bias = env("GM_WGRAD_ROUND_BIAS", 25);
        k.split_pieces = env("GM_SPLIT_PIECES", 3);                   // 3: every operand carries its full 24 significand bits (the reference multiplies in fp32, learner.py:36,47); 2 = opt-in fast mode
        k.split16_min_rows = env("GM_SPLIT16_MIN_ROWS", 65536);
        k.cu_mask_support = env("GM_CU_MASK_SUPPORT", 0);
        k.wgrad_split_min_chunks = env("GM_WGRAD_SPLIT_MIN_CHUNKS", -1);
        k.agg_stream = env("GM_AGG_STREAM", 1);
        k.agg_stream_min_rows = env("GM_AGG_STREAM_MIN_ROWS", 100000);
        k.agg_stream_wgs = env("GM_AGG_STREAM_WGS", 0);
        k.agg_stream_cost = env("GM_AGG_STREAM_COST", 96);
        k.agg_stream_gather = env("GM_AGG_STREAM_GATHER", 1);
        k.agg_stream_depth = env("GM_AGG_STREAM_DEPTH", 12);
    });
    return k;
}

// Tuning knobs by the name of their environment variable, after start-up (tests force the large-launch kernels onto small fixtures with this;
// bench.py times variants in one process).  Not synchronised with concurrent library calls: set knobs while no other thread is inside the library.
static std::atomic<int> g_tuning_epoch{0};
// counts the changes made through gm_set_tuning: callers that cache sizes derived from the knobs (the host mirror's workspace sizes) key on it
extern "C" int32_t gm_tuning_epoch(void) { return g_tuning_epoch.load(std::memory_order_relaxed); }
static int gm_knobs::* gm_find_knob(const char* name) {
    (void)gm_knob();
    static const struct { const char* name; int gm_knobs::*field; } tab[] = {
        {"GM_AGG_MIN_WAVES", &gm_knobs::agg_min_waves}, {"GM_AGG_MIN_WIN", &gm_knobs::agg_min_win}, {"GM_AGG_UNR", &gm_knobs::agg_unr}, {"GM_AGG_NT", &gm_knobs::agg_nt},
        {"GM_AGG_VARIANT", &gm_knobs::agg_variant}, {"GM_GEMM_SPLIT_MIN_TILES", &gm_knobs::gemm_split_min_tiles}, {"GM_GEMM_SPLIT_GRID", &gm_knobs::gemm_split_grid},
        {"GM_GEMM_FUSED_ROUNDS", &gm_knobs::gemm_fused_rounds}, {"GM_GEMM_PLAIN_ROUNDS", &gm_knobs::gemm_plain_rounds}, {"GM_CENTRE_STORE", &gm_knobs::centre_store}, {"GM_GEMM_HALF_TILES", &gm_knobs::gemm_half_tiles},
        {"GM_GEMM_BN", &gm_knobs::gemm_bn}, {"GM_GEMM_MID_TILES", &gm_knobs::gemm_mid_tiles}, {"GM_GEMM_GLDS", &gm_knobs::gemm_glds}, {"GM_GEMM_NT", &gm_knobs::gemm_nt},
        {"GM_GEMM_SMALL", &gm_knobs::gemm_small}, {"GM_WGRAD_SPLIT", &gm_knobs::wgrad_split}, {"GM_DZ_GLDS", &gm_knobs::dz_glds}, {"GM_HEAD_STAGE", &gm_knobs::head_stage}, {"GM_HEAD_THREADS", &gm_knobs::head_threads}, {"GM_QUERY_STREAMS", &gm_knobs::query_streams}, {"GM_AGG_MID_LIST", &gm_knobs::agg_mid_list}, {"GM_AGG_MID_WIN", &gm_knobs::agg_mid_win}, {"GM_AGG_STREAM", &gm_knobs::agg_stream}, {"GM_AGG_STREAM_DEPTH", &gm_knobs::agg_stream_depth}, {"GM_AGG_STREAM_MIN_ROWS", &gm_knobs::agg_stream_min_rows},
        {"GM_SPLIT16_MIN_ROWS", &gm_knobs::split16_min_rows}, {"GM_WGRAD_SPLIT_MIN_CHUNKS", &gm_knobs::wgrad_split_min_chunks}, {"GM_WGRAD_ROUND_BIAS", &gm_knobs::wgrad_round_bias}, {"GM_TIMING", &gm_knobs::timing},
        {"GM_FUSE_DIFF", &gm_knobs::fuse_diff}, {"GM_EXTRACT_PREF16", &gm_knobs::extract_pref16},
    };
    for (const auto& e : tab)
        if (!strcmp(name, e.name)) return e.field;
    return nullptr;
}
extern "C" int gm_set_tuning(const char* name, int32_t value) {
    GM_REQUIRE(name, GM_EINVAL, "set_tuning: NULL name");
    int gm_knobs::* f = gm_find_knob(name);
    GM_REQUIRE(f, GM_EINVAL, "set_tuning: unknown or start-up-only knob %s", name);
    g_knobs.*f = value; g_tuning_epoch.fetch_add(1, std::memory_order_relaxed);
    return GM_OK;
}
// current value of a knob gm_set_tuning knows (0 for an unknown name)
extern "C" int32_t gm_get_tuning(const char* name) {
    int gm_knobs::* f = name ? gm_find_knob(name) : nullptr;
    return f ? g_knobs.*f : 0;
}

int gm_heavy_deg() { return gm_knob().heavy_deg > 0 ? std::max(2, gm_knob().heavy_deg) : 64; }
// Rows with more in-edges than this leave the wave windows for whole workgroups (hub parts).  A window row walks its edges four loads at a
// time, so a 60-edge row is a ~30-us serial chain -- the fixed cost of every launch over SPARSE induced subgraphs (arxiv shape: mean
// in-degree 2, a few hundred rows above 32), where 32 is better (4-task shard 4.73 -> 4.64 ms, roofline 0.50 -> 0.515 at task_num 32).
// Dense subgraphs (Tissue shape: mean in-degree 24) would turn a quarter of their rows into workgroups: they keep 64 (32 costs +8 %).
int gm_heavy_deg_for(int64_t rows, int64_t edges) {
    if (gm_knob().heavy_deg > 0) return std::max(2, gm_knob().heavy_deg);
    return edges <= 8 * rows ? 32 : 64;
}

// ---------------------------------------------------------------- per-device facts (one process may drive several GPUs)
static std::mutex g_dev_mu;

int gm_num_cus() {
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[dev] = n;
    return n;
}

static std::map<hipStream_t, int> g_stream_cus;
int gm_stream_cus(hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto it = g_stream_cus.find(s);
        if (it != g_stream_cus.end()) return it->second;
    }
    return gm_num_cus();
}
void gm_stream_set_cus(hipStream_t s, int cus) { std::lock_guard<std::mutex> lk(g_dev_mu); g_stream_cus[s] = cus; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KiB) once per (device, kernel)
int gm_func_full_lds(const void* fn) {
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    GM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (done.count({dev, fn})) return GM_OK;
    GM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({dev, fn});
    return GM_OK;
}
