// Error handling, allocation helpers, model layout, aggregate-launch profiling.
#include <stdarg.h>
#include <stdlib.h>
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

int gm_dev_alloc(void** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    pool_keep_memory();
    hipError_t e = hipMallocAsync(p, bytes, s);
    if (e != hipSuccess) {
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
    if (hipFreeAsync(p, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
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

// ---------------------------------------------------------------- aggregate-launch profiling
struct ProfState {
    int on = 0;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
    int64_t bytes = 0, launches = 0;
};
static thread_local ProfState g_prof;

extern "C" void gm_profile_enable(int32_t on) {
    g_prof.on = on;
    g_prof.used = 0; g_prof.bytes = 0; g_prof.launches = 0;
}
void gm_prof_reset() { g_prof.used = 0; g_prof.bytes = 0; g_prof.launches = 0; }

void gm_prof_agg_begin(hipStream_t s, int64_t bytes) {
    if (!g_prof.on) return;
    if (g_prof.used + 2 > g_prof.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof.ev.push_back(a); g_prof.ev.push_back(b);
    }
    (void)hipEventRecord(g_prof.ev[g_prof.used], s);
    g_prof.bytes += bytes; g_prof.launches += 1;
}
void gm_prof_agg_end(hipStream_t s) {
    if (!g_prof.on || g_prof.used + 2 > g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
    g_prof.used += 2;
}

extern "C" int gm_profile_aggregate(double* total_ms, int64_t* launches, int64_t* algorithmic_bytes) {
    double tot = 0;
    for (size_t k = 0; k + 1 < g_prof.used + 1 && k + 1 < g_prof.ev.size() + 1 && k < g_prof.used; k += 2) {
        GM_HIP(hipEventSynchronize(g_prof.ev[k + 1]));
        float ms = 0;
        GM_HIP(hipEventElapsedTime(&ms, g_prof.ev[k], g_prof.ev[k + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = g_prof.launches;
    if (algorithmic_bytes) *algorithmic_bytes = g_prof.bytes;
    return GM_OK;
}

int gm_heavy_deg() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("GM_HEAVY_DEG"); v = e ? atoi(e) : 64; if (v < 2) v = 2; }
    return v;
}
