// Receptive-field ("cone") tables of a batch, for gm_hparams_t.cone.
//
// Only the centre rows of the last GCN layer reach the head (learner.py:159-170), so layer L is needed at the
// centres, layer L-1 at the in-neighbours of the centres, ... and the input features at the rows L hops upstream.
// Level l (0..L) is the ascending list of batch rows whose layer-l activation is needed (level L = the centres in
// centre order); the edges between consecutive levels are kept as two compact CSRs (by destination for the forward,
// by source for the backward) whose column ids index the neighbouring level.  Every sum the dense schedule forms for
// a needed row is formed here with the same terms in the same order; rows outside the cone are never computed.
#include <algorithm>
#include "gm_internal.h"

// ------------------------------------------------------------------------------------------ device scan (int32)
#define SCAN_T 256
#define SCAN_I 8
#define SCAN_B (SCAN_T * SCAN_I)

__device__ __forceinline__ int block_excl_scan(int v, int* sm, int* total) {     // blockDim.x == SCAN_T
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        const int t = tid >= o ? sm[tid - o] : 0;
        __syncthreads();
        sm[tid] += t;
        __syncthreads();
    }
    const int incl = sm[tid];
    if (total) *total = sm[SCAN_T - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const int32_t* in, int64_t n, int32_t* bsum) {
    __shared__ int sm[SCAN_T];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) if (base + i < n) s += in[base + i];
    int tot;
    block_excl_scan(s, sm, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// exclusive scan of the block sums in place; bsum[nb] = grand total
__global__ __launch_bounds__(SCAN_T) void k_scan_top(int32_t* bsum, int nb, int32_t* total_out) {
    __shared__ int sm[SCAN_T];
    int carry = 0;
    for (int c0 = 0; c0 < nb; c0 += SCAN_T) {
        const int i = c0 + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, sm, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { bsum[nb] = carry; if (total_out) *total_out = carry; }
}
// out[i] = exclusive prefix; out[n] = total when tail != 0.  in == out is allowed.
__global__ __launch_bounds__(SCAN_T) void k_scan_final(const int32_t* in, int64_t n, const int32_t* bsum, int32_t* out, int tail) {
    __shared__ int sm[SCAN_T];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
    int v[SCAN_I], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    int run = bsum[blockIdx.x] + block_excl_scan(s, sm, nullptr);
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = bsum[gridDim.x];
}

// exclusive scan of n ints on the stream; the sum stays on the device (*d_total, optional).  bsum: scratch of n / SCAN_B + 2 ints.
static void dev_scan(const int32_t* in, int32_t* out, int64_t n, int tail, int32_t* d_total, int32_t* bsum, hipStream_t s) {
    const int nb = (int)((n + SCAN_B - 1) / SCAN_B);
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(SCAN_T), 0, s, in, n, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_T), 0, s, bsum, nb, d_total);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_T), 0, s, in, n, bsum, out, tail);
}

// ------------------------------------------------------------------------------------------ build kernels
__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) p[k] = v;
}
// pos[row[k]] = k ; afterwards k_check_pos counts the centres that lost their slot (two centres on one row)
__global__ void k_scatter_pos(const int32_t* row, int n, int32_t* pos) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) pos[row[k]] = k;
}
__global__ void k_check_pos(const int32_t* row, int n, const int32_t* pos, int32_t* bad) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && pos[row[k]] != k) atomicAdd(bad, 1);
}
// Round 6: the builder knows the level sizes only as UPPER BOUNDS while it queues its kernels (one host round trip at the very end instead of ~10 per
// level): grids are sized by the bound, the actual count is read from device memory (n_ptr).
// one wave per upper-level row: flag the sources of its in-edges, record its in-degree
__global__ __launch_bounds__(256) void k_mark(const int32_t* up_row, const int32_t* n_ptr, const int32_t* indptr, const int32_t* indices, int32_t* flags, int32_t* deg) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= *n_ptr) return;
    const int r = up_row[q], e0 = indptr[r], e1 = indptr[r + 1];
    for (int e = e0 + lane; e < e1; e += 64) flags[indices[e]] = 1;
    if (lane == 0) deg[q] = e1 - e0;
}
__global__ void k_set_off(const int32_t* scan, const int32_t* set_row_off, int sets, int64_t rows, const int32_t* total, int32_t* set_off) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > sets) return;
    const int64_t r = set_row_off[t];
    set_off[t] = (t == sets || r >= rows) ? *total : scan[r];
}
// compact the flagged rows; scan[] becomes the row -> compact id map (-1 outside the level)
__global__ void k_level_rows(const int32_t* flags, int32_t* scan, int64_t rows, const float* norm, const int32_t* feat_row,
                             int32_t* lrow, float* lnorm, int32_t* lfeat) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        if (flags[r]) {
            const int p = scan[r];
            lrow[p] = (int32_t)r; lnorm[p] = norm[r];
            if (lfeat) lfeat[p] = feat_row[r];
        } else scan[r] = -1;
    }
}
// forward CSR of the upper level: every in-edge of an upper row, sources renamed to compact ids of the lower level
__global__ __launch_bounds__(256) void k_fill_in(const int32_t* up_row, const int32_t* n_ptr, const int32_t* indptr, const int32_t* indices, const int32_t* pos_lo,
                                                 const int32_t* cptr, int32_t* cidx) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= *n_ptr) return;
    const int r = up_row[q], e0 = indptr[r], n = indptr[r + 1] - e0, o = cptr[q];
    for (int j = lane; j < n; j += 64) cidx[o + j] = pos_lo[indices[e0 + j]];
}
// backward CSR: out-edges of a lower-level row that end in the upper level, in the batch's by-source order.
// pass 0 counts, pass 1 fills (order preserved with a ballot prefix: deterministic).
__global__ __launch_bounds__(256) void k_out_edges(const int32_t* lo_row, const int32_t* n_ptr, const int32_t* indptr_t, const int32_t* indices_t, const int32_t* pos_up,
                                                   int32_t* cnt, const int32_t* tptr, int32_t* tidx) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= *n_ptr) return;
    const int u = lo_row[p], e0 = indptr_t[u], e1 = indptr_t[u + 1];
    int base = tidx ? tptr[p] : 0;
    for (int eb = e0; eb < e1; eb += 64) {
        const int e = eb + lane;
        const int q = e < e1 ? pos_up[indices_t[e]] : -1;
        const unsigned long long m = __ballot(q >= 0);
        if (tidx && q >= 0) tidx[base + __popcll(m & ((1ull << lane) - 1ull))] = q;
        base += __popcll(m);
    }
    if (!tidx && lane == 0) cnt[p] = base;
}
// hub rows of a compact CSR, ascending (ordered compaction: flag -> scan -> scatter; no host sort): flag[r] = in-degree of r above thr, r < *n_ptr
__global__ void k_heavy_flag(const int32_t* indptr, const int32_t* n_ptr, int bound, int thr, int32_t* flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < bound) flag[r] = (r < *n_ptr && indptr[r + 1] - indptr[r] > thr) ? 1 : 0;
}
__global__ void k_heavy_scatter(const int32_t* flag, const int32_t* pos, int bound, int cap, int32_t* list) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < bound && flag[r] && pos[r] < cap) list[pos[r]] = r;
}

// ------------------------------------------------------------------------------------------ host
void gm_cone_free(gm_cone* c, hipStream_t s) {
    if (!c) return;
    gm_dev_free(c->slab, s);        // every level array lives in it
    delete c;
}

// 256-byte aligned carving; base == NULL: sizing pass
struct ConeCarver {
    char* base; size_t used = 0;
    explicit ConeCarver(void* p) : base((char*)p) {}
    template <class T> T* take(size_t n) {
        const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) / 256 * 256;
        T* r = base ? (T*)(base + used) : nullptr;
        used += bytes;
        return r;
    }
};

// GEMM row tiles and weight-gradient chunks of one level (never straddling two sets: each set has its own weights); uploaded through pinned staging
static int level_tables(gm_cone_level& v, int sets, gm_stager& sg) {
    std::vector<int32_t> tiles, chunks, coff(sets + 1, 0);
    const int64_t cr = gm_wgrad_chunk_rows(v.h_set_off);
    for (int t = 0; t < sets; ++t) {
        const int r0 = v.h_set_off[t], r1 = v.h_set_off[t + 1];
        for (int r = r0; r < r1; r += GM_GEMM_BM) { tiles.push_back(t); tiles.push_back(r); tiles.push_back(std::min(GM_GEMM_BM, r1 - r)); }
        for (int r = r0; r < r1; r += (int)cr) { chunks.push_back(t); chunks.push_back(r); chunks.push_back(std::min<int>((int)cr, r1 - r)); }
        coff[t + 1] = (int32_t)(chunks.size() / 3);
    }
    v.n_tiles = (int32_t)(tiles.size() / 3); v.n_chunks = (int32_t)(chunks.size() / 3);
    GM_TRY(sg.upload(v.d_tiles, tiles)); GM_TRY(sg.upload(v.d_chunks, chunks)); GM_TRY(sg.upload(v.d_set_chunk_off, coff));
    return GM_OK;
}

// The build queues EVERYTHING from upper bounds and makes ONE host round trip at the end (round 6).  Before, every level cost ~10 round trips (three
// scans, two hub lists with a host sort each, the set offsets, pageable table uploads): ~45 per meta-batch, each waiting for its few microseconds of
// kernels to be scheduled beside a running meta-step -- 6 ms of a prefetched batch build at the arxiv shape, three times the receptive-field step it feeds.
// Bounds: level L = the centres (exact); level L - 1 holds at most as many rows as the centres have in-edges (gm_batch::n_e1, which is also the
// exact edge count into level L); deeper levels at most every row / every edge of the batch.  Level arrays are carved from ONE allocation of the
// bounds' size (arxiv query batch, two layers: ~45 MB), grids are sized by the bounds and read the real counts from device memory.
static int cone_build(const gm_batch* b, int L, hipStream_t s, gm_cone* c) {
    const int64_t rows = b->rows; const int sets = b->sets;
    c->L = L; c->heavy_deg = gm_heavy_deg();
    GM_REQUIRE(rows < ((int64_t)1 << 30) && b->edges < ((int64_t)1 << 30), GM_ERANGE, "cone: batch too large");
    int64_t Bn[GM_MAX_GCN + 1], Be[GM_MAX_GCN + 1];               // row bound of level l; bound of the edges from level l - 1 into level l
    Bn[L] = b->n_c; Be[L] = b->n_e1;
    for (int l = L - 1; l >= 0; --l) { Bn[l] = (l == L - 1) ? std::min<int64_t>(rows, b->n_e1) : rows; Be[l] = l > 0 ? b->edges : 0; }
    auto tile_cap = [&](int64_t n) { return (size_t)(n / GM_GEMM_BM + sets + 1) * 3; };
    auto chunk_cap = [&](int64_t n) { return (size_t)(n / 128 + sets + 1) * 3; };      // (gm_wgrad_chunk_rows returns at least 128 rows per chunk)
    auto hcap = [&](int64_t e) { return (size_t)(e / c->heavy_deg + 1); };
    auto carve = [&](ConeCarver& cv) {
        for (int l = 0; l <= L; ++l) {
            gm_cone_level& v = c->lv[l];
            v.d_row = cv.take<int32_t>(Bn[l]); v.d_norm = cv.take<float>(Bn[l]); v.d_set_off = cv.take<int32_t>(sets + 1);
            v.d_feat_row = l == 0 ? cv.take<int32_t>(Bn[l]) : nullptr;
            v.d_tiles = cv.take<int32_t>(tile_cap(Bn[l])); v.d_chunks = cv.take<int32_t>(chunk_cap(Bn[l])); v.d_set_chunk_off = cv.take<int32_t>(sets + 1);
            if (l > 0) {
                v.d_indptr = cv.take<int32_t>(Bn[l] + 1); v.d_indices = cv.take<int32_t>(Be[l]);
                v.d_indptr_t = cv.take<int32_t>(Bn[l - 1] + 1); v.d_indices_t = cv.take<int32_t>(Be[l]);
                v.d_heavy[0] = cv.take<int32_t>(hcap(Be[l])); v.d_heavy[1] = cv.take<int32_t>(hcap(Be[l]));
            }
        }
    };
    { ConeCarver size(nullptr); carve(size); GM_TRY(gm_dev_alloc(&c->slab, size.used + 256, s)); }
    { ConeCarver cv(c->slab); carve(cv); }
    // scratch of the build (freed, stream-ordered, when it returns): position maps, flags, degree / count arrays, scan partials, the device-side counts
    const int64_t maxb = std::max<int64_t>(rows, 1);
    void* tmp = nullptr;
    ConeCarver ts(nullptr);
    auto carve_tmp = [&](ConeCarver& cv, int32_t*& posA, int32_t*& posB, int32_t*& flags, int32_t*& deg, int32_t*& hpos, int32_t*& bsum, int32_t*& cnts) {
        posA = cv.take<int32_t>(maxb); posB = cv.take<int32_t>(maxb); flags = cv.take<int32_t>(maxb); deg = cv.take<int32_t>(maxb + 1); hpos = cv.take<int32_t>(maxb + 1);
        bsum = cv.take<int32_t>(maxb / SCAN_B + 4); cnts = cv.take<int32_t>(8 * (GM_MAX_GCN + 1));
    };
    int32_t *posA, *posB, *flags, *deg, *hpos, *bsum, *cnts;
    carve_tmp(ts, posA, posB, flags, deg, hpos, bsum, cnts);
    GM_TRY(gm_dev_alloc(&tmp, ts.used + 256, s));
    struct TmpGuard { void* p; hipStream_t s; ~TmpGuard() { gm_dev_free(p, s); } } tmp_guard{tmp, s};      // (stream-ordered free on every way out)
    { ConeCarver cv(tmp); carve_tmp(cv, posA, posB, flags, deg, hpos, bsum, cnts); }
    // device-side counts, per level l: [0] n, [1] nnz (edges into l, by destination), [2] the same counted by source, [3] / [4] hub rows; cnts[8 L + 5] = bad
    auto cnt = [&](int l, int k) { return cnts + 8 * l + k; };
    gm_stager sg(s);
    int rc = GM_OK;
    auto fail = [&](int r) { return r; };
    if (hipMemsetAsync(cnts, 0, 4 * 8 * (GM_MAX_GCN + 1), s) != hipSuccess) { gm_set_error("cone: memset failed"); return fail(GM_EHIP); }
    // ---- level L: the centres, in centre order
    gm_cone_level& top = c->lv[L];
    top.n = b->n_c; top.nnz = 0;
    if (hipMemcpyAsync(top.d_row, b->d_crow, 4 * (size_t)top.n, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(top.d_norm, b->d_cnorm, 4 * (size_t)top.n, hipMemcpyDeviceToDevice, s) != hipSuccess) { gm_set_error("cone: copy failed"); return fail(GM_EHIP); }
    top.h_set_off.resize(sets + 1);
    for (int t = 0; t <= sets; ++t) top.h_set_off[t] = b->h_set_sub_off[t] * b->centres;
    if ((rc = sg.upload(top.d_set_off, top.h_set_off)) != GM_OK || (rc = level_tables(top, sets, sg)) != GM_OK) return fail(rc);
    { const int32_t n32 = top.n; if ((rc = sg.upload(cnt(L, 0), &n32, 4)) != GM_OK) return fail(rc); }
    const int fill_blocks = (int)std::min<int64_t>(2048, (rows + 255) / 256);
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks), dim3(256), 0, s, posA, rows, -1);
    if (top.n > 0) {
        hipLaunchKernelGGL(k_scatter_pos, dim3((top.n + 255) / 256), dim3(256), 0, s, top.d_row, top.n, posA);
        hipLaunchKernelGGL(k_check_pos, dim3((top.n + 255) / 256), dim3(256), 0, s, top.d_row, top.n, posA, cnt(L, 5));
    }
    // ---- levels L-1 .. 0
    for (int l = L - 1; l >= 0; --l) {
        gm_cone_level& up = c->lv[l + 1]; gm_cone_level& lo = c->lv[l];
        const int bu = (int)std::max<int64_t>(Bn[l + 1], 1), bl = (int)std::max<int64_t>(Bn[l], 1);
        GM_HIP(hipMemsetAsync(flags, 0, 4 * (size_t)rows, s));
        GM_HIP(hipMemsetAsync(deg, 0, 4 * ((size_t)bu + 1), s));
        hipLaunchKernelGGL(k_mark, dim3((bu + 3) / 4), dim3(256), 0, s, up.d_row, cnt(l + 1, 0), b->d_indptr, b->d_indices, flags, deg);
        dev_scan(flags, posB, rows, 0, cnt(l, 0), bsum, s);                                  // row -> compact id of level l; its size
        dev_scan(deg, up.d_indptr, bu, 1, cnt(l + 1, 1), bsum, s);                           // forward CSR bounds of level l + 1; nnz
        hipLaunchKernelGGL(k_set_off, dim3((sets + 256) / 256), dim3(256), 0, s, posB, b->d_set_row_off, sets, rows, cnt(l, 0), lo.d_set_off);
        hipLaunchKernelGGL(k_level_rows, dim3(fill_blocks), dim3(256), 0, s, flags, posB, rows, b->d_norm, b->d_feat_row, lo.d_row, lo.d_norm, lo.d_feat_row);
        // forward CSR (by destination) and backward CSR (by source)
        hipLaunchKernelGGL(k_fill_in, dim3((bu + 3) / 4), dim3(256), 0, s, up.d_row, cnt(l + 1, 0), b->d_indptr, b->d_indices, posB, up.d_indptr, up.d_indices);
        GM_HIP(hipMemsetAsync(deg, 0, 4 * ((size_t)bl + 1), s));
        hipLaunchKernelGGL(k_out_edges, dim3((bl + 3) / 4), dim3(256), 0, s, lo.d_row, cnt(l, 0), b->d_indptr_t, b->d_indices_t, posA, deg, (const int32_t*)nullptr, (int32_t*)nullptr);
        dev_scan(deg, up.d_indptr_t, bl, 1, cnt(l + 1, 2), bsum, s);
        hipLaunchKernelGGL(k_out_edges, dim3((bl + 3) / 4), dim3(256), 0, s, lo.d_row, cnt(l, 0), b->d_indptr_t, b->d_indices_t, posA, (int32_t*)nullptr, up.d_indptr_t, up.d_indices_t);
        // hub rows of both CSRs, ascending
        for (int o = 0; o < 2; ++o) {
            const int bound = o ? bl : bu;
            hipLaunchKernelGGL(k_heavy_flag, dim3((bound + 255) / 256), dim3(256), 0, s, o ? up.d_indptr_t : up.d_indptr, cnt(o ? l : l + 1, 0), bound, c->heavy_deg, flags);
            dev_scan(flags, hpos, bound, 0, cnt(l + 1, 3 + o), bsum, s);
            hipLaunchKernelGGL(k_heavy_scatter, dim3((bound + 255) / 256), dim3(256), 0, s, flags, hpos, bound, (int)hcap(Be[l + 1]), up.d_heavy[o]);
        }
        GM_HIP(hipGetLastError());
        std::swap(posA, posB);
    }
    // ---- the one round trip: counts and per-level set offsets
    const int32_t* h_cnt = sg.download(cnts, (size_t)8 * (GM_MAX_GCN + 1));
    const int32_t* h_off[GM_MAX_GCN + 1] = {};
    for (int l = 0; l < L; ++l) h_off[l] = sg.download(c->lv[l].d_set_off, (size_t)sets + 1);
    bool okd = h_cnt != nullptr;
    for (int l = 0; l < L; ++l) okd = okd && h_off[l];
    if (!okd) { gm_set_error("cone: pinned staging failed"); return fail(GM_ENOMEM); }
    if (hipStreamSynchronize(s) != hipSuccess) { gm_set_error("cone: stream sync failed"); return fail(GM_EHIP); }
    if (h_cnt[8 * L + 5]) { c->ok = false; return GM_OK; }      // two centres on one row (a self pair): callers fall back to the dense schedule
    for (int l = L - 1; l >= 0; --l) {
        gm_cone_level& up = c->lv[l + 1]; gm_cone_level& lo = c->lv[l];
        lo.n = h_cnt[8 * l]; up.nnz = h_cnt[8 * (l + 1) + 1];
        if (lo.n > Bn[l] || up.nnz > Be[l + 1] || h_cnt[8 * (l + 1) + 2] != up.nnz) {
            gm_set_error("cone: level %d: %d rows (bound %lld), %d in-edges (bound %lld), %d counted by source (corrupt batch CSR?)", l, lo.n, (long long)Bn[l], up.nnz,
                         (long long)Be[l + 1], h_cnt[8 * (l + 1) + 2]);
            return fail(GM_EHIP);
        }
        for (int o = 0; o < 2; ++o) up.n_heavy[o] = std::min<int32_t>(h_cnt[8 * (l + 1) + 3 + o], (int32_t)hcap(Be[l + 1]));
        lo.h_set_off.assign(h_off[l], h_off[l] + sets + 1);
        if ((rc = level_tables(lo, sets, sg)) != GM_OK) return fail(rc);
    }
    // the tables of the lower levels went up after the round trip: complete before a consumer on ANOTHER stream may use them (a few small copies, no kernels)
    if (hipStreamSynchronize(s) != hipSuccess) { gm_set_error("cone: stream sync failed"); return GM_EHIP; }
    c->ok = true;
    return GM_OK;
}

int gm_batch_cone(const gm_batch* b, int L, hipStream_t s, const gm_cone** out) {
    *out = nullptr;
    GM_REQUIRE(b && L >= 1 && L <= GM_MAX_GCN, GM_EINVAL, "cone: n_gcn=%d outside [1,%d]", L, GM_MAX_GCN);
    if (b->cone[L]) { *out = b->cone[L]; return GM_OK; }
    gm_phase_timer tm("cone");
    gm_cone* c = new gm_cone();
    const int rc = cone_build(b, L, s, c);
    if (rc != GM_OK) { gm_cone_free(c, s); return rc; }
    b->cone[L] = c;
    *out = c;
    return GM_OK;
}

extern "C" int gm_batch_prepare_cone(const gm_batch_t* b, int32_t n_gcn, void* stream) {
    GM_REQUIRE(b, GM_EINVAL, "prepare_cone: NULL batch");
    const gm_cone* c = nullptr;
    return gm_batch_cone(b, n_gcn, (hipStream_t)stream, &c);
}

extern "C" int gm_batch_cone_dims(const gm_batch_t* b, int32_t n_gcn, int32_t* ok, int64_t* level_rows, int64_t* level_edges) {
    GM_REQUIRE(b && n_gcn >= 1 && n_gcn <= GM_MAX_GCN, GM_EINVAL, "cone_dims: bad arguments");
    const gm_cone* c = b->cone[n_gcn];
    GM_REQUIRE(c, GM_EINVAL, "cone_dims: call gm_batch_prepare_cone first");
    if (ok) *ok = c->ok ? 1 : 0;
    for (int l = 0; l <= n_gcn; ++l) {
        if (level_rows) level_rows[l] = c->ok ? c->lv[l].n : 0;
        if (level_edges) level_edges[l] = c->ok ? c->lv[l].nnz : 0;
    }
    return GM_OK;
}

extern "C" int gm_batch_cone_read(const gm_batch_t* b, int32_t n_gcn, int32_t level, int32_t what, void* host, int64_t host_bytes) {
    GM_REQUIRE(b && host && n_gcn >= 1 && n_gcn <= GM_MAX_GCN && level >= 0 && level <= n_gcn, GM_EINVAL, "cone_read: bad arguments");
    const gm_cone* c = b->cone[n_gcn];
    GM_REQUIRE(c && c->ok, GM_EINVAL, "cone_read: no cone for n_gcn=%d", n_gcn);
    const gm_cone_level& v = c->lv[level];
    const int n_lo = level > 0 ? c->lv[level - 1].n : 0;
    const void* p = nullptr; int64_t bytes = 0;
    switch (what) {
        case 0: p = v.d_row; bytes = 4ll * v.n; break;
        case 1: p = v.d_indptr; bytes = level > 0 ? 4ll * (v.n + 1) : 0; break;
        case 2: p = v.d_indices; bytes = 4ll * v.nnz; break;
        case 3: p = v.d_indptr_t; bytes = level > 0 ? 4ll * (n_lo + 1) : 0; break;
        case 4: p = v.d_indices_t; bytes = 4ll * v.nnz; break;
        case 5: p = v.d_set_off; bytes = 4ll * (b->sets + 1); break;
        default: gm_set_error("cone_read: unknown field %d", what); return GM_EINVAL;
    }
    GM_REQUIRE(host_bytes >= bytes, GM_EINVAL, "cone_read: host buffer too small (%lld < %lld)", (long long)host_bytes, (long long)bytes);
    if (bytes > 0) GM_HIP(hipMemcpy(host, p, (size_t)bytes, hipMemcpyDeviceToHost));
    return GM_OK;
}
