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
__global__ __launch_bounds__(SCAN_T) void k_scan_top(int32_t* bsum, int nb) {
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
    if (threadIdx.x == 0) bsum[nb] = carry;
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

// exclusive scan of n ints; *total (host) = sum.  Synchronises the stream.
static int dev_scan(const int32_t* in, int32_t* out, int64_t n, int tail, int32_t* total, hipStream_t s) {
    if (n <= 0) { if (total) *total = 0; if (tail) GM_HIP(hipMemsetAsync(out, 0, 4, s)); return GM_OK; }
    const int nb = (int)((n + SCAN_B - 1) / SCAN_B);
    int32_t* bsum = nullptr;
    GM_TRY(gm_alloc(&bsum, (size_t)nb + 1, s));
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(SCAN_T), 0, s, in, n, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_T), 0, s, bsum, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_T), 0, s, in, n, bsum, out, tail);
    int32_t tot = 0;
    hipError_t e = hipMemcpyAsync(&tot, bsum + nb, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    gm_dev_free(bsum, s);
    if (e != hipSuccess) { gm_set_error("cone: scan failed: %s", hipGetErrorString(e)); return GM_EHIP; }
    if (total) *total = tot;
    return GM_OK;
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
// one wave per upper-level row: flag the sources of its in-edges, record its in-degree
__global__ __launch_bounds__(256) void k_mark(const int32_t* up_row, int n_up, const int32_t* indptr, const int32_t* indices, int32_t* flags, int32_t* deg) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_up) return;
    const int r = up_row[q], e0 = indptr[r], e1 = indptr[r + 1];
    for (int e = e0 + lane; e < e1; e += 64) flags[indices[e]] = 1;
    if (lane == 0) deg[q] = e1 - e0;
}
__global__ void k_set_off(const int32_t* scan, const int32_t* set_row_off, int sets, int64_t rows, int32_t total, int32_t* set_off) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > sets) return;
    const int64_t r = set_row_off[t];
    set_off[t] = (t == sets || r >= rows) ? total : scan[r];
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
__global__ __launch_bounds__(256) void k_fill_in(const int32_t* up_row, int n_up, const int32_t* indptr, const int32_t* indices, const int32_t* pos_lo,
                                                 const int32_t* cptr, int32_t* cidx) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_up) return;
    const int r = up_row[q], e0 = indptr[r], n = indptr[r + 1] - e0, o = cptr[q];
    for (int j = lane; j < n; j += 64) cidx[o + j] = pos_lo[indices[e0 + j]];
}
// backward CSR: out-edges of a lower-level row that end in the upper level, in the batch's by-source order.
// pass 0 counts, pass 1 fills (order preserved with a ballot prefix: deterministic).
__global__ __launch_bounds__(256) void k_out_edges(const int32_t* lo_row, int n_lo, const int32_t* indptr_t, const int32_t* indices_t, const int32_t* pos_up,
                                                   int32_t* cnt, const int32_t* tptr, int32_t* tidx) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_lo) return;
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
__global__ void k_find_heavy_c(const int32_t* indptr, int n, int32_t* list, int32_t* count, int cap, int thr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && indptr[r + 1] - indptr[r] > thr) { const int k = atomicAdd(count, 1); if (k < cap) list[k] = r; }
}

// ------------------------------------------------------------------------------------------ host
void gm_cone_free(gm_cone* c, hipStream_t s) {
    if (!c) return;
    for (int l = 0; l <= GM_MAX_GCN; ++l) {
        gm_cone_level& v = c->lv[l];
        gm_dev_free(v.d_row, s); gm_dev_free(v.d_norm, s); gm_dev_free(v.d_feat_row, s); gm_dev_free(v.d_set_off, s);
        gm_dev_free(v.d_tiles, s); gm_dev_free(v.d_chunks, s); gm_dev_free(v.d_set_chunk_off, s);
        gm_dev_free(v.d_indptr, s); gm_dev_free(v.d_indices, s); gm_dev_free(v.d_indptr_t, s); gm_dev_free(v.d_indices_t, s);
        gm_dev_free(v.d_heavy[0], s); gm_dev_free(v.d_heavy[1], s);
    }
    delete c;
}

static int upload(int32_t** d, const std::vector<int32_t>& v, hipStream_t s) {
    GM_TRY(gm_alloc(d, v.size(), s));
    if (!v.empty()) GM_HIP(hipMemcpyAsync(*d, v.data(), 4 * v.size(), hipMemcpyHostToDevice, s));
    return GM_OK;
}

// GEMM row tiles and weight-gradient chunks of one level (never straddling two sets: each set has its own weights)
static int level_tables(gm_cone_level& v, int sets, hipStream_t s) {
    std::vector<int32_t> tiles, chunks, coff(sets + 1, 0);
    const int64_t cr = gm_wgrad_chunk_rows(v.h_set_off);
    for (int t = 0; t < sets; ++t) {
        const int r0 = v.h_set_off[t], r1 = v.h_set_off[t + 1];
        for (int r = r0; r < r1; r += GM_GEMM_BM) { tiles.push_back(t); tiles.push_back(r); tiles.push_back(std::min(GM_GEMM_BM, r1 - r)); }
        for (int r = r0; r < r1; r += (int)cr) { chunks.push_back(t); chunks.push_back(r); chunks.push_back(std::min<int>((int)cr, r1 - r)); }
        coff[t + 1] = (int32_t)(chunks.size() / 3);
    }
    v.n_tiles = (int32_t)(tiles.size() / 3); v.n_chunks = (int32_t)(chunks.size() / 3);
    GM_TRY(upload(&v.d_tiles, tiles, s)); GM_TRY(upload(&v.d_chunks, chunks, s)); GM_TRY(upload(&v.d_set_chunk_off, coff, s));
    GM_HIP(hipStreamSynchronize(s));
    return GM_OK;
}

static int heavy_list(const int32_t* indptr, int n, int nnz, int thr, int32_t** list, int32_t* count, hipStream_t s) {
    *count = 0;
    const int cap = nnz / thr + 1;
    int32_t* d_cnt = nullptr;
    GM_TRY(gm_alloc(list, cap, s)); GM_TRY(gm_alloc(&d_cnt, 1, s));
    GM_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
    if (n > 0) hipLaunchKernelGGL(k_find_heavy_c, dim3((n + 255) / 256), dim3(256), 0, s, indptr, n, *list, d_cnt, cap, thr);
    int32_t c = 0;
    GM_HIP(hipMemcpyAsync(&c, d_cnt, 4, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    gm_dev_free(d_cnt, s);
    c = std::min(c, cap);
    if (c > 1) {                          // deterministic order
        std::vector<int32_t> h(c);
        GM_HIP(hipMemcpyAsync(h.data(), *list, 4 * (size_t)c, hipMemcpyDeviceToHost, s));      // stream-ordered: a plain hipMemcpy would serialise with the default stream
        GM_HIP(hipStreamSynchronize(s));
        std::sort(h.begin(), h.end());
        GM_HIP(hipMemcpyAsync(*list, h.data(), 4 * (size_t)c, hipMemcpyHostToDevice, s));
        GM_HIP(hipStreamSynchronize(s));
    }
    *count = c;
    return GM_OK;
}

static int cone_build(const gm_batch* b, int L, hipStream_t s, gm_cone* c, int32_t* posA, int32_t* posB, int32_t* flags, int32_t* d_bad) {
    const int64_t rows = b->rows; const int sets = b->sets;
    const int fill_blocks = (int)std::min<int64_t>(2048, (rows + 255) / 256);
    c->L = L; c->heavy_deg = gm_heavy_deg();
    // ---- level L: the centres, in centre order
    gm_cone_level& top = c->lv[L];
    top.n = b->n_c;
    GM_TRY(gm_alloc(&top.d_row, top.n, s)); GM_TRY(gm_alloc(&top.d_norm, top.n, s));
    GM_HIP(hipMemcpyAsync(top.d_row, b->d_crow, 4 * (size_t)top.n, hipMemcpyDeviceToDevice, s));
    GM_HIP(hipMemcpyAsync(top.d_norm, b->d_cnorm, 4 * (size_t)top.n, hipMemcpyDeviceToDevice, s));
    top.h_set_off.resize(sets + 1);
    for (int t = 0; t <= sets; ++t) top.h_set_off[t] = b->h_set_sub_off[t] * b->centres;
    GM_TRY(upload(&top.d_set_off, top.h_set_off, s));
    GM_TRY(level_tables(top, sets, s));
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks), dim3(256), 0, s, posA, rows, -1);
    if (top.n > 0) hipLaunchKernelGGL(k_scatter_pos, dim3((top.n + 255) / 256), dim3(256), 0, s, top.d_row, top.n, posA);
    GM_HIP(hipMemsetAsync(d_bad, 0, 4, s));
    if (top.n > 0) hipLaunchKernelGGL(k_check_pos, dim3((top.n + 255) / 256), dim3(256), 0, s, top.d_row, top.n, posA, d_bad);
    int32_t bad = 0;
    GM_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    if (bad) { c->ok = false; return GM_OK; }      // two centres on one row (a self pair): callers fall back to the dense schedule
    // ---- levels L-1 .. 0
    for (int l = L - 1; l >= 0; --l) {
        gm_cone_level& up = c->lv[l + 1]; gm_cone_level& lo = c->lv[l];
        int32_t* deg = nullptr;
        GM_TRY(gm_alloc(&deg, (size_t)up.n + 1, s));
        GM_HIP(hipMemsetAsync(flags, 0, 4 * (size_t)rows, s));
        if (up.n > 0) hipLaunchKernelGGL(k_mark, dim3((up.n + 3) / 4), dim3(256), 0, s, up.d_row, up.n, b->d_indptr, b->d_indices, flags, deg);
        int32_t n_lo = 0, nnz = 0;
        int rc = dev_scan(flags, posB, rows, 0, &n_lo, s);
        if (rc == GM_OK) { GM_TRY(gm_alloc(&up.d_indptr, (size_t)up.n + 1, s)); rc = dev_scan(deg, up.d_indptr, up.n, 1, &nnz, s); }
        gm_dev_free(deg, s);
        GM_TRY(rc);
        lo.n = n_lo; up.nnz = nnz;
        GM_TRY(gm_alloc(&lo.d_row, lo.n, s)); GM_TRY(gm_alloc(&lo.d_norm, lo.n, s)); GM_TRY(gm_alloc(&lo.d_set_off, sets + 1, s));
        if (l == 0) GM_TRY(gm_alloc(&lo.d_feat_row, lo.n, s));
        hipLaunchKernelGGL(k_set_off, dim3((sets + 256) / 256), dim3(256), 0, s, posB, b->d_set_row_off, sets, rows, n_lo, lo.d_set_off);
        hipLaunchKernelGGL(k_level_rows, dim3(fill_blocks), dim3(256), 0, s, flags, posB, rows, b->d_norm, b->d_feat_row, lo.d_row, lo.d_norm, lo.d_feat_row);
        lo.h_set_off.resize(sets + 1);
        GM_HIP(hipMemcpyAsync(lo.h_set_off.data(), lo.d_set_off, 4 * (size_t)(sets + 1), hipMemcpyDeviceToHost, s));
        GM_HIP(hipStreamSynchronize(s));
        GM_TRY(level_tables(lo, sets, s));
        // forward CSR (by destination) and backward CSR (by source)
        GM_TRY(gm_alloc(&up.d_indices, nnz, s)); GM_TRY(gm_alloc(&up.d_indices_t, nnz, s)); GM_TRY(gm_alloc(&up.d_indptr_t, (size_t)lo.n + 1, s));
        if (up.n > 0) hipLaunchKernelGGL(k_fill_in, dim3((up.n + 3) / 4), dim3(256), 0, s, up.d_row, up.n, b->d_indptr, b->d_indices, posB, up.d_indptr, up.d_indices);
        int32_t* cnt = nullptr;
        GM_TRY(gm_alloc(&cnt, (size_t)lo.n + 1, s));
        if (lo.n > 0) hipLaunchKernelGGL(k_out_edges, dim3((lo.n + 3) / 4), dim3(256), 0, s, lo.d_row, lo.n, b->d_indptr_t, b->d_indices_t, posA, cnt, (const int32_t*)nullptr, (int32_t*)nullptr);
        int32_t nnz_t = 0;
        rc = dev_scan(cnt, up.d_indptr_t, lo.n, 1, &nnz_t, s);
        gm_dev_free(cnt, s);
        GM_TRY(rc);
        GM_REQUIRE(nnz_t == nnz, GM_EHIP, "cone: level %d has %d in-edges but %d out-edges (corrupt batch CSR?)", l + 1, nnz, nnz_t);
        if (lo.n > 0) hipLaunchKernelGGL(k_out_edges, dim3((lo.n + 3) / 4), dim3(256), 0, s, lo.d_row, lo.n, b->d_indptr_t, b->d_indices_t, posA, (int32_t*)nullptr, up.d_indptr_t, up.d_indices_t);
        GM_HIP(hipGetLastError());
        GM_TRY(heavy_list(up.d_indptr, up.n, nnz, c->heavy_deg, &up.d_heavy[0], &up.n_heavy[0], s));
        GM_TRY(heavy_list(up.d_indptr_t, lo.n, nnz, c->heavy_deg, &up.d_heavy[1], &up.n_heavy[1], s));
        std::swap(posA, posB);
    }
    c->ok = true;
    return GM_OK;
}

int gm_batch_cone(const gm_batch* b, int L, hipStream_t s, const gm_cone** out) {
    *out = nullptr;
    GM_REQUIRE(b && L >= 1 && L <= GM_MAX_GCN, GM_EINVAL, "cone: n_gcn=%d outside [1,%d]", L, GM_MAX_GCN);
    if (b->cone[L]) { *out = b->cone[L]; return GM_OK; }
    gm_phase_timer tm("cone");
    gm_cone* c = new gm_cone();
    int32_t *posA = nullptr, *posB = nullptr, *flags = nullptr, *d_bad = nullptr;
    int rc = gm_alloc(&posA, b->rows, s);
    if (rc == GM_OK) rc = gm_alloc(&posB, b->rows, s);
    if (rc == GM_OK) rc = gm_alloc(&flags, b->rows, s);
    if (rc == GM_OK) rc = gm_alloc(&d_bad, 1, s);
    if (rc == GM_OK) rc = cone_build(b, L, s, c, posA, posB, flags, d_bad);
    if (rc == GM_OK && hipStreamSynchronize(s) != hipSuccess) { gm_set_error("cone: stream sync failed"); rc = GM_EHIP; }
    gm_dev_free(posA, s); gm_dev_free(posB, s); gm_dev_free(flags, s); gm_dev_free(d_bad, s);
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
