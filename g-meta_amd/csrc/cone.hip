// Receptive-field ("cone") tables of a batch, for gm_hparams_t.cone.
//
// Only the centre rows of the last GCN layer reach the head (learner.py:159-170), so layer L is needed at the
// centres, layer L-1 at the in-neighbours of the centres, ... and the input features at the rows L hops upstream.
// Level l (0..L) is the ascending list of batch rows whose layer-l activation is needed (level L = the centres in
// centre order); the edges between consecutive levels are kept as two compact CSRs (by destination for the forward,
// by source for the backward) whose column ids index the neighbouring level.  Every sum the dense schedule forms for
// a needed row is formed here with the same terms in the same order; rows outside the cone are never computed.
#include <algorithm>
#include <memory>
#include <vector>
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

// Single-pass chained scans (decoupled look-back): one launch per scan instead of three (block sums, their scan, the final pass).  A meta-batch's
// table build is a chain of ~60 dependent launches of a few microseconds each beside a running meta-step: its length in LAUNCHES is what it costs.
// Blocks take their position from a ticket (so a block only ever waits for blocks that started before it), publish {generation, state, value} as
// one 64-bit word and walk back over their predecessors until one has published an inclusive prefix.  The status words are zeroed once per build;
// every scan of the build uses its own generation and ticket, so stale words are never mistaken for this scan's.
struct ScanChain { unsigned long long* st; int32_t* ticket; int gen; };
#define SCAN_AGG 1ull
#define SCAN_INC 2ull

__device__ __forceinline__ int chain_block_id(const ScanChain& c, int* sh) {
    if (threadIdx.x == 0) *sh = atomicAdd(&c.ticket[c.gen], 1);
    __syncthreads();
    const int bid = *sh;
    __syncthreads();
    return bid;
}
// exclusive prefix of block `bid` given its total (every thread calls; the result is broadcast through *sh)
__device__ __forceinline__ int chain_prefix(const ScanChain& c, int bid, int tot, int* sh) {
    if (threadIdx.x == 0) {
        const unsigned long long tag = (unsigned long long)c.gen << 34;
        int p = 0;
        if (bid > 0) {
            __hip_atomic_store(&c.st[bid], tag | (SCAN_AGG << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int j = bid - 1;; --j) {
                unsigned long long v;
                while (true) {
                    v = __hip_atomic_load(&c.st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 34) == (unsigned long long)c.gen && ((v >> 32) & 3ull)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                p += (int)(unsigned)v;
                if (((v >> 32) & 3ull) == SCAN_INC) break;
            }
        }
        __hip_atomic_store(&c.st[bid], tag | (SCAN_INC << 32) | (unsigned)(p + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sh = p;
    }
    __syncthreads();
    const int p = *sh;
    __syncthreads();
    return p;
}

// out[i] = exclusive prefix of in[0..n); out[n] = total when tail != 0; *total (optional) = the sum.  in == out is allowed.
__global__ __launch_bounds__(SCAN_T) void k_scan(const int32_t* in, int n, int32_t* out, int tail, int32_t* total, ScanChain c) {
    __shared__ int sm[SCAN_T]; __shared__ int sh;
    const int bid = chain_block_id(c, &sh);
    const int base = bid * SCAN_B + threadIdx.x * SCAN_I;
    int v[SCAN_I], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    int tot;
    const int ex = block_excl_scan(s, sm, &tot);
    int run = chain_prefix(c, bid, tot, &sh) + ex;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (bid == (int)gridDim.x - 1 && threadIdx.x == SCAN_T - 1) { if (tail) out[n] = run; if (total) *total = run; }
}
static void dev_scan(const int32_t* in, int32_t* out, int64_t n, int tail, int32_t* d_total, ScanChain& c, hipStream_t s) {
    const int nb = (int)std::max<int64_t>(1, (n + SCAN_B - 1) / SCAN_B);
    ++c.gen;
    hipLaunchKernelGGL(k_scan, dim3(nb), dim3(SCAN_T), 0, s, in, (int)n, out, tail, d_total, c);
}

// ------------------------------------------------------------------------------------------ build kernels
__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) p[k] = v;
}
// pos[row[k]] = k ; afterwards k_check_pos counts the centres that lost their slot (two centres on one row)
__global__ void k_scatter_pos(const int32_t* row, int n, int32_t* pos, int32_t* n_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) *n_out = n;
    if (k < n) pos[row[k]] = k;
}
__global__ void k_check_pos(const int32_t* row, int n, const int32_t* pos, int32_t* bad) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && pos[row[k]] != k) atomicAdd(bad, 1);
}
// Round 6: the builder knows the level sizes only as UPPER BOUNDS while it queues its kernels (one host round trip at the very end instead of ~10 per
// level): grids are sized by the bound, the actual count is read from device memory (n_ptr).
// one wave per upper-level row: flag the sources of its in-edges, record its in-degree
__global__ __launch_bounds__(256) void k_mark(const int32_t* up_row, const int32_t* n_ptr, int bound, const int32_t* indptr, const int32_t* indices, int32_t* flags, int32_t* deg) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= *n_ptr) { if (lane == 0 && q < bound) deg[q] = 0; return; }      // (the scan that follows runs over the bound)
    const int r = up_row[q], e0 = indptr[r], e1 = indptr[r + 1];
    for (int e = e0 + lane; e < e1; e += 64) flags[indices[e]] = 1;
    if (lane == 0) deg[q] = e1 - e0;
}
// set_off[t] = level rows below the first row of set t (the level's row list is ascending); a copy goes to the build's download area
__global__ void k_set_off(const int32_t* lrow, const int32_t* n_ptr, const int32_t* set_row_off, int sets, int64_t rows, int32_t* set_off, int32_t* copy) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > sets) return;
    const int n = *n_ptr;
    const int64_t r = set_row_off[t];
    int lo = 0, hi = n;
    if (t == sets || r >= rows) lo = n;
    else while (lo < hi) { const int mid = (lo + hi) >> 1; if (lrow[mid] < r) lo = mid + 1; else hi = mid; }
    set_off[t] = lo; copy[t] = lo;
}
// compact the flagged rows (ascending) in one chained pass; pos[] becomes the row -> compact id map (-1 outside the level), *total the level's size
__global__ __launch_bounds__(SCAN_T) void k_level_rows(const int32_t* flags, int32_t* pos, int rows, const float* norm, const int32_t* feat_row,
                                                       int32_t* lrow, float* lnorm, int32_t* lfeat, int32_t* total, ScanChain c) {
    __shared__ int sm[SCAN_T]; __shared__ int sh;
    const int bid = chain_block_id(c, &sh);
    const int base = bid * SCAN_B + threadIdx.x * SCAN_I;
    int v[SCAN_I], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { v[i] = (base + i < rows && flags[base + i]) ? 1 : 0; s += v[i]; }
    int tot;
    const int ex = block_excl_scan(s, sm, &tot);
    int run = chain_prefix(c, bid, tot, &sh) + ex;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        const int r = base + i;
        if (r < rows) {
            if (v[i]) {
                lrow[run] = r; lnorm[run] = norm[r];
                if (lfeat) lfeat[run] = feat_row[r];
                pos[r] = run;
            } else pos[r] = -1;
        }
        run += v[i];
    }
    if (bid == (int)gridDim.x - 1 && threadIdx.x == SCAN_T - 1) *total = run;
}
// forward CSR of the upper level: every in-edge of an upper row, sources renamed to compact ids of the lower level
// ... and the out-degree of every lower row into the upper level (integer atomics: the counts do not depend on their order), which sizes the backward CSR
__global__ __launch_bounds__(256) void k_fill_in(const int32_t* up_row, const int32_t* n_ptr, const int32_t* indptr, const int32_t* indices, const int32_t* pos_lo,
                                                 const int32_t* cptr, int32_t* cidx, int32_t* deg_t) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= *n_ptr) return;
    const int r = up_row[q], e0 = indptr[r], n = indptr[r + 1] - e0, o = cptr[q];
    for (int j = lane; j < n; j += 64) { const int u = pos_lo[indices[e0 + j]]; cidx[o + j] = u; atomicAdd(&deg_t[u], 1); }
}
// backward CSR: out-edges of a lower-level row that end in the upper level, in the batch's by-source order (order preserved with a ballot prefix:
// deterministic); the row bounds tptr come from the scan of k_fill_in's counts.
__global__ __launch_bounds__(256) void k_out_edges(const int32_t* lo_row, const int32_t* n_ptr, const int32_t* indptr_t, const int32_t* indices_t, const int32_t* pos_up,
                                                   const int32_t* tptr, int32_t* tidx) {
    // eight lanes per lower-level row, eight rows per wave (round 6: a whole wave per row before -- 940 k waves for the ~2 out-edges of a level-0 row of the
    // arxiv query batch, 59 us a launch)
    const int lane = threadIdx.x & 63, grp = lane >> 3, gl = lane & 7;
    const int p = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool have = p < *n_ptr;
    int e0 = 0, e1 = 0;
    if (have) { const int u = lo_row[p]; e0 = indptr_t[u]; e1 = indptr_t[u + 1]; }
    int base = have ? tptr[p] : 0;
    for (int eb = e0; __any(eb < e1); eb += 8) {
        const int e = eb + gl;
        const int q = e < e1 ? pos_up[indices_t[e]] : -1;
        const unsigned m = (unsigned)(__ballot(q >= 0) >> (grp * 8)) & 0xffu;
        if (q >= 0) tidx[base + __popc(m & ((1u << gl) - 1u))] = q;
        base += __popc(m);
    }
}
// hub rows of a compact CSR (in-degree above thr, r < *n_ptr), ascending, by one chained compaction; *total = their number (the list keeps the first `cap`)
__global__ __launch_bounds__(SCAN_T) void k_heavy_rows(const int32_t* indptr, const int32_t* n_ptr, int thr, int cap, int32_t* list, int32_t* total, ScanChain c) {
    __shared__ int sm[SCAN_T]; __shared__ int sh;
    const int bid = chain_block_id(c, &sh);
    const int base = bid * SCAN_B + threadIdx.x * SCAN_I, n = *n_ptr;
    int v[SCAN_I], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { v[i] = (base + i < n && indptr[base + i + 1] - indptr[base + i] > thr) ? 1 : 0; s += v[i]; }
    int tot;
    const int ex = block_excl_scan(s, sm, &tot);
    int run = chain_prefix(c, bid, tot, &sh) + ex;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { if (v[i] && run < cap) list[run] = base + i; run += v[i]; }
    if (bid == (int)gridDim.x - 1 && threadIdx.x == SCAN_T - 1) *total = run;
}

// ------------------------------------------------------------------------------------------ host
void gm_cone_free(gm_cone* c, hipStream_t s) {
    if (!c) return;
    (void)s;                        // (the level arrays live in the batch's slabs: batch_free releases them)
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

// GEMM row tiles and weight-gradient chunks of one level (never straddling two sets: each set has its own weights), in ONE upload through pinned
// staging: [set offsets (with_set_off: the top level's are known on the host) | chunk offsets per set | tiles | chunks], each part 64-int aligned in d_tab
static int level_tables(gm_cone_level& v, int sets, gm_stager& sg, int32_t* d_tab, size_t tab_cap, bool with_set_off) {
    auto pad = [](size_t n) { return (n + 63) / 64 * 64; };
    std::vector<int32_t> h;
    const size_t o_soff = 0, o_coff = pad(sets + 1);
    h.resize(2 * pad(sets + 1), 0);
    for (int t = 0; t <= sets; ++t) h[o_soff + t] = v.h_set_off[t];
    const int64_t cr = gm_wgrad_chunk_rows(v.h_set_off);
    std::vector<int32_t> tiles, chunks;
    for (int t = 0; t < sets; ++t) {
        const int r0 = v.h_set_off[t], r1 = v.h_set_off[t + 1];
        for (int r = r0; r < r1; r += GM_GEMM_BM) { tiles.push_back(t); tiles.push_back(r); tiles.push_back(std::min(GM_GEMM_BM, r1 - r)); }
        for (int r = r0; r < r1; r += (int)cr) { chunks.push_back(t); chunks.push_back(r); chunks.push_back(std::min<int>((int)cr, r1 - r)); }
        h[o_coff + t + 1] = (int32_t)(chunks.size() / 3);
    }
    v.n_tiles = (int32_t)(tiles.size() / 3); v.n_chunks = (int32_t)(chunks.size() / 3);
    const size_t o_tiles = h.size(), o_chunks = o_tiles + pad(tiles.size());
    GM_REQUIRE(o_chunks + pad(chunks.size()) <= tab_cap, GM_ERANGE, "cone: level tables (%zu ints) exceed their bound (%zu)", o_chunks + pad(chunks.size()), tab_cap);
    h.resize(o_chunks + pad(chunks.size()), 0);
    std::copy(tiles.begin(), tiles.end(), h.begin() + o_tiles); std::copy(chunks.begin(), chunks.end(), h.begin() + o_chunks);
    v.d_set_off = d_tab + o_soff; v.d_set_chunk_off = d_tab + o_coff; v.d_tiles = d_tab + o_tiles; v.d_chunks = d_tab + o_chunks;
    const size_t skip = with_set_off ? 0 : o_coff;        // (a lower level's set offsets were written by the device)
    return sg.upload(d_tab + skip, h.data() + skip, (h.size() - skip) * sizeof(int32_t));
}

// launch(): everything up to the queued downloads; the caller synchronises the stream ONCE (for one build or for the two of a meta-batch:
// gm_batch_prepare_cone_pair); finish(): counts, the lower levels' GEMM / weight-gradient tables.
struct ConeBuild {
    const gm_batch* b; int L; hipStream_t s; gm_cone* c;
    int64_t Bn[GM_MAX_GCN + 1], Be[GM_MAX_GCN + 1];               // row bound of level l; bound of the edges from level l - 1 into level l
    int32_t* d_tab[GM_MAX_GCN + 1] = {}; size_t tab_cap[GM_MAX_GCN + 1] = {};      // per level: set offsets | chunk offsets | tiles | chunks (level_tables)
    char* tmp = nullptr; size_t tmp_cap = 0;                      // scratch of the build: position maps, flags, degree / count arrays, scan status words, the device-side counts
    gm_stager sg;
    const int32_t* h_cnt = nullptr;                               // the download: counts, then the lower levels' set offsets
    ConeBuild(const gm_batch* b_, int L_, hipStream_t s_, gm_cone* c_) : b(b_), L(L_), s(s_), c(c_), sg(s_) {}
    ~ConeBuild() { gm_slab_release(tmp, tmp_cap, s); }            // (behind the kernels that used it, on every way out)
    ConeBuild(const ConeBuild&) = delete;
    ConeBuild& operator=(const ConeBuild&) = delete;
    size_t hcap(int64_t e) const { return (size_t)(e / c->heavy_deg + 1); }
    static size_t pad64(size_t n) { return (n + 63) / 64 * 64; }
    size_t cnt_ints() const { return 8 * (GM_MAX_GCN + 1); }
    int launch();
    int finish();
};

int ConeBuild::launch() {
    const int64_t rows = b->rows; const int sets = b->sets;
    c->L = L; c->heavy_deg = gm_heavy_deg();
    GM_REQUIRE(rows < ((int64_t)1 << 30) && b->edges < ((int64_t)1 << 30), GM_ERANGE, "cone: batch too large");
    Bn[L] = b->n_c; Be[L] = b->n_e1;
    for (int l = L - 1; l >= 0; --l) { Bn[l] = (l == L - 1) ? std::min<int64_t>(rows, b->n_e1) : rows; Be[l] = l > 0 ? b->edges : 0; }
    auto carve = [&](ConeCarver& cv) {
        for (int l = 0; l <= L; ++l) {
            gm_cone_level& v = c->lv[l];
            // (gm_wgrad_chunk_rows returns at least 128 rows per chunk; tiles and chunks never straddle sets: + sets + 1 each)
            tab_cap[l] = 2 * pad64(sets + 1) + pad64((size_t)(Bn[l] / GM_GEMM_BM + sets + 1) * 3) + pad64((size_t)(Bn[l] / 128 + sets + 1) * 3);
            d_tab[l] = cv.take<int32_t>(tab_cap[l]);
            v.d_set_off = d_tab[l];
            if (l < L) { v.d_row = cv.take<int32_t>(Bn[l]); v.d_norm = cv.take<float>(Bn[l]); }
            else { v.d_row = b->d_crow; v.d_norm = b->d_cnorm; }      // level L = the centres, in centre order: the batch's own lists (the cone lives inside the batch)
            v.d_feat_row = l == 0 ? cv.take<int32_t>(Bn[l]) : nullptr;
            if (l > 0) {
                v.d_indptr = cv.take<int32_t>(Bn[l] + 1); v.d_indices = cv.take<int32_t>(Be[l]);
                v.d_indptr_t = cv.take<int32_t>(Bn[l - 1] + 1); v.d_indices_t = cv.take<int32_t>(Be[l]);
                v.d_heavy[0] = cv.take<int32_t>(hcap(Be[l])); v.d_heavy[1] = cv.take<int32_t>(hcap(Be[l]));
            }
        }
    };
    // (from the batch's slabs -- gm_balloc_bytes leaves room for a two-layer build in the first one -- and released with them)
    { ConeCarver size(nullptr); carve(size); GM_TRY(gm_balloc_bytes(const_cast<gm_batch*>(b), &c->slab, size.used + 256, s)); }
    { ConeCarver cv(c->slab); carve(cv); }
    const int64_t maxb = std::max<int64_t>(rows, 1);
    const size_t n_st = (size_t)(maxb / SCAN_B + 2), n_ticket = 8 * (GM_MAX_GCN + 1) + 8, n_soff = (size_t)L * (sets + 1);
    ConeCarver ts(nullptr);
    int32_t *posA, *posB, *flags, *deg, *deg_t, *cnts, *ticket; unsigned long long* st;
    auto carve_tmp = [&](ConeCarver& cv) {
        // zeroed by one memset: [status words | tickets | counts + set-offset copies]; counts + copies come down in one download
        st = cv.take<unsigned long long>(n_st); ticket = cv.take<int32_t>(n_ticket); cnts = cv.take<int32_t>(cnt_ints() + n_soff);
        posA = cv.take<int32_t>(maxb); posB = cv.take<int32_t>(maxb); flags = cv.take<int32_t>(maxb); deg_t = cv.take<int32_t>(maxb + 1); deg = cv.take<int32_t>(maxb + 1);
    };
    carve_tmp(ts);
    GM_TRY(gm_slab_acquire(&tmp, &tmp_cap, ts.used + 256, s));
    { ConeCarver cv(tmp); carve_tmp(cv); }
    const size_t zero_bytes = (size_t)((char*)posA - (char*)st);
    // device-side counts, per level l: [0] n, [1] nnz (edges into l, by destination), [2] the same counted by source, [3] / [4] hub rows; cnts[8 L + 5] = bad
    auto cnt = [&](int l, int k) { return cnts + 8 * l + k; };
    int32_t* soff_copy = cnts + cnt_ints();
    ScanChain chain{st, ticket, 0};
    GM_HIP(hipMemsetAsync(st, 0, zero_bytes, s));
    // ---- level L: the centres, in centre order
    gm_cone_level& top = c->lv[L];
    top.n = b->n_c; top.nnz = 0;
    top.h_set_off.resize(sets + 1);
    for (int t = 0; t <= sets; ++t) top.h_set_off[t] = b->h_set_sub_off[t] * b->centres;
    GM_TRY(level_tables(top, sets, sg, d_tab[L], tab_cap[L], true));
    const int fill_blocks = (int)std::min<int64_t>(2048, (rows + 255) / 256);
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks), dim3(256), 0, s, posA, rows, -1);
    hipLaunchKernelGGL(k_scatter_pos, dim3((top.n + 255) / 256 + 1), dim3(256), 0, s, top.d_row, top.n, posA, cnt(L, 0));
    if (top.n > 0) hipLaunchKernelGGL(k_check_pos, dim3((top.n + 255) / 256), dim3(256), 0, s, top.d_row, top.n, posA, cnt(L, 5));
    // ---- levels L-1 .. 0: ten launches each
    for (int l = L - 1; l >= 0; --l) {
        gm_cone_level& up = c->lv[l + 1]; gm_cone_level& lo = c->lv[l];
        const int bu = (int)std::max<int64_t>(Bn[l + 1], 1), bl = (int)std::max<int64_t>(Bn[l], 1);
        const int row_blocks = (int)std::max<int64_t>(1, (rows + SCAN_B - 1) / SCAN_B);
        GM_HIP(hipMemsetAsync(flags, 0, (size_t)((char*)deg - (char*)flags), s));           // flags and deg_t (adjacent)
        hipLaunchKernelGGL(k_mark, dim3((bu + 3) / 4), dim3(256), 0, s, up.d_row, cnt(l + 1, 0), bu, b->d_indptr, b->d_indices, flags, deg);
        ++chain.gen;                                                                           // the level's rows (ascending), row -> compact id, its size
        hipLaunchKernelGGL(k_level_rows, dim3(row_blocks), dim3(SCAN_T), 0, s, flags, posB, (int)rows, b->d_norm, b->d_feat_row, lo.d_row, lo.d_norm, lo.d_feat_row, cnt(l, 0), chain);
        hipLaunchKernelGGL(k_set_off, dim3((sets + 256) / 256), dim3(256), 0, s, lo.d_row, cnt(l, 0), b->d_set_row_off, sets, rows, lo.d_set_off, soff_copy + (size_t)l * (sets + 1));
        dev_scan(deg, up.d_indptr, bu, 1, cnt(l + 1, 1), chain, s);                            // forward CSR bounds of level l + 1; nnz
        // forward CSR (by destination) and backward CSR (by source)
        hipLaunchKernelGGL(k_fill_in, dim3((bu + 3) / 4), dim3(256), 0, s, up.d_row, cnt(l + 1, 0), b->d_indptr, b->d_indices, posB, up.d_indptr, up.d_indices, deg_t);
        dev_scan(deg_t, up.d_indptr_t, bl, 1, cnt(l + 1, 2), chain, s);
        hipLaunchKernelGGL(k_out_edges, dim3((bl + 31) / 32), dim3(256), 0, s, lo.d_row, cnt(l, 0), b->d_indptr_t, b->d_indices_t, posA, up.d_indptr_t, up.d_indices_t);
        // hub rows of both CSRs, ascending
        for (int o = 0; o < 2; ++o) {
            const int bound = o ? bl : bu;
            ++chain.gen;
            hipLaunchKernelGGL(k_heavy_rows, dim3((bound + SCAN_B - 1) / SCAN_B), dim3(SCAN_T), 0, s, o ? up.d_indptr_t : up.d_indptr, cnt(o ? l : l + 1, 0), c->heavy_deg,
                               (int)hcap(Be[l + 1]), up.d_heavy[o], cnt(l + 1, 3 + o), chain);
        }
        GM_HIP(hipGetLastError());
        std::swap(posA, posB);
    }
    GM_REQUIRE(chain.gen < (int)n_ticket, GM_ERANGE, "cone: %d scans, %zu tickets", chain.gen, n_ticket);
    // ---- the one round trip: counts and per-level set offsets
    h_cnt = sg.download(cnts, cnt_ints() + n_soff);
    if (!h_cnt) { gm_set_error("cone: pinned staging failed"); return GM_ENOMEM; }
    return GM_OK;
}

int ConeBuild::finish() {                // (the stream has passed launch()'s download); queues the lower levels' tables: the caller synchronises once more
    const int sets = b->sets;
    if (h_cnt[8 * L + 5]) { c->ok = false; return GM_OK; }      // two centres on one row (a self pair): callers fall back to the dense schedule
    const int32_t* h_soff = h_cnt + cnt_ints();
    for (int l = L - 1; l >= 0; --l) {
        gm_cone_level& up = c->lv[l + 1]; gm_cone_level& lo = c->lv[l];
        lo.n = h_cnt[8 * l]; up.nnz = h_cnt[8 * (l + 1) + 1];
        if (lo.n > Bn[l] || up.nnz > Be[l + 1] || h_cnt[8 * (l + 1) + 2] != up.nnz) {
            gm_set_error("cone: level %d: %d rows (bound %lld), %d in-edges (bound %lld), %d counted by source (corrupt batch CSR?)", l, lo.n, (long long)Bn[l], up.nnz,
                         (long long)Be[l + 1], h_cnt[8 * (l + 1) + 2]);
            return GM_EHIP;
        }
        for (int o = 0; o < 2; ++o) up.n_heavy[o] = std::min<int32_t>(h_cnt[8 * (l + 1) + 3 + o], (int32_t)hcap(Be[l + 1]));
        lo.h_set_off.assign(h_soff + (size_t)l * (sets + 1), h_soff + (size_t)(l + 1) * (sets + 1));
        GM_TRY(level_tables(lo, sets, sg, d_tab[l], tab_cap[l], false));
    }
    c->ok = true;
    return GM_OK;
}

// The build queues EVERYTHING from upper bounds and makes ONE host round trip at the end (round 6).  Before, every level cost ~10 round trips (three
// scans, two hub lists with a host sort each, the set offsets, pageable table uploads): ~45 per meta-batch, each waiting for its few microseconds of
// kernels to be scheduled beside a running meta-step -- 6 ms of a prefetched batch build at the arxiv shape, three times the receptive-field step it feeds.
// Bounds: level L = the centres (exact); level L - 1 holds at most as many rows as the centres have in-edges (gm_batch::n_e1, which is also the
// exact edge count into level L); deeper levels at most every row / every edge of the batch.  Level arrays are carved from ONE allocation of the
// bounds' size (arxiv query batch, two layers: ~45 MB), grids are sized by the bounds and read the real counts from device memory.
// n batches (the support and the query batch of a meta-batch) share the two synchronisations: the second build's ~110 launches are queued while
// the first one's run.
static int cone_build_many(const gm_batch* const* bs, int n, int L, hipStream_t s, gm_cone** cs) {
    std::vector<std::unique_ptr<ConeBuild>> builds;
    for (int k = 0; k < n; ++k) {
        builds.emplace_back(new ConeBuild(bs[k], L, s, cs[k]));
        GM_TRY(builds.back()->launch());
    }
    if (hipStreamSynchronize(s) != hipSuccess) { gm_set_error("cone: stream sync failed"); return GM_EHIP; }
    for (auto& cb : builds) GM_TRY(cb->finish());
    // the tables of the lower levels went up after the round trip: complete before a consumer on ANOTHER stream may use them (a few small copies, no kernels)
    if (hipStreamSynchronize(s) != hipSuccess) { gm_set_error("cone: stream sync failed"); return GM_EHIP; }
    return GM_OK;
}

int gm_batch_cone(const gm_batch* b, int L, hipStream_t s, const gm_cone** out) {
    *out = nullptr;
    GM_REQUIRE(b && L >= 1 && L <= GM_MAX_GCN, GM_EINVAL, "cone: n_gcn=%d outside [1,%d]", L, GM_MAX_GCN);
    if (b->cone[L]) { *out = b->cone[L]; return GM_OK; }
    gm_phase_timer tm("cone");
    gm_cone* c = new gm_cone();
    const int rc = cone_build_many(&b, 1, L, s, &c);
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

extern "C" int gm_batch_prepare_cone_pair(const gm_batch_t* a, const gm_batch_t* b, int32_t n_gcn, void* stream) {
    GM_REQUIRE(a && b && a != b, GM_EINVAL, "prepare_cone_pair: two distinct batches expected");
    GM_REQUIRE(n_gcn >= 1 && n_gcn <= GM_MAX_GCN, GM_EINVAL, "cone: n_gcn=%d outside [1,%d]", n_gcn, GM_MAX_GCN);
    const gm_batch* todo[2]; gm_cone* cs[2]; int n = 0;
    for (const gm_batch* x : {a, b}) if (!x->cone[n_gcn]) { todo[n] = x; cs[n] = new gm_cone(); ++n; }
    if (n == 0) return GM_OK;
    gm_phase_timer tm("cone-pair");
    const int rc = cone_build_many(todo, n, n_gcn, (hipStream_t)stream, cs);
    for (int k = 0; k < n; ++k) {
        if (rc != GM_OK) gm_cone_free(cs[k], (hipStream_t)stream); else todo[k]->cone[n_gcn] = cs[k];
    }
    return rc;
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
