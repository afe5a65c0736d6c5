// Stream aggregate: the batched segmented sum of agg.hip (DGL's update_all(copy_src, sum), learner.py:38-39,44-45, with GraphConv's
// normalisations, learner.py:29-32,49) as a REGISTER-LIGHT gather engine on LDS-DMA.
//
// Why a second kernel.  The window kernel (k_agg_win) hides memory latency with occupancy: ~36 VGPRs, 32 waves per CU, four 1-KiB row loads
// in flight per wave.  The persistent split GEMM of the other stream owns 16 waves x 120 VGPRs = 480 of the 512 registers of every SIMD lane:
// while it runs, no aggregate wave fits beside it, the two queues of a meta-step time-slice the chip and the HBM-bound aggregate never
// overlaps the matrix-bound update.  Here a wave keeps its gathers in flight in LDS instead of in registers:
//   * every gathered source row is ONE `global_load_lds_dwordx4` (1 KiB per instruction at width 256) into a wave-private ring of R
//     slots; the wave's in-order VM counter is the ring's only synchronisation (`s_waitcnt vmcnt(R - 1)` before slot i is read,
//     R - 1 younger gathers stay in flight): no destination registers, no barrier, R KiB in flight per wave;
//   * edge descriptors {source row, weight} come through the SCALAR cache (s_load_dwordx4 from an interleaved per-edge table laid out in
//     row order, one 4-edge block ahead), row bounds likewise (16 rows at a time, parked in the lanes of one VGPR), so the vector memory
//     queue carries nothing but gathers and the output stores -- nothing the compiler would wait `vmcnt(0)` for;
//   * a wave owns a contiguous run of rows (cost-balanced segments built with the batch, XCD-contiguous like the window kernel's blocks),
//     accumulates a row in edge order with the same fma chain as k_agg_win (bitwise the same rows) and writes it once.
// ~24 VGPRs and R KiB of LDS per wave: one 4-wave workgroup (48 KiB) fits next to a persistent GEMM workgroup (101 KiB, 480 VGPRs), several fit
// an otherwise empty CU.  Hub rows (in-degree above the batch's hub threshold) are left out of the stream tables (their row bound carries a
// flag) and written by whole workgroups as before.
#include <algorithm>
#include "gm_internal.h"

typedef int as_i4 __attribute__((ext_vector_type(4)));

#define AS_WAVES 4                      // waves per workgroup: one per SIMD

// ---- scalar-cache loads (SMEM returns out of order and the compiler does not count asm loads: every use is behind an explicit lgkmcnt(0))
__device__ __forceinline__ int as_sload(const void* p, int byte_off) {      // one dword, waited for
    int r;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p), "s"(byte_off) : "memory");
    return r;
}
// 8 dwords at p (4-byte aligned), issued only
__device__ __forceinline__ void as_sload8(const void* p, as_i4& a, as_i4& b) {
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
__device__ __forceinline__ void as_swait8(as_i4& a, as_i4& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) :: "memory");
}
// 16 dwords at p (4-byte aligned), issued only
__device__ __forceinline__ void as_sload16(const void* p, as_i4& a, as_i4& b, as_i4& c, as_i4& d) {
    asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\ts_load_dwordx4 %3, %4, 0x30"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p) : "memory");
}
__device__ __forceinline__ void as_swait16(as_i4& a, as_i4& b, as_i4& c, as_i4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) :: "memory");
}

// v_writelane_b32 with a run-time lane (no clang builtin; on gfx950 the value and the lane select cannot both be SGPRs, the select may be M0):
// lane `lane` of v <- val, both wave-uniform and set by scalar instructions; ignores EXEC.  M0 is the compiler's: saved and restored.
__device__ __forceinline__ void as_writelane(int& v, int val, int lane) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %1, %2, m0\n\ts_mov_b32 m0, %0" : "=&s"(keep), "+v"(v) : "s"(val), "s"(lane));
}

struct AggS {
    const int32_t* indptr_s;            // [rows + 1 (+ pad)] row bounds in the stream edge table; bit 31 of entry r + 1: row r is a hub row (not written here)
    const int2* ed;                     // stream edge table, row order: {source row of x, weight bits}
    const float* x; unsigned row_bytes; // gathered matrix, bytes per row (ldx * 4; rows * row_bytes < 4 GiB)
    float* out; int width; int nt;      // [rows, width]
    const int32_t* rowlist;             // optional: output row of stream row i (list launches)
    const int2* seg; int n_seg;         // per wave: rows [x, y) of the stream tables
};

// One gather: the edge's weight is parked in lane wlane of wring, then LDS[slot .. slot + LPR * 16) <- x[voff .. ), 16 bytes per lane of the
// lower LPR lanes (M0 = the slot's LDS byte address; EXEC is all ones around this statement: the kernel's control flow is wave-uniform).
template <int LPR>
__device__ __forceinline__ void as_issue(int& wring, int w, int wlane, unsigned lds_slot, const float* xbase, unsigned voff) {
    unsigned keep;
    if constexpr (LPR == 64)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %1, %2, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "+v"(wring) : "s"(w), "s"(wlane), "v"(voff), "s"(xbase), "s"(lds_slot) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %1, %2, m0\n\ts_mov_b32 m0, %6\n\t"
                     "s_mov_b32 exec_lo, %7\n\ts_mov_b32 exec_hi, 0\n\ts_nop 1\n\t"
                     "global_load_lds_dwordx4 %4, %5\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "+v"(wring) : "s"(w), "s"(wlane), "v"(voff), "s"(xbase), "s"(lds_slot), "s"(LPR == 32 ? 0xffffffffu : 0xffffu) : "memory");
}

template <int LPR, int R>
__global__ __launch_bounds__(AS_WAVES * 64) void k_agg_stream(AggS a) {
    constexpr int SLOT = LPR * 16;                                  // bytes per ring slot = one row
    __shared__ __attribute__((aligned(16))) char ring_all[AS_WAVES * R * SLOT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-contiguous segment order: hardware block b runs on XCD b % 8
    const int nwg = gridDim.x, b = blockIdx.x;
    const int sidx = ((b % GM_NXCD) * (nwg / GM_NXCD) + b / GM_NXCD) * AS_WAVES + wave;
    if (sidx >= a.n_seg) return;
    const int r_begin = as_sload(a.seg, sidx * 8), r_end = as_sload(a.seg, sidx * 8 + 4);
    if (r_begin >= r_end) return;
    const bool act = lane < LPR;                                    // narrow rows: the upper lanes only carry parked scalars (no early return: the compiler may
                                                                    // drop the contents of lanes it believes dead)
    const int e_lo = as_sload(a.indptr_s, r_begin * 4) & 0x7fffffff, e_hi = as_sload(a.indptr_s, r_end * 4) & 0x7fffffff;
    char* ring = ring_all + wave * (R * SLOT);
    const unsigned ring_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring);
    const unsigned lane16 = act ? (unsigned)lane * 16u : 0u;

    // ---- row walk of the consuming side: rc = current row, p1 = its end (flagged), bounds of 16 rows at a time parked in lanes (row & 31) of vp1
    int vp1 = 0;
    auto win_load = [&](int r16) {                                  // bounds of rows [r16, r16 + 16) -> lanes (r16 & 31) ... of vp1 (a scalar-cache round
        as_i4 q0, q1, q2, q3;                                       // trip every 16 rows; not prefetched: 16 more live SGPRs cost more than they hide)
        as_sload16(a.indptr_s + r16 + 1, q0, q1, q2, q3);
        as_swait16(q0, q1, q2, q3);
        const int l0 = r16 & 31;
#pragma unroll
        for (int k = 0; k < 4; ++k) { as_writelane(vp1, q0[k], l0 + k); as_writelane(vp1, q1[k], l0 + 4 + k); as_writelane(vp1, q2[k], l0 + 8 + k); as_writelane(vp1, q3[k], l0 + 12 + k); }
    };
    int rc = r_begin;
    win_load(rc & ~15);
    int p1 = __builtin_amdgcn_readlane(vp1, rc & 31);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto row_done = [&]() {                                         // write row rc (unless a hub row), move to the next
        if (p1 >= 0 && act) {
            const int64_t orow = a.rowlist ? (int64_t)as_sload(a.rowlist, rc * 4) : (int64_t)rc;
            float4* dst = reinterpret_cast<float4*>(a.out + orow * a.width + lane * 4);
            if (a.nt) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v vv = {acc.x, acc.y, acc.z, acc.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(dst));
            } else *dst = acc;
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ++rc;
        if ((rc & 15) == 0 && rc < r_end) win_load(rc);
        p1 = __builtin_amdgcn_readlane(vp1, rc & 31);
    };

    // ---- the gather pipeline
    int wring = 0;                                                  // weight of edge e parked in lane e & 63 until it is consumed (R <= 64)
    int slot = 0;                                                   // ring slot of the edge being issued == of the edge being consumed (e and e - R)
    auto consume = [&](int ec, bool steady) {
        while (ec == (p1 & 0x7fffffff)) row_done();                 // rows that end before this edge (empty rows included)
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float4 v = *reinterpret_cast<const float4*>(ring + slot * SLOT + lane16);
        const float w = __int_as_float(__builtin_amdgcn_readlane(wring, ec & 63));
        acc.x = __fmaf_rn(v.x, w, acc.x); acc.y = __fmaf_rn(v.y, w, acc.y); acc.z = __fmaf_rn(v.z, w, acc.z); acc.w = __fmaf_rn(v.w, w, acc.w);
    };
    auto step = [&](int e, int u, int w) {
        if (e - R >= e_lo) consume(e - R, true);                    // frees the slot edge e lands in
        as_issue<LPR>(wring, w, e & 63, ring_lds + (unsigned)(slot * SLOT), a.x, (unsigned)u * a.row_bytes + lane16);
        slot = slot + 1 == R ? 0 : slot + 1;
    };
    if (e_hi > e_lo) {
        const int kb0 = e_lo >> 2, kb1 = (e_hi - 1) >> 2;           // descriptor blocks of 4 edges (32 bytes), one block ahead
        as_i4 d0, d1, n0, n1;
        as_sload8(a.ed + (int64_t)kb0 * 4, d0, d1);
        as_swait8(d0, d1);
        for (int kb = kb0; kb <= kb1; ++kb) {
            if (kb < kb1) as_sload8(a.ed + (int64_t)(kb + 1) * 4, n0, n1);
            const int e0 = kb * 4;
#define AS_STEP(J, D, K) if (e0 + J >= e_lo && e0 + J < e_hi) step(e0 + J, D[K], D[K + 1]);
            AS_STEP(0, d0, 0) AS_STEP(1, d0, 2) AS_STEP(2, d1, 0) AS_STEP(3, d1, 2)
#undef AS_STEP
            if (kb < kb1) { as_swait8(n0, n1); d0 = n0; d1 = n1; }
        }
        // drain: the last min(R, edges) gathers
        int ec = e_hi - R > e_lo ? e_hi - R : e_lo;
        slot = (ec - e_lo) % R;
        for (; ec < e_hi; ++ec) { consume(ec, false); slot = slot + 1 == R ? 0 : slot + 1; }
    }
    while (rc < r_end) row_done();                                  // the last row with edges and the empty rows behind it
}

// ---------------------------------------------------------------------------------------------------------------- tables (built with the batch)
// Row bounds of the stream edge table: the batch's CSR bounds minus the edges of the hub rows before each row (hub rows keep no edges
// here; bit 31 of a hub row's END bound flags it).  hubs: ascending hub rows, cum[k] = edges of the hub rows before hub k (cum[n] = all).
__global__ void k_stream_bounds(const int32_t* indptr, int64_t rows, const int32_t* hubs, const int32_t* cum, int n_hubs, int32_t* indptr_s, int pad) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;         // entry r (r = rows: the end of the last row)
    if (r > rows + pad) return;
    if (r > rows) { indptr_s[r] = 0; return; }
    int lo = 0, hi = n_hubs;                                                  // hubs before row r
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hubs[mid] < r) lo = mid + 1; else hi = mid; }
    const bool prev_is_hub = r > 0 && lo > 0 && hubs[lo - 1] == r - 1;
    indptr_s[r] = (indptr[r] - cum[lo]) | (prev_is_hub ? (int)0x80000000 : 0);
}
// {source, weight} of every non-hub edge at its stream position; one thread per row
__global__ void k_stream_edges(const int32_t* indptr, const int32_t* indptr_s, int64_t rows, const int32_t* src, const float* wgt, int2* ed) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    if (indptr_s[r + 1] < 0) return;                                          // hub row
    const int p0 = indptr[r], n = indptr[r + 1] - p0, o = indptr_s[r] & 0x7fffffff;
    for (int j = 0; j < n; ++j) ed[o + j] = make_int2(src[p0 + j], __float_as_int(wgt ? wgt[p0 + j] : 1.f));
}
// Wave segments: segment k starts at the first row whose cost prefix (stream edges + rows before it) reaches k / n_seg of the total
__global__ void k_stream_segs(const int32_t* indptr_s, int64_t rows, int n_seg, int2* seg) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const int64_t total = (int64_t)(indptr_s[rows] & 0x7fffffff) + rows;
    auto first_row = [&](int kk) -> int {
        if (kk >= n_seg) return (int)rows;
        const int64_t target = total * kk / n_seg;
        int64_t lo = 0, hi = rows;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)(indptr_s[mid] & 0x7fffffff) + mid < target) lo = mid + 1; else hi = mid; }
        return (int)lo;
    };
    seg[k] = make_int2(first_row(k), first_row(k + 1));
}

int gm_stream_wgs() {                                               // workgroups of a stream launch: a multiple of the XCD count
    const int per_cu = gm_knob().agg_stream_wgs > 0 ? gm_knob().agg_stream_wgs : 3;
    return std::max(GM_NXCD, gm_num_cus() * per_cu / GM_NXCD * GM_NXCD);
}

// Stream tables of one orientation (o = 0: by destination; 1: by source).  hubs_host / deg_host: the orientation's ascending hub rows and
// their degrees (host copies from the finalisation's round trip).  Needs the batch's per-edge tables (d_enorm, d_efeat).
int gm_stream_tables(gm_batch* b, int o, const int32_t* hubs_host, const int32_t* deg_host, int n_hubs, hipStream_t s, gm_stager* sg) {
    if (b->rows <= 0 || b->edges <= 0 || !b->d_enorm[o]) return GM_OK;
    const int32_t* indptr = o ? b->d_indptr_t : b->d_indptr;
    std::vector<int32_t> cum(n_hubs + 1, 0);
    for (int k = 0; k < n_hubs; ++k) cum[k + 1] = cum[k] + deg_host[k];
    int32_t *d_cum = nullptr, *d_hubs = nullptr;
    GM_TRY(gm_balloc(b, &d_cum, cum.size(), s)); GM_TRY(sg->upload(d_cum, cum));
    GM_TRY(gm_balloc(b, &d_hubs, (size_t)std::max(n_hubs, 1), s));
    if (n_hubs > 0) GM_TRY(sg->upload((void*)d_hubs, (const void*)hubs_host, sizeof(int32_t) * (size_t)n_hubs));
    const int pad = 48;                                             // the 16-row bound windows read up to 31 entries past the end
    GM_TRY(gm_balloc(b, &b->d_sindptr[o], (size_t)b->rows + 1 + pad, s));
    hipLaunchKernelGGL(k_stream_bounds, dim3((unsigned)((b->rows + 1 + pad + 255) / 256)), dim3(256), 0, s, indptr, (int64_t)b->rows, d_hubs, d_cum, n_hubs, b->d_sindptr[o], pad);
    const size_t n_ed = (size_t)(b->edges - cum[n_hubs]) + 16;     // (+ the tail of the last 8-edge descriptor block)
    GM_TRY(gm_balloc(b, &b->d_sed[o], n_ed, s));
    hipLaunchKernelGGL(k_stream_edges, dim3((unsigned)((b->rows + 255) / 256)), dim3(256), 0, s, indptr, b->d_sindptr[o], (int64_t)b->rows,
                       o ? b->d_indices_t : b->d_indices, b->d_enorm[o], b->d_sed[o]);
    if (o == 0 && b->d_efeat) {                                     // layer 1 gathers rows of the store's feature table
        GM_TRY(gm_balloc(b, &b->d_sed_feat, n_ed, s));
        hipLaunchKernelGGL(k_stream_edges, dim3((unsigned)((b->rows + 255) / 256)), dim3(256), 0, s, indptr, b->d_sindptr[o], (int64_t)b->rows, b->d_efeat, b->d_enorm[o], b->d_sed_feat);
    }
    b->stream_nseg = gm_stream_wgs() * AS_WAVES;
    GM_TRY(gm_balloc(b, &b->d_sseg[o], (size_t)b->stream_nseg, s));
    hipLaunchKernelGGL(k_stream_segs, dim3((b->stream_nseg + 255) / 256), dim3(256), 0, s, b->d_sindptr[o], (int64_t)b->rows, b->stream_nseg, b->d_sseg[o]);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

template <int LPR, int R>
static void launch_stream(const AggS& a, hipStream_t s) {
    hipLaunchKernelGGL((k_agg_stream<LPR, R>), dim3(a.n_seg / AS_WAVES), dim3(AS_WAVES * 64), 0, s, a);
}

// The stream launch of a full aggregate over the batch the tables belong to; false: not eligible (the caller takes the window kernel)
bool gm_stream_ok(const gm_agg_args& g) {
    return g.stream_indptr && g.stream_ed && g.stream_seg && !g.s_out && !g.bias && !g.mask_h && !g.mask_b && !g.relu && !g.relu_bits && !g.skip_on && !g.rowlist &&
           (g.width == 64 || g.width == 128 || g.width == 256) && g.ldx % 4 == 0 && (((uintptr_t)g.x | (uintptr_t)g.out) & 15) == 0 &&
           (uint64_t)g.stream_xrows * (uint64_t)g.ldx * 4u < ((uint64_t)1 << 32);
}
int gm_launch_stream(const gm_agg_args& g, int nt, hipStream_t s) {
    AggS a{g.stream_indptr, g.stream_ed, g.x, (unsigned)(g.ldx * 4), g.out, g.width, nt, nullptr, g.stream_seg, g.stream_nseg};
    const int depth = gm_knob().agg_stream_depth;
    if (g.width == 256) { if (depth == 8) launch_stream<64, 8>(a, s); else if (depth == 16) launch_stream<64, 16>(a, s); else launch_stream<64, 12>(a, s); }
    else if (g.width == 128) { if (depth == 8) launch_stream<32, 16>(a, s); else if (depth == 16) launch_stream<32, 32>(a, s); else launch_stream<32, 24>(a, s); }
    else { if (depth == 8) launch_stream<16, 32>(a, s); else launch_stream<16, 48>(a, s); }
    GM_HIP(hipGetLastError());
    return GM_OK;
}
