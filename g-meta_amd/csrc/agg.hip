// Batched subgraph message passing on gfx950: segmented (CSR) sum over in-edges, i.e. DGL's
// graph.update_all(fn.copy_src, fn.sum) (learner.py:38-39,44-45) with GraphConv's degree
// normalisations fused in (learner.py:29-32,49).  HBM-bound: every row of X is read once from HBM
// (re-reads by other destination rows of the same subgraph hit the XCD's L2, because a subgraph's
// rows are contiguous and consecutive row blocks are mapped to the same XCD) and every row of the
// output is written once.  A lane group of LPR lanes owns one destination row; each lane holds VEC
// consecutive floats, so a 256-wide row is one 1-KiB coalesced access per wave.
#include <algorithm>
#include <stdlib.h>
#include <mutex>
#include "gm_internal.h"

#define AGG_BLOCK 256

struct AggK {
    const int32_t* indptr; const int32_t* indices; const float* x; const int32_t* x_row; int64_t ldx;
    const float* s_in; const float* s_out; const float* mask_h; const float* bias; int64_t bias_stride;
    const int32_t* set_row_off; int n_sets; int relu; float* out; int64_t rows; int width; int nblocks;
    const uint8_t* mask_b; uint8_t* relu_bits;
    const int32_t* heavy; int n_heavy, heavy_deg;
    const int32_t* sched; int sched_len;      // optional block schedule: entry >= 0 window block, <= -2 hub row, -1 nothing
    int nt;                    // 1: non-temporal output stores (Z is not re-read by this kernel; keep L2 for the X gathers)
    int win;                   // rows per wave window (64 for big batches; smaller when the batch would underfill the chip)
    // hub rows split over several blocks (gm_agg_schedule): hub = [part_off: n_heavy+1][part_hub: parts][arrival counters: n_heavy],
    // part g of the schedule covers edges [hub_part * p, hub_part * (p+1)) of its row; the last block to arrive sums the partial rows
    // (hub_scratch, [parts][hub_ld]) in part order -- deterministic -- and runs the epilogue.  hub == NULL: one block per hub row.
    const int32_t* hub; float* hub_scratch; int hub_part, hub_ld;
    const float* e_w; const int32_t* x_idx;     // per-edge source scale / source row of x (see gm_agg_args)
    int skip_lo, skip_hi;                       // rows with skip_lo <= degree <= skip_hi are left to the fused aggregate+GEMM kernel (empty range: none)
    const int32_t* rowlist; int64_t n_list;     // optional: the windows walk this (ascending) row list instead of rows 0 .. rows
};

// edge e -> (row of x to read, its scale): from the per-edge tables when the launch has them, else through indices / s_in / x_row
__device__ __forceinline__ void agg_edge(const AggK& a, const int e, int& u, float& w) {
    if (a.e_w && (a.x_idx || !a.x_row)) {
        w = a.e_w[e]; u = a.x_idx ? a.x_idx[e] : a.indices[e];
    } else {
        u = a.indices[e]; w = a.s_in ? a.s_in[u] : 1.f;
        if (a.x_idx) u = a.x_idx[e]; else if (a.x_row) u = a.x_row[u];
    }
}
template <int VEC> struct VecT;
template <> struct VecT<4> { using T = float4; };
template <> struct VecT<1> { using T = float; };

__device__ __forceinline__ void vfma(float4& a, const float4& v, float s) { a.x += v.x * s; a.y += v.y * s; a.z += v.z * s; a.w += v.w * s; }
__device__ __forceinline__ void vfma(float& a, const float& v, float s) { a += v * s; }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }
__device__ __forceinline__ float vget(const float4& a, int k) { return k == 0 ? a.x : k == 1 ? a.y : k == 2 ? a.z : a.w; }
__device__ __forceinline__ float vget(const float& a, int) { return a; }
__device__ __forceinline__ void vset(float4& a, int k, float v) { if (k == 0) a.x = v; else if (k == 1) a.y = v; else if (k == 2) a.z = v; else a.w = v; }
__device__ __forceinline__ void vset(float& a, int, float v) { a = v; }

template <int VEC, int LPR>
__global__ __launch_bounds__(AGG_BLOCK) void k_agg(AggK a) {
    using V = typename VecT<VEC>::T;
    constexpr int RPW = GM_WAVE / LPR;                 // rows per wave
    constexpr int RPB = RPW * (AGG_BLOCK / GM_WAVE);   // rows per block
    // XCD-aware mapping: hardware block b runs on XCD b % 8; give each XCD a contiguous range of row
    // blocks so that one subgraph's rows (and its gathers) stay in one L2.
    const int nb = a.nblocks, b = blockIdx.x;
    const int q = nb / GM_NXCD, r = nb % GM_NXCD, xcd = b % GM_NXCD, idx = b / GM_NXCD;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    const int64_t row = (int64_t)lb * RPB + wave * RPW + sub;
    if (row >= a.rows) return;
    const int e0 = a.indptr[row], e1 = a.indptr[row + 1];
    const float so = a.s_out ? a.s_out[row] : 1.0f;
    int set = 0;
    if (a.bias && a.bias_stride) {                      // set of this row (few sets: short binary search)
        int lo = 0, hi = a.n_sets;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.set_row_off[mid] <= row) lo = mid; else hi = mid; }
        set = lo;
    }
    for (int c0 = l * VEC; c0 < a.width; c0 += LPR * VEC) {
        V acc0, acc1; vzero(acc0); vzero(acc1);
        int e = e0;
        for (; e + 4 <= e1; e += 4) {                   // 4 independent gathers in flight per lane
            int u0, u1, u2, u3; float w0, w1, w2, w3;
            agg_edge(a, e, u0, w0); agg_edge(a, e + 1, u1, w1); agg_edge(a, e + 2, u2, w2); agg_edge(a, e + 3, u3, w3);
            const V v0 = *reinterpret_cast<const V*>(a.x + (int64_t)u0 * a.ldx + c0);
            const V v1 = *reinterpret_cast<const V*>(a.x + (int64_t)u1 * a.ldx + c0);
            const V v2 = *reinterpret_cast<const V*>(a.x + (int64_t)u2 * a.ldx + c0);
            const V v3 = *reinterpret_cast<const V*>(a.x + (int64_t)u3 * a.ldx + c0);
            vfma(acc0, v0, w0); vfma(acc1, v1, w1); vfma(acc0, v2, w2); vfma(acc1, v3, w3);
        }
        for (; e < e1; ++e) {
            int u; float w;
            agg_edge(a, e, u, w);
            vfma(acc0, *reinterpret_cast<const V*>(a.x + (int64_t)u * a.ldx + c0), w);
        }
        V res;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float v = (vget(acc0, k) + vget(acc1, k)) * so;
            if (a.bias) v += a.bias[(int64_t)set * a.bias_stride + c0 + k];
            if (a.relu) v = v < 0.f ? 0.f : v;      // NaN propagates like torch relu (meta.py:163 guard)
            if (a.mask_b) v = ((a.mask_b[(row * a.width + c0) >> 2] >> k) & 1u) ? v : 0.f;      // VEC == 4 only (launcher)
            else if (a.mask_h) v = a.mask_h[row * a.width + c0 + k] > 0.f ? v : 0.f;
            vset(res, k, v);
        }
        if (a.relu_bits) {                          // VEC == 4 only (launcher)
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < VEC; ++k) bits |= (vget(res, k) > 0.f ? 1u : 0u) << k;
            a.relu_bits[(row * a.width + c0) >> 2] = (uint8_t)bits;
        }
        *reinterpret_cast<V*>(a.out + row * a.width + c0) = res;
    }
}

// One 1024-thread workgroup per heavy row (in-degree > gm_heavy_deg(): hub nodes inside their own neighbourhood, up to
// ~1000 edges): enough loads in flight on one CU (16 groups x 8 x 1 KiB) to stream the row instead of crawling through it.
// The 1024/LPR lane groups take interleaved LPR-edge chunks (coalesced index loads, 8 row loads in flight each), the
// partial rows are summed through LDS in a fixed order (deterministic), then the usual epilogue.
#define AGG_HEAVY_BLOCK 1024
template <int LPR, int NCH, int NTHREADS>
__device__ __forceinline__ void agg_heavy_row(const AggK& a, const int g, float* part) {
    constexpr int NG = NTHREADS / LPR;
    constexpr int PS = 16;                                 // edges per group visit (<= LPR): two batches of 8 row loads
    const int tid = threadIdx.x, gi = tid / LPR, l = tid % LPR, lane = tid & 63;
    const int gbase = (lane / LPR) * LPR;                 // first lane of this group inside its wave
    int h = g, p = 0, P = 1;
    if (a.hub) { h = a.hub[a.n_heavy + 1 + g]; p = g - a.hub[h]; P = a.hub[h + 1] - a.hub[h]; }
    const int row = a.heavy[h];
    int e0 = a.indptr[row], e1 = a.indptr[row + 1];
    if (P > 1) { e0 += p * a.hub_part; if (p < P - 1) e1 = e0 + a.hub_part; }      // the last part takes the remainder (up to 1.5 parts)
    const float* xl = a.x + l * 4;
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int eb = e0 + gi * PS; eb < e1; eb += NG * PS) {
        int mu = 0; float mw = 0.f;
        if (l < PS && eb + l < e1) agg_edge(a, eb + l, mu, mw);
        const int cnt = min(PS, e1 - eb);
        for (int j = 0; j < cnt; j += 8) {
            float4 v[8][NCH]; float ww[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int sl = gbase + j + i;
                const int uu = __shfl(mu, sl, 64);
                ww[i] = (j + i < cnt) ? __shfl(mw, sl, 64) : 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[i][c] = *reinterpret_cast<const float4*>(xl + (int64_t)uu * a.ldx + c * LPR * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < NCH; ++c) vfma(acc[c], v[i][c], ww[i]);
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(&part[(gi * NCH + c) * LPR * 4 + l * 4]) = acc[c];
    __syncthreads();
    float4 s4[NCH];
    if (gi == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            s4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < NG; ++k) { const float4 t = *reinterpret_cast<const float4*>(&part[(k * NCH + c) * LPR * 4 + l * 4]); s4[c].x += t.x; s4[c].y += t.y; s4[c].z += t.z; s4[c].w += t.w; }
        }
    }
    if (P > 1) {
        // Partial row -> scratch with write-through (sc1) 16-byte stores, drained by every wave; one relaxed agent-scope ticket;
        // only the last block to arrive goes on: one lane's agent-scope acquire, then plain loads of the P partial rows, summed
        // in part order.  Correct wherever the parts run (the schedule keeps them on one XCD only because that is faster).
        typedef float f4v __attribute__((ext_vector_type(4)));
        if (gi == 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float* dst = a.hub_scratch + (int64_t)g * a.hub_ld + l * 4 + c * LPR * 4;
                const f4v val = {s4[c].x, s4[c].y, s4[c].z, s4[c].w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // also: every read of `part` above is done
        int* ctr = const_cast<int*>(a.hub) + a.n_heavy + 1 + a.hub[a.n_heavy] + h;
        int* flag = reinterpret_cast<int*>(part);
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == P - 1) {
                __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *flag = (old == P - 1);
        }
        __syncthreads();
        if (!*flag || gi != 0) return;
        const float* sc = a.hub_scratch + (int64_t)a.hub[h] * a.hub_ld + l * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            s4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < P; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(sc + (int64_t)k * a.hub_ld + c * LPR * 4);
                s4[c].x += t.x; s4[c].y += t.y; s4[c].z += t.z; s4[c].w += t.w;
            }
        }
    }
    if (gi != 0) return;
    const float so = a.s_out ? a.s_out[row] : 1.f;
    const float* bp = nullptr;
    if (a.bias) {
        int set = 0;
        if (a.bias_stride) {
            int lo = 0, hi = a.n_sets;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.set_row_off[mid] <= row) lo = mid; else hi = mid; }
            set = lo;
        }
        bp = a.bias + (int64_t)set * a.bias_stride + l * 4;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float4 v = make_float4(s4[c].x * so, s4[c].y * so, s4[c].z * so, s4[c].w * so);
        if (bp) { const float4 bb = *reinterpret_cast<const float4*>(bp + c * LPR * 4); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
        if (a.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
        if (a.mask_b) {
            const unsigned m = a.mask_b[((int64_t)row * a.width + l * 4 + c * LPR * 4) >> 2];
            v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
        } else if (a.mask_h) {
            const float4 m = *reinterpret_cast<const float4*>(a.mask_h + (int64_t)row * a.width + l * 4 + c * LPR * 4);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (a.relu_bits) a.relu_bits[((int64_t)row * a.width + l * 4 + c * LPR * 4) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
        *reinterpret_cast<float4*>(a.out + (int64_t)row * a.width + l * 4 + c * LPR * 4) = v;
    }
}
template <int LPR, int NCH>
__global__ __launch_bounds__(AGG_HEAVY_BLOCK) void k_agg_heavy(AggK a) {
    __shared__ __attribute__((aligned(16))) float part[(AGG_HEAVY_BLOCK / LPR) * LPR * 4 * NCH];
    agg_heavy_row<LPR, NCH, AGG_HEAVY_BLOCK>(a, blockIdx.x, part);
}

// Wave-cooperative kernel (the production path for widths 64/128/256/512).  Induced subgraphs are very sparse
// (arxiv config: median in-degree 1, p90 2), so a row-per-wave kernel spends its life in the dependent chain
// indptr -> indices -> scale -> row.  Here a wave owns a WINDOW of 64 consecutive rows: lane i fetches row i's CSR
// bounds, its first two sources and their scales with coalesced loads (one dependent chain per 64 rows instead of
// per row); then the row loads are issued from register-held addresses (broadcast by ds_bpermute), UNR rows at a
// time, so up to 2*UNR independent 16-B loads per lane are in flight.  Edges beyond the second (p99 ~ 19) take a
// conventional loop.  LPR = width/4 lanes own a row (width 256: the whole wave, 1 KiB per access).
template <int LPR, int NCH, int UNR, int MB>
__global__ __launch_bounds__(AGG_BLOCK) void k_agg_win(AggK a) {
    constexpr int G = GM_WAVE / LPR;           // rows processed side by side
    const int nb = a.nblocks, b = blockIdx.x;
    const int q = nb / GM_NXCD, r = nb % GM_NXCD, xcd = b % GM_NXCD, idx = b / GM_NXCD;
    int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (a.sched) {
        // One launch for everything: the schedule interleaves a 256-thread block per hub row (in-degree > heavy_deg) with the
        // window blocks of the hub's own subgraph, on the XCD that owns them -- the hub gathers ~every row of its subgraph,
        // which the neighbouring window blocks are pulling through that XCD's L2 at that moment.  (As a separate launch the
        // hub rows re-read their subgraphs from HBM: 0.25 % of the rows caused ~40 % of the fetch traffic.)
        const int e = a.sched[xcd * a.sched_len + idx];
        if (e == -1) return;
        if (e < -1) {
            __shared__ __attribute__((aligned(16))) float part[AGG_BLOCK * 4 * NCH];
            agg_heavy_row<LPR, NCH, AGG_BLOCK>(a, -e - 2, part);
            return;
        }
        lb = e;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPR, l = lane % LPR;
    const int64_t R0 = ((int64_t)lb * (AGG_BLOCK / GM_WAVE) + wave) * a.win;
    const int64_t nrows_w = a.rowlist ? a.n_list : a.rows;          // rows the windows walk: the list's, or all
    if (R0 >= nrows_w) return;
    // ---- per-lane row descriptor (coalesced)
    const bool mine = lane < a.win && R0 + lane < nrows_w;
    const int64_t myrow = a.rowlist ? (mine ? (int64_t)a.rowlist[R0 + lane] : 0) : R0 + lane;
    const int myrow32 = (int)myrow;
    int p0 = 0, dg = 0, u0 = 0, u1 = 0; float w0 = 0.f, w1 = 0.f, so = 1.f;
    if (mine) {
        p0 = a.indptr[myrow]; dg = a.indptr[myrow + 1] - p0;
        if (a.s_out) so = a.s_out[myrow];
        if (dg >= a.skip_lo && dg <= a.skip_hi) dg = -2;                   // not this launch's row
        if (dg >= 1) agg_edge(a, p0, u0, w0);
        if (dg >= 2) agg_edge(a, p0 + 1, u1, w1);
    }
    const int nwin = (int)min((int64_t)a.win, nrows_w - R0);
    const float* xl = a.x + l * 4;
    for (int t0 = 0; t0 < nwin; t0 += G * UNR) {
        float4 acc[UNR][NCH];
        int rdg[UNR], rp0[UNR], rrow[UNR]; float rso[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int rr = t0 + k * G + g;                      // row of this lane group inside the window
            const int src = rr < GM_WAVE ? rr : 0;
            rdg[k] = rr < nwin ? __shfl(dg, src, 64) : -1;
            // (row-list mode: the row id is broadcast HERE, where the wave is convergent -- behind the divergent `continue` below the source
            // lane of a partial last window may be masked off and ds_bpermute would hand back 0)
            rrow[k] = __shfl(myrow32, src, 64);
            if (a.n_heavy && rdg[k] > a.heavy_deg) rdg[k] = -1;            // written by its own workgroup (k_agg_heavy)
            rp0[k] = __shfl(p0, src, 64); rso[k] = __shfl(so, src, 64);
            const int a0 = __shfl(u0, src, 64), a1 = __shfl(u1, src, 64);
            const float f0 = __shfl(w0, src, 64), f1 = __shfl(w1, src, 64);
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[k][c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rdg[k] >= 1) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) vfma(acc[k][c], *reinterpret_cast<const float4*>(xl + (int64_t)a0 * a.ldx + c * LPR * 4), f0);
            }
            if (rdg[k] >= 2) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) vfma(acc[k][c], *reinterpret_cast<const float4*>(xl + (int64_t)a1 * a.ldx + c * LPR * 4), f1);
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (rdg[k] < 0) continue;
            const int rr_ = t0 + k * G + g;
            const int64_t row = a.rowlist ? (int64_t)rrow[k] : R0 + rr_;
            // rows with more than two in-edges (p99 ~ 19, hubs up to ~1000): the group's LPR lanes fetch the next LPR
            // sources + scales with one coalesced load each, then 8 row loads at a time are issued from registers.
            const int eend = rp0[k] + rdg[k];
            for (int eb = rp0[k] + 2; eb < eend; eb += LPR) {
                int mu = 0; float mw = 0.f;
                if (eb + l < eend) agg_edge(a, eb + l, mu, mw);
                const int cnt = min(LPR, eend - eb);
                for (int j = 0; j < cnt; j += MB) {
                    float4 v[MB][NCH]; float ww[MB];
#pragma unroll
                    for (int i = 0; i < MB; ++i) {
                        const int sl = g * LPR + ((j + i) & (LPR - 1));
                        const int uu = __shfl(mu, sl, 64);
                        ww[i] = (j + i < cnt) ? __shfl(mw, sl, 64) : 0.f;      // out-of-range slots re-read a valid row with weight 0
#pragma unroll
                        for (int c = 0; c < NCH; ++c) v[i][c] = *reinterpret_cast<const float4*>(xl + (int64_t)uu * a.ldx + c * LPR * 4);
                    }
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int c = 0; c < NCH; ++c) vfma(acc[k][c], v[i][c], ww[i]);
                }
            }
            const float* bp = nullptr;
            if (a.bias) {
                int set = 0;
                if (a.bias_stride) {
                    int lo = 0, hi = a.n_sets;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.set_row_off[mid] <= row) lo = mid; else hi = mid; }
                    set = lo;
                }
                bp = a.bias + (int64_t)set * a.bias_stride + l * 4;
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float s_ = rso[k];
                float4 v = make_float4(acc[k][c].x * s_, acc[k][c].y * s_, acc[k][c].z * s_, acc[k][c].w * s_);
                if (bp) { const float4 bb = *reinterpret_cast<const float4*>(bp + c * LPR * 4); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                if (a.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                if (a.mask_b) {
                    const unsigned m = a.mask_b[(row * a.width + l * 4 + c * LPR * 4) >> 2];
                    v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
                } else if (a.mask_h) {
                    const float4 m = *reinterpret_cast<const float4*>(a.mask_h + row * a.width + l * 4 + c * LPR * 4);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
                if (a.relu_bits) a.relu_bits[(row * a.width + l * 4 + c * LPR * 4) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
                float4* dst = reinterpret_cast<float4*>(a.out + row * a.width + l * 4 + c * LPR * 4);
                if (a.nt) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    f4v vv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(dst));
                } else *dst = v;
            }
        }
    }
}

// Rows per wave window: 64, halved until the launch has enough waves to fill the chip.  "Enough" depends on the work a row carries: sparse
// batches (the arxiv / FirstMM shapes, ~2 in-edges per row) want >= 32k waves and may go down to 2-row windows; dense ones (Tissue shape,
// ~30 in-edges per row) are better off with half as many, larger windows (4 rows at least) -- measured with the two-piece GEMMs in place:
// 65536 / 2 -> these rules: 4-task arxiv shard 4.31 -> 4.17 ms, task_num 32 24.43 -> 24.25, Tissue shape 3.65 -> 3.36.
int gm_agg_window(int64_t rows, int64_t edges) {
    const bool dense = edges > 8 * rows;
    const int min_waves = gm_knob().agg_min_waves > 0 ? gm_knob().agg_min_waves : (dense ? 16384 : 32768);
    const int min_win = gm_knob().agg_min_win > 0 ? gm_knob().agg_min_win : (dense ? 4 : 2);
    int win = 64;
    while (win > min_win && rows / win < min_waves) win >>= 1;
    return win;
}

int gm_agg_schedule_flat(gm_batch* b, int64_t rows, int win, const int32_t* pos, int n_heavy, const std::vector<int32_t>& tab, int32_t** d_sched, int32_t* len_out,
                         hipStream_t s, gm_stager* sg) {
    const bool split = !tab.empty();
    const int RPB = win * (AGG_BLOCK / GM_WAVE);
    const int nwb = std::max(1, (int)((rows + RPB - 1) / RPB));
    const int q = nwb / GM_NXCD, r = nwb % GM_NXCD;
    // Two passes over the (ascending) hub rows instead of eight growing lists (this ran on the host between the build's kernels: 110 us per
    // orientation of the 1.14 M-row batch): the hub blocks of every XCD's window range are counted first, which gives the common list length,
    // then the flat [8][len] array is written in place -- runs of window blocks with the hub entries spliced in behind their block.
    auto x_start = [&](int x) { return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; };
    auto n_entries = [&](int k) { return split ? tab[k + 1] - tab[k] : 1; };
    int extra[GM_NXCD] = {}; int first[GM_NXCD + 1];
    {
        int hk = 0;
        for (int x = 0; x < GM_NXCD; ++x) {
            const int end = x_start(x) + (x < r ? q + 1 : q);
            first[x] = hk;
            while (hk < n_heavy && pos[hk] / RPB < end) {
                GM_REQUIRE(hk == 0 || pos[hk] >= pos[hk - 1], GM_EINVAL, "aggregate schedule: hub-row list is not ascending");
                extra[x] += n_entries(hk); ++hk;
            }
        }
        first[GM_NXCD] = hk;
        GM_REQUIRE(hk == n_heavy, GM_EINVAL, "aggregate schedule: hub-row list is not ascending / out of range");
    }
    size_t len = 0;
    for (int x = 0; x < GM_NXCD; ++x) len = std::max(len, (size_t)((x < r ? q + 1 : q) + extra[x]));
    std::vector<int32_t> flat(GM_NXCD * len);
    for (int x = 0; x < GM_NXCD; ++x) {
        int32_t* o = flat.data() + x * len; int32_t* const o_end = o + len;
        int wb = x_start(x); const int end = wb + (x < r ? q + 1 : q);
        for (int hk = first[x]; hk < first[x + 1]; ++hk) {
            const int hb = pos[hk] / RPB;
            while (wb <= hb) *o++ = wb++;                                  // window blocks up to and including the hub row's
            if (split) for (int g = tab[hk]; g < tab[hk + 1]; ++g) *o++ = -g - 2;
            else *o++ = -hk - 2;
        }
        while (wb < end) *o++ = wb++;
        while (o < o_end) *o++ = -1;
    }
    GM_TRY(gm_balloc(b, d_sched, flat.size(), s));
    GM_TRY(sg->upload(*d_sched, flat));          // (through pinned staging: no host round trip)
    *len_out = (int32_t)len;
    return GM_OK;
}

int gm_agg_schedule(gm_batch* b, int64_t rows, int win, const int32_t* heavy_host, const int32_t* heavy_deg_host, int n_heavy, gm_agg_sched* out, hipStream_t s, gm_stager* sg) {
    *out = gm_agg_sched{};
    const int on = gm_knob().agg_sched, part_env = gm_knob().agg_hub_part;      // part_env 0: one block per hub row
    if (!on || n_heavy <= 0 || rows <= 0) return GM_OK;                  // no hub rows: the plain window launch
    // edges per hub part: a multiple of 16, at most 32 parts for the widest row
    int hub_part = 0;
    std::vector<int32_t> tab;                                           // [part_off: n_heavy+1][part_hub: parts][counters: n_heavy]
    if (part_env > 0 && heavy_deg_host) {
        int maxdeg = 0;
        for (int k = 0; k < n_heavy; ++k) maxdeg = std::max(maxdeg, heavy_deg_host[k]);
        hub_part = std::max((part_env + 15) / 16 * 16, ((maxdeg + 31) / 32 + 15) / 16 * 16);
        tab.assign(n_heavy + 1, 0);
        for (int k = 0; k < n_heavy; ++k) tab[k + 1] = tab[k] + std::max(1, (heavy_deg_host[k] + hub_part / 2) / hub_part);      // nearest: a row is split from 1.5 parts upwards
        const int parts = tab[n_heavy];
        if (parts == n_heavy) { hub_part = 0; tab.clear(); }             // nothing to split
        else {
            tab.resize(n_heavy + 1 + parts + n_heavy, 0);
            for (int k = 0; k < n_heavy; ++k) for (int g = tab[k]; g < tab[k + 1]; ++g) tab[n_heavy + 1 + g] = k;
        }
    }
    GM_TRY(gm_agg_schedule_flat(b, rows, win, heavy_host, n_heavy, tab, &out->d_sched, &out->len, s, sg));
    if (hub_part) {
        GM_TRY(gm_balloc(b, &out->d_hub, tab.size(), s));
        GM_TRY(sg->upload(out->d_hub, tab));
        GM_TRY(gm_balloc(b, &out->d_hub_scratch, (size_t)tab[n_heavy] * GM_AGG_HUB_LD, s));
        out->hub_part = hub_part; out->hub_words = (int32_t)tab.size(); out->parts = tab[n_heavy];
    }
    out->tab = tab;
    return GM_OK;
}

template <int LPR, int NCH>
static void launch_win(const AggK& a0, hipStream_t s) {
    AggK a = a0;
    // rows per wave window: 64 at most, halved until the launch has ~64k waves (or 2 rows are left).  Small windows
    // keep the rows in flight on an XCD within reach of its 4-MiB L2 -- a source row is gathered by ~2 destination rows
    // of the same subgraph, and the second gather only hits if it follows the first closely (measured on the 1.1 M-row
    // query batch: 4.2 -> 4.6 TB/s) -- and spread small batches (support sets, a 4-task shard) over the whole chip.
    a.win = (a.sched || a.rowlist) ? a0.win : gm_agg_window(a.rows, 0);
    const int RPB = a.win * (AGG_BLOCK / GM_WAVE);
    a.nblocks = (int)(((a.rowlist ? a.n_list : a.rows) + RPB - 1) / RPB);
    int grid = a.nblocks;
    if (a.sched) grid = GM_NXCD * a.sched_len;
    else if (a.n_heavy > 0) hipLaunchKernelGGL((k_agg_heavy<LPR, NCH>), dim3(a.n_heavy), dim3(AGG_HEAVY_BLOCK), 0, s, a);
    const int unr = gm_knob().agg_unr;
#define GM_AGG_CASE(U_, M_) if (unr == U_ * 10 + M_) { hipLaunchKernelGGL((k_agg_win<LPR, NCH, U_, M_>), dim3(grid), dim3(AGG_BLOCK), 0, s, a); return; }
    GM_AGG_CASE(1, 2) GM_AGG_CASE(1, 4) GM_AGG_CASE(2, 2) GM_AGG_CASE(2, 4) GM_AGG_CASE(4, 2) GM_AGG_CASE(4, 4) GM_AGG_CASE(3, 4) GM_AGG_CASE(2, 6) GM_AGG_CASE(2, 8) GM_AGG_CASE(1, 8)
#undef GM_AGG_CASE
    hipLaunchKernelGGL((k_agg_win<LPR, NCH, 2, 4>), dim3(grid), dim3(AGG_BLOCK), 0, s, a);
}

template <int VEC, int LPR>
static void launch_one(const AggK& a0, hipStream_t s) {
    AggK a = a0;
    constexpr int RPB = (GM_WAVE / LPR) * (AGG_BLOCK / GM_WAVE);
    a.nblocks = (int)((a.rows + RPB - 1) / RPB);
    hipLaunchKernelGGL((k_agg<VEC, LPR>), dim3(a.nblocks), dim3(AGG_BLOCK), 0, s, a);
}

// Non-temporal stores of the output: 0 never, 2 always, 1 (default) from 128 MB of output upwards -- a small output stays in the caches for
// the GEMM that reads it next (Tissue shape -1.3 %, FirstMM shape -1 %); at 146 MB (the support batch at task_num 32, the query batch of a
// 4-task shard) the two are within noise of each other, at 572k rows ordinary stores lose 1.5 %.
static int agg_nt(int64_t rows, int width) {
    const int k = gm_knob().agg_nt;
    return k == 1 ? (rows * (int64_t)width * 4 >= ((int64_t)128 << 20) ? 1 : 0) : (k ? 1 : 0);
}
static int agg_variant() { return gm_knob().agg_variant; }

int gm_launch_aggregate(const gm_agg_args& g, hipStream_t s) {
    if (g.rows <= 0) return GM_OK;
    AggK a{g.indptr, g.indices, g.x, g.x_row, g.ldx, g.s_in, g.s_out, g.mask_h, g.bias, g.bias_stride,
           g.set_row_off, g.n_sets, g.relu, g.out, g.rows, g.width, 0, g.mask_b, g.relu_bits, g.heavy, g.n_heavy, g.heavy_deg,
           g.sched, g.sched_len, agg_nt(g.rows, g.width), g.sched ? g.sched_win : 64,
           g.sched ? g.hub : nullptr, g.sched ? g.hub_scratch : nullptr, g.hub_part, GM_AGG_HUB_LD, g.e_w, g.x_idx, g.skip_on ? g.skip_lo : 1, g.skip_on ? g.skip_hi : 0,
           g.rowlist, g.n_list};
    if (g.rowlist) a.win = g.list_win;
    const bool vec4 = (g.width % 4 == 0) && (g.ldx % 4 == 0) && (((uintptr_t)g.x & 15) == 0) && (((uintptr_t)g.out & 15) == 0);
    const bool bias_ok = !g.bias || ((((uintptr_t)g.bias & 15) == 0) && (g.bias_stride % 4 == 0));
    const bool mask_ok = !g.mask_h || (((uintptr_t)g.mask_h & 15) == 0);
    GM_REQUIRE(!(g.mask_b || g.relu_bits) || vec4, GM_EINVAL, "aggregate: packed relu masks need width %% 4 == 0 and 16-byte aligned operands");
    const bool win = vec4 && bias_ok && mask_ok && (g.width == 64 || g.width == 128 || g.width == 256 || g.width == 512) && agg_variant() != 1;
    GM_REQUIRE(!g.rowlist || win, GM_EINVAL, "aggregate: a row list needs the window kernel");
    if (!win) { a.heavy = nullptr; a.n_heavy = 0; a.sched = nullptr; a.hub = nullptr; }      // the generic kernel walks every row itself
    if (win && gm_knob().agg_stream && gm_stream_ok(g)) return gm_launch_stream(g, a.nt, s);      // LDS-DMA stream kernel (row segments + hub parts in one launch)
    if (win) {
        if (g.width == 64) launch_win<16, 1>(a, s);
        else if (g.width == 128) launch_win<32, 1>(a, s);
        else if (g.width == 256) launch_win<64, 1>(a, s);
        else launch_win<64, 2>(a, s);
    } else if (vec4) {
        const int n4 = g.width / 4;
        if (n4 > 32) launch_one<4, 64>(a, s);
        else if (n4 > 16) launch_one<4, 32>(a, s);
        else if (n4 > 8) launch_one<4, 16>(a, s);
        else if (n4 > 4) launch_one<4, 8>(a, s);
        else if (n4 > 2) launch_one<4, 4>(a, s);
        else launch_one<4, 2>(a, s);
    } else {
        const int w = g.width;
        if (w > 32) launch_one<1, 64>(a, s);
        else if (w > 16) launch_one<1, 32>(a, s);
        else if (w > 8) launch_one<1, 16>(a, s);
        else if (w > 4) launch_one<1, 8>(a, s);
        else if (w > 2) launch_one<1, 4>(a, s);
        else launch_one<1, 2>(a, s);
    }
    GM_HIP(hipGetLastError());
    return GM_OK;
}

int gm_agg_stream_args(gm_agg_args& a, const gm_batch* b, int o, bool gather, hipStream_t s) {
    if (!gm_knob().agg_stream || a.s_out || a.bias || a.mask_h || a.mask_b || a.relu || a.relu_bits || a.rowlist || a.skip_on) return GM_OK;      // never a stream launch
    gm_batch::stream_pending& sp = b->spend[o];
    {
        // first launch of this orientation that can take the stream kernel: build its tables now, on the launch's stream (the batch's slabs take them; a
        // build on another stream than the batch's is ordered behind the batch's own by the caller's use of the batch, and marked for the frees).  The flag
        // is read under the lock: two streams may bring the same batch here at once (public gm_aggregate), and an uncontended lock costs nothing beside a launch
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (sp.pending) {
            gm_batch* mb = const_cast<gm_batch*>(b);
            gm_stager sg(s);
            GM_TRY(gm_stream_tables(mb, o, sp.hubs.data(), sp.deg.data(), (int)sp.hubs.size(), sp.n_parts, sp.has_tab ? &sp.tab : nullptr, s, &sg));
            sp.pending = false;
            sp.hubs = std::vector<int32_t>(); sp.deg = std::vector<int32_t>(); sp.tab = std::vector<int32_t>();
            gm_batch_mark_use(b, s);
        }
    }
    if (!b->d_sptr[o] || !b->d_sseg[o] || (gather && (o != 0 || !b->d_su_feat))) return GM_OK;
    a.stream = b; a.stream_o = o; a.stream_feat = gather ? 1 : 0;
    a.stream_xrows = gather ? b->store->total_nodes : b->rows;
    return GM_OK;
}

extern "C" int64_t gm_aggregate_bytes(const gm_batch_t* b, int32_t width) {
    // B_agg(n,e,F) = 4(n+1) [indptr] + 4e [indices] + 4n [norm] + 4nF [read X once] + 4nF [write Z]
    if (!b) return -1;
    return 4 * (b->rows + 1) + 4 * b->edges + 4 * b->rows + 8 * b->rows * (int64_t)width;
}

extern "C" int gm_aggregate(const gm_batch_t* b, int32_t transposed, int32_t gather, const float* x, int32_t width,
                            const float* s_in, const float* s_out, float* out, void* stream) {
    GM_REQUIRE(b && out && width >= 1, GM_EINVAL, "aggregate: bad arguments");
    GM_REQUIRE(gather || x, GM_EINVAL, "aggregate: x is NULL and gather == 0");
    GM_REQUIRE(!gather || width == b->store->feat_dim, GM_EINVAL, "aggregate: gather needs width == feat_dim");
    gm_agg_args a{};
    a.indptr = transposed ? b->d_indptr_t : b->d_indptr;
    a.indices = transposed ? b->d_indices_t : b->d_indices;
    a.x = gather ? b->store->d_feat : x;
    a.x_row = gather ? b->d_feat_row : nullptr;
    if (gather && !transposed) a.x_idx = b->d_efeat;
    if (s_in && s_in == b->d_norm) a.e_w = b->d_enorm[transposed ? 1 : 0];
    a.ldx = gather ? b->store->feat_ld : width; a.s_in = s_in; a.s_out = s_out; a.out = out; a.rows = b->rows; a.width = width;
    a.heavy = b->d_heavy[transposed ? 1 : 0]; a.n_heavy = b->n_heavy[transposed ? 1 : 0]; a.heavy_deg = b->heavy_deg;
    a.sched = b->d_sched[transposed ? 1 : 0]; a.sched_len = b->sched_len[transposed ? 1 : 0]; a.sched_win = b->sched_win; GM_TRY(gm_agg_hub(a, b, transposed ? 1 : 0, (hipStream_t)stream));
    if (a.e_w) GM_TRY(gm_agg_stream_args(a, b, transposed ? 1 : 0, gather != 0, (hipStream_t)stream));
    gm_prof_agg_begin((hipStream_t)stream, gm_aggregate_bytes(b, width));
    int rc = gm_launch_aggregate(a, (hipStream_t)stream);
    gm_prof_agg_end((hipStream_t)stream);
    gm_batch_mark_use(b, (hipStream_t)stream);
    return rc;
}
