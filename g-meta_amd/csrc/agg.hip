// Batched subgraph message passing on gfx950: segmented (CSR) sum over in-edges, i.e. DGL's
// graph.update_all(fn.copy_src, fn.sum) (learner.py:38-39,44-45) with GraphConv's degree
// normalisations fused in (learner.py:29-32,49).  HBM-bound: every row of X is read once from HBM
// (re-reads by other destination rows of the same subgraph hit the XCD's L2, because a subgraph's
// rows are contiguous and consecutive row blocks are mapped to the same XCD) and every row of the
// output is written once.  A lane group of LPR lanes owns one destination row; each lane holds VEC
// consecutive floats, so a 256-wide row is one 1-KiB coalesced access per wave.
#include <algorithm>
#include "gm_internal.h"

#define AGG_BLOCK 256

struct AggK {
    const int32_t* indptr; const int32_t* indices; const float* x; const int32_t* x_row; int64_t ldx;
    const float* s_in; const float* s_out; const float* mask_h; const float* bias; int64_t bias_stride;
    const int32_t* set_row_off; int n_sets; int relu; float* out; int64_t rows; int width; int nblocks;
};

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
            int u0 = a.indices[e], u1 = a.indices[e + 1], u2 = a.indices[e + 2], u3 = a.indices[e + 3];
            const float w0 = a.s_in ? a.s_in[u0] : 1.f, w1 = a.s_in ? a.s_in[u1] : 1.f, w2 = a.s_in ? a.s_in[u2] : 1.f, w3 = a.s_in ? a.s_in[u3] : 1.f;
            if (a.x_row) { u0 = a.x_row[u0]; u1 = a.x_row[u1]; u2 = a.x_row[u2]; u3 = a.x_row[u3]; }
            const V v0 = *reinterpret_cast<const V*>(a.x + (int64_t)u0 * a.ldx + c0);
            const V v1 = *reinterpret_cast<const V*>(a.x + (int64_t)u1 * a.ldx + c0);
            const V v2 = *reinterpret_cast<const V*>(a.x + (int64_t)u2 * a.ldx + c0);
            const V v3 = *reinterpret_cast<const V*>(a.x + (int64_t)u3 * a.ldx + c0);
            vfma(acc0, v0, w0); vfma(acc1, v1, w1); vfma(acc0, v2, w2); vfma(acc1, v3, w3);
        }
        for (; e < e1; ++e) {
            int u = a.indices[e];
            const float w = a.s_in ? a.s_in[u] : 1.f;
            if (a.x_row) u = a.x_row[u];
            vfma(acc0, *reinterpret_cast<const V*>(a.x + (int64_t)u * a.ldx + c0), w);
        }
        V res;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float v = (vget(acc0, k) + vget(acc1, k)) * so;
            if (a.bias) v += a.bias[(int64_t)set * a.bias_stride + c0 + k];
            if (a.relu) v = v < 0.f ? 0.f : v;      // NaN propagates like torch relu (meta.py:163 guard)
            if (a.mask_h) v = a.mask_h[row * a.width + c0 + k] > 0.f ? v : 0.f;
            vset(res, k, v);
        }
        *reinterpret_cast<V*>(a.out + row * a.width + c0) = res;
    }
}

template <int VEC, int LPR>
static void launch_one(const AggK& a0, hipStream_t s) {
    AggK a = a0;
    constexpr int RPB = (GM_WAVE / LPR) * (AGG_BLOCK / GM_WAVE);
    a.nblocks = (int)((a.rows + RPB - 1) / RPB);
    hipLaunchKernelGGL((k_agg<VEC, LPR>), dim3(a.nblocks), dim3(AGG_BLOCK), 0, s, a);
}

int gm_launch_aggregate(const gm_agg_args& g, hipStream_t s) {
    if (g.rows <= 0) return GM_OK;
    AggK a{g.indptr, g.indices, g.x, g.x_row, g.ldx, g.s_in, g.s_out, g.mask_h, g.bias, g.bias_stride,
           g.set_row_off, g.n_sets, g.relu, g.out, g.rows, g.width, 0};
    const bool vec4 = (g.width % 4 == 0) && (g.ldx % 4 == 0) && (((uintptr_t)g.x & 15) == 0) && (((uintptr_t)g.out & 15) == 0);
    if (vec4) {
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
    a.ldx = width; a.s_in = s_in; a.s_out = s_out; a.out = out; a.rows = b->rows; a.width = width;
    gm_prof_agg_begin((hipStream_t)stream, gm_aggregate_bytes(b, width));
    int rc = gm_launch_aggregate(a, (hipStream_t)stream);
    gm_prof_agg_end((hipStream_t)stream);
    return rc;
}
