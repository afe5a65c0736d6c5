// h-hop subgraph extraction, node sampling, induced-subgraph CSR build and batching on gfx950.
// Replaces Subgraphs.generate_subgraph / generate_subgraph_link_pred (sdp.py:295-346) and
// dgl.batch (sdp.py:399-406).  One workgroup per subgraph; the membership set is an LDS bitmap
// over the parent graph's nodes, so the node list comes out in ascending order for free and
// local ids are prefix popcounts.  Edge lists are read from HBM coalesced (wave per frontier node).
#include <algorithm>
#include <initializer_list>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include "gm_internal.h"

#define EX_BLOCK 512
#define EX_WAVES (EX_BLOCK / GM_WAVE)

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t sample_salt(uint64_t seed, int g, int i, int j) {
    uint32_t s = lowbias32((uint32_t)(seed & 0xffffffffu) ^ 0x9E3779B9u);
    s = lowbias32(s ^ (uint32_t)(seed >> 32));
    s = lowbias32(s + (uint32_t)g * 0x85EBCA6Bu);
    s = lowbias32(s ^ (uint32_t)i);
    s = lowbias32(s + (uint32_t)(j + 1) * 0xC2B2AE35u);
    return s;
}
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Bitmap words live in LDS (G = false) or, for parent graphs too large for it, in a per-workgroup slab of global memory
// (G = true).  In the global case every word access is an agent-scope relaxed atomic: the bits are set with L2 atomics, and a
// CU's L1 is not coherent with those, so plain loads could return stale words.
template <bool G> __device__ __forceinline__ uint32_t wld(const uint32_t* p) {
    if (G) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool G> __device__ __forceinline__ void wst(uint32_t* p, uint32_t v) {
    if (G) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool G> __device__ __forceinline__ bool bit_test(const uint32_t* bm, int v) { return (wld<G>(&bm[v >> 5]) >> (v & 31)) & 1u; }
__device__ __forceinline__ void bit_set(uint32_t* bm, int v) { atomicOr(&bm[v >> 5], 1u << (v & 31)); }
// The per-word prefix counts are 16-bit where a subgraph holds fewer than 65,536 nodes and the bitmaps live in LDS (PT = uint16_t; round 6): with the
// 21-KiB membership bitmap of a 169 k-node parent that is 35 instead of 45 KiB per workgroup -- four resident workgroups per CU instead of three.
template <bool G, typename PT> __device__ __forceinline__ int pref_ld(const PT* p) {
    if constexpr (sizeof(PT) == 4) return (int)wld<G>(reinterpret_cast<const uint32_t*>(p)); else return (int)*p;
}
template <bool G, typename PT> __device__ __forceinline__ int bit_rank(const uint32_t* bm, const PT* pref, int v) {
    return pref_ld<G, PT>(&pref[v >> 5]) + __popc(wld<G>(&bm[v >> 5]) & ((1u << (v & 31)) - 1u));
}

// Exclusive scan of part[0..EX_BLOCK) in LDS by wave 0; returns the total through *total (LDS).
__device__ __forceinline__ void scan_partials(int* part, int* total) {
    __syncthreads();
    if (threadIdx.x < GM_WAVE) {
        const int l = threadIdx.x;
        constexpr int PER = EX_BLOCK / GM_WAVE;
        int loc[PER], s = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { loc[k] = s; s += part[l * PER + k]; }
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (l >= o) inc += t; }
        const int base = inc - s;
#pragma unroll
        for (int k = 0; k < PER; ++k) part[l * PER + k] = base + loc[k];
        if (l == 63) *total = inc;
    }
    __syncthreads();
}

// pref[w] = number of set bits in seen[0..w); returns the total.
template <bool G, typename PT>
__device__ __forceinline__ int bitmap_prefix(const uint32_t* seen, PT* pref, int W, int* part, int* total) {
    const int chunk = (W + EX_BLOCK - 1) / EX_BLOCK;
    const int w0 = threadIdx.x * chunk, w1 = min(W, w0 + chunk);
    int s = 0;
    for (int w = w0; w < w1; ++w) s += __popc(wld<G>(&seen[w]));
    part[threadIdx.x] = s;
    scan_partials(part, total);
    int run = part[threadIdx.x];
    for (int w = w0; w < w1; ++w) {
        if constexpr (sizeof(PT) == 4) wst<G>(reinterpret_cast<uint32_t*>(&pref[w]), (uint32_t)run); else pref[w] = (PT)run;
        run += __popc(wld<G>(&seen[w]));
    }
    __syncthreads();
    return *total;
}

struct ExStore {
    const int64_t* node_off;
    const int64_t* in_ptr; const int32_t* in_idx;
    const int64_t* out_ptr; const int32_t* out_idx;
    int sym;       // the out-CSR is element for element the in-CSR (an undirected graph stored in both directions, rows ascending): the by-source
                   // CSR of an induced subgraph then IS its by-destination CSR, so the adjacency lists are walked once instead of twice
};

// Adjacency walks are latency chains (row bounds -> neighbour ids -> bitmap word), so a wave takes EIGHT nodes at a time, one per group of eight
// lanes (the median parent degree is below 16), two neighbour loads in flight per lane; nodes with more than EX_BIG_DEG neighbours are left to a
// second pass in which a whole wave walks one node (a hub in one group would stall the other seven).
#define EX_GL 8
#define EX_GROUPS (GM_WAVE / EX_GL)
#define EX_BIG_DEG 256
#define EX_INFL 4          // neighbour loads in flight per lane (round 6: 2 before -- a node of 17..32 neighbours took two dependent round trips, now one)
__device__ __forceinline__ int group_sum(int v) {
#pragma unroll
    for (int o = EX_GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// number of neighbours of v (list ptr/idx) inside the bitmap; the eight lanes of a group call it together (gl = lane within the group)
template <bool G>
__device__ __forceinline__ int group_count(const int64_t a, const int64_t b, const int32_t* idx, const uint32_t* seen, int gl) {
    int c = 0;
    for (int64_t q = a + gl; __any(q < b); q += EX_INFL * EX_GL) {
        int u[EX_INFL];
#pragma unroll
        for (int k = 0; k < EX_INFL; ++k) u[k] = q + k * EX_GL < b ? idx[q + k * EX_GL] : -1;
#pragma unroll
        for (int k = 0; k < EX_INFL; ++k) if (u[k] >= 0) c += bit_test<G>(seen, u[k]);
    }
    return group_sum(c);
}

// Hub nodes (more than EX_BIG_DEG neighbours: thousands in a preferential-attachment parent, and every 2-hop neighbourhood holds dozens of them --
// most of a subgraph's walk volume) are walked by a whole wave, EX_WINFL x 64 neighbour ids in flight (round 6: 64 before, a round trip per 64 ids).
#define EX_WINFL 4
// Marks every in-neighbour of v (graph-local id) in `seen`; called by a whole wave.
__device__ __forceinline__ void wave_mark_preds(const ExStore& S, int64_t base, int v, uint32_t* seen, int lane) {
    const int64_t p0 = S.in_ptr[base + v], p1 = S.in_ptr[base + v + 1];
    for (int64_t q = p0 + lane; q < p1; q += EX_WINFL * GM_WAVE) {
        int u[EX_WINFL];
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) u[k] = q + k * GM_WAVE < p1 ? S.in_idx[q + k * GM_WAVE] : -1;
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) if (u[k] >= 0) bit_set(seen, u[k]);
    }
}
// neighbours of a hub node (list [a, b) of idx) inside the bitmap, per lane (the caller adds the lanes up)
template <bool G>
__device__ __forceinline__ int wave_count(const int64_t a, const int64_t b, const int32_t* idx, const uint32_t* seen, int lane) {
    int c = 0;
    for (int64_t q = a + lane; q < b; q += EX_WINFL * GM_WAVE) {
        int u[EX_WINFL];
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) u[k] = q + k * GM_WAVE < b ? idx[q + k * GM_WAVE] : -1;
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) if (u[k] >= 0) c += bit_test<G>(seen, u[k]);
    }
    return c;
}

// Phase A: node set (BFS or given), sampling, sorted node list, induced in/out degrees.
// P16: 16-bit prefix words (LDS bitmaps, subgraphs below 65,536 nodes).  NEEDX: keep the `expanded` bitmap that de-duplicates frontier expansions -- needed from
// the third hop on (a hop-2 node is reached through many hop-1 nodes); with two hops it only catches parallel edges of the centre, and without it the
// region behind `seen` shrinks to the 16-bit prefix words.
template <bool G, bool P16 = false, bool NEEDX = true>
__global__ __launch_bounds__(EX_BLOCK) void k_nodes(ExStore S, const gm_seed_t* seeds, int n_seeds, int h, int sample_n,
                                                    uint64_t rng_seed, int link, const int32_t* given, const int64_t* given_off,
                                                    int cap, int32_t* nodes_slab, int32_t* degi_slab, int32_t* dego_slab,
                                                    int32_t* n_sub, int32_t* e_sub, int Wmax, uint32_t* gbits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* seen = G ? gbits + (size_t)blockIdx.x * 2 * Wmax : lds;
    static_assert(!(G && P16), "16-bit prefix words live in LDS");
    typedef typename std::conditional<P16, uint16_t, uint32_t>::type PT;
    uint32_t* xbm = seen + Wmax;              // the `expanded` bitmap of the BFS (NEEDX); the same region holds the prefix words afterwards
    PT* pref = reinterpret_cast<PT*>(xbm);
    const int PW = (P16 && !NEEDX) ? (Wmax + 1) / 2 : Wmax;      // words of that region
    int* part = (int*)(G ? lds : lds + Wmax + PW);      // [EX_BLOCK]
    uint32_t* hist = (uint32_t*)(part + EX_BLOCK);   // [256]
    int* sc = (int*)(hist + 256);             // scalars: 0 total, 1 kk, 2 prefix, 3 edge count in, 4 edge count out
    const int seed = blockIdx.x;
    if (seed >= n_seeds) return;
    const int g = seeds[seed].graph, ci = seeds[seed].i, cj = link ? seeds[seed].j : -1;
    const int64_t base = S.node_off[g];
    const int n = (int)(S.node_off[g + 1] - base);
    const int W = (n + 31) >> 5;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;

    for (int w = tid; w < W; w += EX_BLOCK) { wst<G>(&seen[w], 0u); if (NEEDX) wst<G>(&xbm[w], 0u); }
    if (tid < 8) sc[tid] = 0;
    __syncthreads();
    if (given) {
        const int64_t a = given_off[seed], b = given_off[seed + 1];
        for (int64_t k = a + tid; k < b; k += EX_BLOCK) bit_set(seen, given[k]);
        __syncthreads();
    } else {
        const int H = link ? 2 : h;
        const int64_t p0 = S.in_ptr[base + ci], p1 = S.in_ptr[base + ci + 1];
        if (tid == 0) { bit_set(seen, ci); if (NEEDX) bit_set(xbm, ci); }
        for (int64_t q = p0 + tid; q < p1; q += EX_BLOCK) bit_set(seen, S.in_idx[q]);          // hop 1 (sdp.py:301,305,308)
        __syncthreads();
        if (H >= 2) {                                                                         // hop 2 (sdp.py:302,309)
            // eight frontier nodes per wave at a time; frontier hubs go to the list in `part` and get a whole wave each afterwards
            const int grp = lane / EX_GL, gl = lane % EX_GL;
            int* big = part; int* nbig = &sc[5];
            for (int64_t q0 = p0 + (int64_t)wave * EX_GROUPS; q0 < p1; q0 += (int64_t)EX_WAVES * EX_GROUPS) {
                const int64_t q = q0 + grp;
                int v = -1; int64_t a = 0, b = 0;
                if (q < p1) {
                    v = S.in_idx[q];
                    int first = 1;
                    if constexpr (NEEDX) {
                        first = 0;
                        if (gl == 0) { const uint32_t bit = 1u << (v & 31); first = !(atomicOr(&xbm[v >> 5], bit) & bit); }
                        first = __shfl(first, grp * EX_GL, 64);
                    }
                    if (first) { a = S.in_ptr[base + v]; b = S.in_ptr[base + v + 1]; }
                    if (b - a > EX_BIG_DEG) {
                        int slot = EX_BLOCK;
                        if (gl == 0) slot = atomicAdd(nbig, 1);
                        slot = __shfl(slot, grp * EX_GL, 64);
                        if (slot < EX_BLOCK) { if (gl == 0) big[slot] = v; b = a; }      // (list full: the group walks it itself)
                    }
                }
                for (int64_t r = a + gl; __any(r < b); r += EX_INFL * EX_GL) {
                    int u[EX_INFL];
#pragma unroll
                    for (int k = 0; k < EX_INFL; ++k) u[k] = r + k * EX_GL < b ? S.in_idx[r + k * EX_GL] : -1;
#pragma unroll
                    for (int k = 0; k < EX_INFL; ++k) if (u[k] >= 0) bit_set(seen, u[k]);
                }
            }
            __syncthreads();
            const int nb = min(*nbig, EX_BLOCK);
            for (int k = wave; k < nb; k += EX_WAVES) wave_mark_preds(S, base, big[k], seen, lane);
            __syncthreads();
        }
        if (H >= 3) {                                                                         // hop 3 (sdp.py:310)
            for (int64_t q = p0 + wave; q < p1; q += EX_WAVES) {
                const int v = S.in_idx[q];
                const int64_t a = S.in_ptr[base + v], b = S.in_ptr[base + v + 1];
                for (int64_t r = a; r < b; r += GM_WAVE) {
                    int u = -1, first = 0;
                    if (r + lane < b) {
                        u = S.in_idx[r + lane];
                        const uint32_t bit = 1u << (u & 31);
                        first = !(atomicOr(&xbm[u >> 5], bit) & bit);
                    }
                    unsigned long long m = __ballot(first);
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        wave_mark_preds(S, base, __shfl(u, src, 64), seen, lane);
                    }
                }
            }
            __syncthreads();
        }
        if (link) {                                                                           // j side: 1 hop only (sdp.py:331-333)
            if (tid == 0) bit_set(seen, cj);
            const int64_t a = S.in_ptr[base + cj], b = S.in_ptr[base + cj + 1];
            for (int64_t q = a + tid; q < b; q += EX_BLOCK) bit_set(seen, S.in_idx[q]);
            __syncthreads();
        }
        // ---- count, and sample if above the threshold (strict '>' at sdp.py:312,337)
        int c = 0;
        for (int w = tid; w < W; w += EX_BLOCK) c += __popc(wld<G>(&seen[w]));
        c = wave_sum(c);
        if (lane == 0) atomicAdd(&sc[0], c);
        __syncthreads();
        const int count = sc[0];
        __syncthreads();
        if (count > sample_n) {
            // keep the sample_n nodes with the smallest key(node) = lowbias32(node ^ salt): 4-pass radix select.
            const uint32_t salt = sample_salt(rng_seed, g, ci, cj);
            if (tid == 0) { sc[1] = sample_n; sc[2] = 0; }
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int k = tid; k < 256; k += EX_BLOCK) hist[k] = 0;
                __syncthreads();
                const uint32_t prefix = (uint32_t)sc[2];
                for (int w = tid; w < W; w += EX_BLOCK) {
                    uint32_t bits = wld<G>(&seen[w]);
                    while (bits) {
                        const int b = __ffs(bits) - 1; bits &= bits - 1;
                        const uint32_t key = lowbias32((uint32_t)(w * 32 + b) ^ salt);
                        if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    int kk = sc[1], d = 0;
                    while (d < 255 && (int)hist[d] < kk) { kk -= (int)hist[d]; ++d; }
                    sc[1] = kk; sc[2] = (int)(prefix | ((uint32_t)d << shift));
                }
                __syncthreads();
            }
            const uint32_t tau = (uint32_t)sc[2];
            for (int w = tid; w < W; w += EX_BLOCK) {
                uint32_t bits = wld<G>(&seen[w]), keep = 0;
                while (bits) {
                    const int b = __ffs(bits) - 1; bits &= bits - 1;
                    if (lowbias32((uint32_t)(w * 32 + b) ^ salt) <= tau) keep |= 1u << b;
                }
                wst<G>(&seen[w], keep);
            }
            __syncthreads();
            if (tid == 0) { bit_set(seen, ci); if (cj >= 0) bit_set(seen, cj); }   // np.unique(np.append(., centres))
            __syncthreads();
        }
    }
    // ---- ascending node list + local-id prefix
    const int ns = bitmap_prefix<G, PT>(seen, pref, W, part, &sc[0]);
    if (ns > cap) { if (tid == 0) { n_sub[seed] = -ns; e_sub[seed] = 0; } return; }   // host reports the error
    int32_t* nodes = nodes_slab + (int64_t)seed * cap;
    for (int w = tid; w < W; w += EX_BLOCK) {
        uint32_t bits = wld<G>(&seen[w]);
        int r = pref_ld<G, PT>(&pref[w]);
        while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; nodes[r++] = w * 32 + b; }
    }
    __syncthreads();
    // ---- induced in/out degree of every selected node (eight nodes per wave at a time, hubs by a whole wave afterwards)
    {
        const int grp = lane / EX_GL, gl = lane % EX_GL;
        int* big = part; int* nbig = &sc[5];
        if (tid == 0) *nbig = 0;
        __syncthreads();
        int ein = 0, eout = 0;
        int32_t* degi = degi_slab + (int64_t)seed * cap; int32_t* dego = dego_slab + (int64_t)seed * cap;
        // The walk of a batch of eight nodes is a chain node id -> row bounds -> neighbour ids -> bitmap word.  The first two links are taken off the
        // chain (round 6): the node ids are fetched TWO batches ahead and the row bounds ONE batch ahead, so an iteration issues three independent
        // groups of loads and waits one round trip instead of three.
        constexpr int RSTEP = EX_WAVES * EX_GROUPS;
        int r = wave * EX_GROUPS + grp;
        int v1 = r < ns ? nodes[r] : -1;                                              // node of the NEXT batch
        int v2 = r + RSTEP < ns ? nodes[r + RSTEP] : -1;                              // ... of the one after
        int64_t nia = 0, nib = 0, noa = 0, nob = 0;
        if (v1 >= 0) { nia = S.in_ptr[base + v1]; nib = S.in_ptr[base + v1 + 1]; if (!S.sym) { noa = S.out_ptr[base + v1]; nob = S.out_ptr[base + v1 + 1]; } }
        for (int r0 = wave * EX_GROUPS; r0 < ns; r0 += RSTEP, r += RSTEP) {
            int64_t ia = nia, ib = nib, oa = noa, ob = nob;
            const bool have = v1 >= 0;
            v1 = v2;
            v2 = r + 2 * RSTEP < ns ? nodes[r + 2 * RSTEP] : -1;
            nia = nib = noa = nob = 0;
            if (v1 >= 0) { nia = S.in_ptr[base + v1]; nib = S.in_ptr[base + v1 + 1]; if (!S.sym) { noa = S.out_ptr[base + v1]; nob = S.out_ptr[base + v1 + 1]; } }
            bool later = false;                                                          // a hub: its degrees come from the second pass
            if (have) {
                if (ib - ia > EX_BIG_DEG || ob - oa > EX_BIG_DEG) {
                    int slot = EX_BLOCK;
                    if (gl == 0) slot = atomicAdd(nbig, 1);
                    slot = __shfl(slot, grp * EX_GL, 64);
                    if (slot < EX_BLOCK) { if (gl == 0) big[slot] = r; later = true; ia = ib = oa = ob = 0; }      // (list full: the group walks it itself)
                }
            }
            const int ci_ = group_count<G>(ia, ib, S.in_idx, seen, gl);
            const int co_ = S.sym ? ci_ : group_count<G>(oa, ob, S.out_idx, seen, gl);
            if (have && gl == 0 && !later) { degi[r] = ci_; dego[r] = co_; ein += ci_; eout += co_; }
        }
        __syncthreads();
        const int nb = min(*nbig, EX_BLOCK);
        for (int k = wave; k < nb; k += EX_WAVES) {
            const int r = big[k], v = nodes[r];
            int ci_ = 0, co_ = 0;
            ci_ = wave_count<G>(S.in_ptr[base + v], S.in_ptr[base + v + 1], S.in_idx, seen, lane);
            if (!S.sym) co_ = wave_count<G>(S.out_ptr[base + v], S.out_ptr[base + v + 1], S.out_idx, seen, lane);
            ci_ = wave_sum(ci_); co_ = S.sym ? ci_ : wave_sum(co_);
            if (lane == 0) { degi[r] = ci_; dego[r] = co_; ein += ci_; eout += co_; }
        }
        if (ein | eout) { atomicAdd(&sc[3], ein); atomicAdd(&sc[4], eout); }
    }
    __syncthreads();
    if (tid == 0) { n_sub[seed] = ns; e_sub[seed] = (sc[3] == sc[4]) ? sc[3] : -1; }
}

// In-block exclusive scan of deg[0..ns) (global) into ptr[row0 + r] = e0 + excl.
__device__ __forceinline__ void scan_degrees(const int32_t* deg, int ns, int32_t* ptr_out, int e0, int* part, int* total) {
    const int chunk = (ns + EX_BLOCK - 1) / EX_BLOCK;
    const int r0 = threadIdx.x * chunk, r1 = min(ns, r0 + chunk);
    int s = 0;
    for (int r = r0; r < r1; ++r) s += deg[r];
    part[threadIdx.x] = s;
    scan_partials(part, total);
    int run = e0 + part[threadIdx.x];
    for (int r = r0; r < r1; ++r) { ptr_out[r] = run; run += deg[r]; }
}

// Ordered compaction of the neighbours of v that are inside the subgraph, remapped to batch rows (out2: optional second copy -- the
// by-source CSR of a symmetric parent).
template <bool G, typename PT>
__device__ __forceinline__ void wave_fill_row(const int64_t* ptr, const int32_t* idx, int64_t base, int v, const uint32_t* seen,
                                              const PT* pref, int row0, int32_t* out, int32_t* out2, int pos, int lane) {
    const int64_t a = ptr[base + v], b = ptr[base + v + 1];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t q = a; q < b; q += EX_WINFL * GM_WAVE) {
        int u[EX_WINFL];
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) u[k] = q + k * GM_WAVE + lane < b ? idx[q + k * GM_WAVE + lane] : -1;
#pragma unroll
        for (int k = 0; k < EX_WINFL; ++k) {                     // in list order
            const int hit = u[k] >= 0 && bit_test<G>(seen, u[k]);
            const unsigned long long m = __ballot(hit);
            if (hit) { const int x = row0 + bit_rank<G, PT>(seen, pref, u[k]), p = pos + __popcll(m & lt); out[p] = x; if (out2) out2[p] = x; }
            pos += __popcll(m);
        }
    }
}
// The same for one node per group of eight lanes (all lanes of the wave call it; a group without a node passes a == b)
template <bool G, typename PT>
__device__ __forceinline__ void group_fill_row(const int64_t a, const int64_t b, const int32_t* idx, const uint32_t* seen, const PT* pref, int row0,
                                               int32_t* out, int32_t* out2, int pos, int grp, int gl) {
    const unsigned lt = (1u << gl) - 1u;
    for (int64_t q = a + gl; __any(q < b); q += EX_INFL * EX_GL) {
        int u[EX_INFL];
#pragma unroll
        for (int k = 0; k < EX_INFL; ++k) u[k] = q + k * EX_GL < b ? idx[q + k * EX_GL] : -1;
#pragma unroll
        for (int k = 0; k < EX_INFL; ++k) {                      // in list order: the k-th batch of eight neighbours after the (k-1)-th
            const int h = u[k] >= 0 && bit_test<G>(seen, u[k]);
            const unsigned bm = (unsigned)(__ballot(h) >> (grp * EX_GL)) & 0xffu;
            if (h) { const int x = row0 + bit_rank<G, PT>(seen, pref, u[k]), p = pos + __popc(bm & lt); out[p] = x; if (out2) out2[p] = x; }
            pos += __popc(bm);
        }
    }
}

// Phase B: write the batched CSR (by destination and by source), parents, feature rows, norm, centres.
// One launch may fill TWO batches (gm_extract_pair: the support and the query batch of a meta-batch): seeds [0, split) belong to o0, the rest to o1, each
// batch numbered from its own subgraph 0.  (The 288-subgraph support launch of a 32-task meta-batch was ~0.1 ms of one workgroup's latency chain on an otherwise
// empty GPU; as the head of the 2,592-workgroup joint launch it costs nothing.)
struct FillOut {
    const int32_t* sub_off; const int32_t* sub_eoff; int32_t* parent; int32_t* feat_row; float* norm;
    int32_t* indptr; int32_t* indices; int32_t* indptr_t; int32_t* indices_t; int32_t* centre;
};
template <bool G, bool P16 = false>
__global__ __launch_bounds__(EX_BLOCK) void k_fill(ExStore S, const gm_seed_t* seeds, int n_seeds, int link, int cap,
                                                   const int32_t* nodes_slab, const int32_t* degi_slab, const int32_t* dego_slab,
                                                   FillOut o0, FillOut o1, int split, int Wmax, uint32_t* gbits, const int32_t* order) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* seen = G ? gbits + (size_t)blockIdx.x * 2 * Wmax : lds;
    static_assert(!(G && P16), "16-bit prefix words live in LDS");
    typedef typename std::conditional<P16, uint16_t, uint32_t>::type PT;
    PT* pref = reinterpret_cast<PT*>(seen + Wmax);
    int* part = (int*)(G ? lds : lds + Wmax + (P16 ? (Wmax + 1) / 2 : Wmax));
    int* sc = part + EX_BLOCK;
    if ((int)blockIdx.x >= n_seeds) return;
    // workgroups start in blockIdx order: the subgraphs with the most edges first (host order, by the sizes the count pass brought), so that the last
    // of the ~2.5 rounds a 32-task meta-batch makes over the chip's workgroup slots is made of short ones
    const int seed = order ? order[blockIdx.x] : (int)blockIdx.x;
    const bool second = seed >= split;
    const int ls = second ? seed - split : seed, nl = second ? n_seeds - split : split;      // subgraph number inside its batch; the batch's subgraph count
    const int32_t* sub_off = second ? o1.sub_off : o0.sub_off; const int32_t* sub_eoff = second ? o1.sub_eoff : o0.sub_eoff;
    int32_t* parent = second ? o1.parent : o0.parent; int32_t* feat_row = second ? o1.feat_row : o0.feat_row; float* norm = second ? o1.norm : o0.norm;
    int32_t* indptr = second ? o1.indptr : o0.indptr; int32_t* indices = second ? o1.indices : o0.indices;
    int32_t* indptr_t = second ? o1.indptr_t : o0.indptr_t; int32_t* indices_t = second ? o1.indices_t : o0.indices_t; int32_t* centre = second ? o1.centre : o0.centre;
    const int g = seeds[seed].graph;
    const int64_t base = S.node_off[g];
    const int n = (int)(S.node_off[g + 1] - base);
    const int W = (n + 31) >> 5;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int row0 = sub_off[ls], ns = sub_off[ls + 1] - row0, e0 = sub_eoff[ls];
    const int32_t* nodes = nodes_slab + (int64_t)seed * cap;
    const int32_t* degi = degi_slab + (int64_t)seed * cap;
    const int32_t* dego = dego_slab + (int64_t)seed * cap;

    for (int w = tid; w < W; w += EX_BLOCK) wst<G>(&seen[w], 0u);
    __syncthreads();
    for (int r = tid; r < ns; r += EX_BLOCK) {
        const int v = nodes[r];
        bit_set(seen, v);
        parent[row0 + r] = v;
        feat_row[row0 + r] = (int32_t)(base + v);
        const int d = degi[r];
        norm[row0 + r] = 1.0f / sqrtf((float)(d > 1 ? d : 1));          // in_degrees().clamp(min=1) ** -0.5 (learner.py:29)
    }
    __syncthreads();
    bitmap_prefix<G, PT>(seen, pref, W, part, &sc[0]);
    scan_degrees(degi, ns, indptr + row0, e0, part, &sc[0]);
    scan_degrees(dego, ns, indptr_t + row0, e0, part, &sc[1]);
    if (ls == nl - 1 && tid == 0) { indptr[row0 + ns] = e0 + sc[0]; indptr_t[row0 + ns] = e0 + sc[1]; }
    __syncthreads();
    if (tid == 0) {
        const int nc = link ? 2 : 1;
        centre[ls * nc] = bit_rank<G, PT>(seen, pref, seeds[seed].i);
        if (link) centre[ls * nc + 1] = bit_rank<G, PT>(seen, pref, seeds[seed].j);
    }
    // eight rows per wave at a time (one per group of eight lanes); hub nodes by a whole wave afterwards.  A symmetric parent fills both
    // orientations from the one walk.
    {
        const int grp = lane / EX_GL, gl = lane % EX_GL;
        int* big = part; int* nbig = &sc[2];
        if (tid == 0) *nbig = 0;
        __syncthreads();
        int32_t* ind2 = S.sym ? indices_t : nullptr;
        // node ids two batches ahead, row bounds and output offsets one batch ahead (see k_nodes)
        constexpr int RSTEP = EX_WAVES * EX_GROUPS;
        int r = wave * EX_GROUPS + grp;
        int v1 = r < ns ? nodes[r] : -1;
        int v2 = r + RSTEP < ns ? nodes[r + RSTEP] : -1;
        int64_t nia = 0, nib = 0, noa = 0, nob = 0; int npi = 0, npo = 0;
        if (v1 >= 0) {
            nia = S.in_ptr[base + v1]; nib = S.in_ptr[base + v1 + 1]; npi = indptr[row0 + r];
            if (!S.sym) { noa = S.out_ptr[base + v1]; nob = S.out_ptr[base + v1 + 1]; npo = indptr_t[row0 + r]; }
        }
        for (int r0 = wave * EX_GROUPS; r0 < ns; r0 += RSTEP, r += RSTEP) {
            int64_t ia = nia, ib = nib, oa = noa, ob = nob; const int pi = npi, po = npo;
            const bool have = v1 >= 0;
            v1 = v2;
            v2 = r + 2 * RSTEP < ns ? nodes[r + 2 * RSTEP] : -1;
            nia = nib = noa = nob = 0; npi = npo = 0;
            if (v1 >= 0) {
                nia = S.in_ptr[base + v1]; nib = S.in_ptr[base + v1 + 1]; npi = indptr[row0 + r + RSTEP];
                if (!S.sym) { noa = S.out_ptr[base + v1]; nob = S.out_ptr[base + v1 + 1]; npo = indptr_t[row0 + r + RSTEP]; }
            }
            if (have) {
                if (ib - ia > EX_BIG_DEG || ob - oa > EX_BIG_DEG) {
                    int slot = EX_BLOCK;
                    if (gl == 0) slot = atomicAdd(nbig, 1);
                    slot = __shfl(slot, grp * EX_GL, 64);
                    if (slot < EX_BLOCK) { if (gl == 0) big[slot] = r; ia = ib = oa = ob = 0; }      // (list full: the group walks it itself)
                }
            }
            group_fill_row<G, PT>(ia, ib, S.in_idx, seen, pref, row0, indices, ind2, pi, grp, gl);
            if (!S.sym) group_fill_row<G, PT>(oa, ob, S.out_idx, seen, pref, row0, indices_t, nullptr, po, grp, gl);
        }
        __syncthreads();
        const int nb = min(*nbig, EX_BLOCK);
        for (int k = wave; k < nb; k += EX_WAVES) {
            const int r = big[k], v = nodes[r];
            wave_fill_row<G, PT>(S.in_ptr, S.in_idx, base, v, seen, pref, row0, indices, ind2, indptr[row0 + r], lane);
            if (!S.sym) wave_fill_row<G, PT>(S.out_ptr, S.out_idx, base, v, seen, pref, row0, indices_t, nullptr, indptr_t[row0 + r], lane);
        }
    }
}

// per-edge tables: source norm (both orientations) and source feature row (forward orientation)
// Row gains of the aggregates (gm_batch::d_gain, zeroed): bit patterns of non-negative floats order as unsigned integers.
__global__ void k_gains(const int32_t* indptr, const int32_t* indices, const int32_t* indptr_t, const float* norm, int64_t rows, unsigned* gain) {
    float g0 = 0.f, g1 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        // hub rows: norm <= 1, so the degree bounds the sum (their neighbours are mostly low-degree rows: within ~2x of it)
        const int e0 = indptr[i], e1 = indptr[i + 1];
        float s = (float)(e1 - e0);
        if (e1 - e0 <= 32) { s = 0.f; for (int e = e0; e < e1; ++e) s += norm[indices[e]]; }
        g0 = fmaxf(g0, s);
        g1 = fmaxf(g1, norm[i] * (float)(indptr_t[i + 1] - indptr_t[i]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { g0 = fmaxf(g0, __shfl_xor(g0, o)); g1 = fmaxf(g1, __shfl_xor(g1, o)); }
    if ((threadIdx.x & 63) == 0) { if (g0 > 0.f) atomicMax(gain, __float_as_uint(g0)); if (g1 > 0.f) atomicMax(gain + 1, __float_as_uint(g1)); }
}

__global__ void k_edge_tables(const int32_t* indices, const int32_t* indices_t, int64_t edges, const float* norm, const int32_t* feat_row,
                              float* enorm, float* enorm_t, int32_t* efeat) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < edges; e += (int64_t)gridDim.x * blockDim.x) {
        const int u = indices[e], v = indices_t[e];
        enorm[e] = norm[u]; enorm_t[e] = norm[v]; efeat[e] = feat_row[u];
    }
}
// ONE pass over the rows for everything the finalisation derives from the row bounds (round 6; four launches before): hub-row lists of both orientations (atomic append; the host orders them), the fused launch's per-row source table with its
// row / edge counts (gm_batch::d_fuse2 / d_fuse2_feat, unfused_rows / unfused_edges), and the keep-flag row scale gm_batch::d_norm_c with the sign bit set on
// every row (k_centre_rows clears it on the centre rows afterwards).  Hub rows: in-degree (o = 0) / out-degree (o = 1) above `thr`.
__global__ void k_row_tables(const int32_t* indptr, const int32_t* indices, const int32_t* indptr_t, int64_t rows, const float* norm, const int32_t* feat_row,
                             int4* f2, int4* f2_feat, unsigned long long* counts, int32_t* heavy0, int32_t* heavy1, int32_t* hcnt, int cap, int thr, float* norm_c,
                             int2* first0, int2* first1, int n_first) {
    unsigned long long nr = 0, ne = 0;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int p = indptr[r], d = indptr[r + 1] - p, dt = indptr_t[r + 1] - indptr_t[r];
        // [rows: cap][degrees: cap]; the first n_first (row, degree) pairs also go to the round trip's scratch (one download for everything the host waits for)
        // (one atomic per wave and orientation: the hub rows of a wave take consecutive slots; the lists are sorted on the host anyway)
        {
            const unsigned long long m0 = __ballot(d > thr), m1 = __ballot(dt > thr), below = (1ull << (threadIdx.x & 63)) - 1ull;
            if (m0) {
                const int lead = __ffsll((long long)m0) - 1;
                int base = ((int)(threadIdx.x & 63) == lead) ? atomicAdd(hcnt, __popcll(m0)) : 0;
                base = __shfl(base, lead, 64);
                if (d > thr) { const int k = base + __popcll(m0 & below); if (k < cap) { heavy0[k] = (int32_t)r; heavy0[cap + k] = d; } if (k < n_first) first0[k] = make_int2((int)r, d); }
            }
            if (m1) {
                const int lead = __ffsll((long long)m1) - 1;
                int base = ((int)(threadIdx.x & 63) == lead) ? atomicAdd(hcnt + 1, __popcll(m1)) : 0;
                base = __shfl(base, lead, 64);
                if (dt > thr) { const int k = base + __popcll(m1 & below); if (k < cap) { heavy1[k] = (int32_t)r; heavy1[cap + k] = dt; } if (k < n_first) first1[k] = make_int2((int)r, dt); }
            }
        }
        const float nrm = norm[r];
        norm_c[r] = __uint_as_float(__float_as_uint(nrm) | 0x80000000u);
        const int self = (int)r | GM_FUSE_SELF;
        int4 t = make_int4(self, self, __float_as_int(1.f), 0), tf = t;
        if (d == 0) { t = make_int4(GM_FUSE_ZERO, GM_FUSE_ZERO, __float_as_int(1.f), 0); tf = t; }
        else if (d <= GM_FUSE_MAXDEG) {
            const int u0 = indices[p], u1 = d >= 2 ? indices[p + 1] : u0;
            const int w0 = __float_as_int(norm[u0]), w1 = d >= 2 ? __float_as_int(norm[u1]) : 0;
            t = make_int4(u0, u1, w0, w1); tf = make_int4(feat_row[u0], feat_row[u1], w0, w1);
        }
        if (d > GM_FUSE_MAXDEG) { ++nr; ne += (unsigned long long)d; }
        f2[r] = t; f2_feat[r] = tf;
    }
    // one pair of partials per WORKGROUP (wave shuffles, then the four wave partials through LDS): as atomics the two counters are a single contended
    // address each -- per thread 92k atomics took longer than the table itself, per wave the 16k of the 1.14 M-row batch still cost ~100 us
    __shared__ unsigned long long part[2][4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nr += __shfl_down(nr, off, 64); ne += __shfl_down(ne, off, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = nr; part[1][threadIdx.x >> 6] = ne; }
    __syncthreads();
    if (threadIdx.x == 0) {         // per-workgroup partials, summed by the host after the round trip (no same-address atomics at all: 2 x 2,048 of them were a third of this kernel)
        counts[2 * blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        counts[2 * blockIdx.x + 1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}
// distinct sources of the rows with more than maxdeg in-edges: mark, then count
__global__ void k_mark_sources(const int32_t* indptr, const int32_t* indices, int64_t rows, int maxdeg, uint32_t* bits) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int p = indptr[r], d = indptr[r + 1] - p;
        if (d > maxdeg) for (int e = p; e < p + d; ++e) { const int u = indices[e]; atomicOr(&bits[u >> 5], 1u << (u & 31)); }
    }
}
__global__ void k_count_bits(const uint32_t* bits, int64_t words, unsigned long long* out) {
    unsigned long long c = 0;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) c += __popc(bits[w]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
int64_t gm_batch_unfused_sources(const gm_batch* b, hipStream_t s) {
    if (b->unfused_src >= 0) return b->unfused_src;
    const int64_t fallback = std::min<int64_t>(b->unfused_edges, b->rows);
    if (b->rows <= 0 || b->unfused_edges <= 0) return b->unfused_src = 0;
    const int64_t words = (b->rows + 31) / 32;
    uint32_t* bits = nullptr; unsigned long long* cnt = nullptr; unsigned long long h = 0;
    if (gm_alloc(&bits, (size_t)words, s) != GM_OK || gm_alloc(&cnt, 1, s) != GM_OK) { gm_dev_free(bits, s); return fallback; }
    bool ok = hipMemsetAsync(bits, 0, 4 * (size_t)words, s) == hipSuccess && hipMemsetAsync(cnt, 0, 8, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_mark_sources, dim3((int)std::min<int64_t>(2048, (b->rows + 255) / 256)), dim3(256), 0, s, b->d_indptr, b->d_indices, (int64_t)b->rows, GM_FUSE_MAXDEG, bits);
        hipLaunchKernelGGL(k_count_bits, dim3((int)std::min<int64_t>(512, (words + 255) / 256)), dim3(256), 0, s, bits, words, cnt);
        ok = hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    gm_dev_free(bits, s); gm_dev_free(cnt, s);
    return ok ? (b->unfused_src = (int64_t)h) : fallback;
}
// Ordered compaction of the rows with lo <= in-degree <= hi (the window rows of a partial aggregate launch): per-block counts, a one-block scan
// of the counts, then every block writes its rows at its offset (ballot ranks: ascending row ids).
#define MID_BLOCK 1024
__device__ __forceinline__ bool mid_row(const int32_t* indptr, int64_t r, int64_t rows, int lo, int hi) {
    if (r >= rows) return false;
    const int d = indptr[r + 1] - indptr[r];
    return d >= lo && d <= hi;
}
__global__ __launch_bounds__(MID_BLOCK) void k_mid_count(const int32_t* indptr, int64_t rows, int lo, int hi, int32_t* bcnt) {
    __shared__ int wsum[MID_BLOCK / 64];
    const int64_t r = (int64_t)blockIdx.x * MID_BLOCK + threadIdx.x;
    const unsigned long long m = __ballot(mid_row(indptr, r, rows, lo, hi));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < MID_BLOCK / 64; ++k) t += wsum[k]; bcnt[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_mid_scan(int32_t* bcnt, int nb) {      // exclusive scan in place, one block
    __shared__ int part[1024];
    const int per = (nb + 1023) / 1024, a = threadIdx.x * per, b = min(nb, a + per);
    int t = 0;
    for (int k = a; k < b; ++k) t += bcnt[k];
    part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int k = 0; k < 1024; ++k) { const int v = part[k]; part[k] = run; run += v; } }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int k = a; k < b; ++k) { const int v = bcnt[k]; bcnt[k] = run; run += v; }
}
__global__ __launch_bounds__(MID_BLOCK) void k_mid_scatter(const int32_t* indptr, int64_t rows, int lo, int hi, const int32_t* boff, int32_t* list, int cap) {
    __shared__ int wsum[MID_BLOCK / 64];
    const int64_t r = (int64_t)blockIdx.x * MID_BLOCK + threadIdx.x;
    const bool f = mid_row(indptr, r, rows, lo, hi);
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int base = boff[blockIdx.x];
    for (int k = 0; k < wv; ++k) base += wsum[k];
    const int at = base + __popcll(m & ((1ull << lane) - 1ull));
    if (f && at < cap) list[at] = (int32_t)r;
}
// centre rows, their norms and in-degrees (row-sparse backward tables)
// (norm_c != NULL: also clears the keep-flag scale's sign bit on the centre rows -- after k_row_tables set it on every row)
__global__ void k_centre_rows(const int32_t* sub_off, const int32_t* centre, int nc, int n_c, const int32_t* indptr, const float* norm,
                              int32_t* crow, float* cnorm, int32_t* cdeg, float* norm_c, int32_t* centre_copy) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_c) return;
    const int row = sub_off[k / nc] + centre[k];
    centre_copy[k] = centre[k];
    crow[k] = row; cnorm[k] = norm[row]; cdeg[k] = indptr[row + 1] - indptr[row];
    if (norm_c) norm_c[row] = __uint_as_float(__float_as_uint(norm[row]) & 0x7fffffffu);
}
__global__ void k_centre_edges(const int32_t* crow, const int32_t* eoff, int n_c, const int32_t* indptr, const int32_t* indices,
                               const float* norm, int32_t* e_row, int32_t* e_par, float* e_norm) {
    const int k = blockIdx.x;
    if (k >= n_c) return;
    const int p0 = indptr[crow[k]], n = eoff[k + 1] - eoff[k], o = eoff[k];
    for (int j = threadIdx.x; j < n; j += blockDim.x) { const int u = indices[p0 + j]; e_row[o + j] = u; e_par[o + j] = k; e_norm[o + j] = norm[u]; }
}
__global__ void k_copy_add(int32_t* dst, const int32_t* src, int64_t n, int32_t add) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) dst[k] = src[k] + add;
}
__global__ void k_gather_rows(const float* feat, int64_t ld, const int32_t* feat_row, float* out, int64_t rows, int F) {
    const int64_t total = rows * F;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = k / F; const int f = (int)(k - r * F);
        out[k] = feat[(int64_t)feat_row[r] * ld + f];
    }
}

// ------------------------------------------------------------------------------------------ host
void gm_batch_mark_use(const gm_batch* b, hipStream_t st) {
    if (!b || st == b->stream) return;              // same stream: the frees are already ordered behind the consumer
    if (!b->used_ev && hipEventCreateWithFlags(&b->used_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); b->used_ev = nullptr; return; }
    if (hipEventRecord(b->used_ev, st) != hipSuccess) (void)hipGetLastError();
}

// Hub-part counters / partial rows of orientation o are about to be used by a launch on `s`: if the previous such launch went to ANOTHER
// stream, order this one behind everything queued there so far (which includes that launch).  Costs nothing while a batch stays on one stream.
int gm_batch_hub_order(const gm_batch* b, int o, hipStream_t s, int set) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const int k = set * 2 + o;
    if (b->hub_used[k] && b->hub_stream[k] != s) {
        // The remembered stream may have been destroyed by its owner since: a failed record / wait must neither drop the ordering nor leave a
        // sticky error for the next hipGetLastError() -- fall back to draining the device before the scratch is reused.
        bool ordered = false;
        if (b->hub_ev[k] || hipEventCreateWithFlags(&b->hub_ev[k], hipEventDisableTiming) == hipSuccess)
            ordered = hipEventRecord(b->hub_ev[k], b->hub_stream[k]) == hipSuccess && hipStreamWaitEvent(s, b->hub_ev[k], 0) == hipSuccess;
        if (!ordered) {
            (void)hipGetLastError();
            GM_HIP(hipDeviceSynchronize());
        }
    }
    b->hub_used[k] = true; b->hub_stream[k] = s;
    return GM_OK;
}

// Second set of hub-part arrival counters and partial rows: the part tables are copied (device to device, on `s`, behind the batch's build),
// the counters start at zero like the first set's.
int gm_batch_hub_alt(const gm_batch* cb, hipStream_t s) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    gm_batch* b = const_cast<gm_batch*>(cb);
    bool waited = false;
    for (int o = 0; o < 2; ++o) {
        if (!b->d_hub[o] || b->d_hub2[o]) continue;
        int32_t* h2 = nullptr; float* sc2 = nullptr;
        GM_TRY(gm_balloc(b, &h2, (size_t)b->hub_words[o], b->stream));
        GM_TRY(gm_balloc(b, &sc2, (size_t)b->hub_parts[o] * GM_AGG_HUB_LD, b->stream));
        if (!waited && s != b->stream) { hipEvent_t e; GM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); GM_HIP(hipEventRecord(e, b->stream)); GM_HIP(hipStreamWaitEvent(s, e, 0)); GM_HIP(hipEventDestroy(e)); waited = true; }
        GM_HIP(hipMemcpyAsync(h2, b->d_hub[o], sizeof(int32_t) * (size_t)b->hub_words[o], hipMemcpyDeviceToDevice, s));
        // (the first set's counters are zero between launches -- the last arriver resets them -- but a launch of the first set may be in flight on
        // another stream right now: zero the copy's counters explicitly)
        const size_t n_heavy = (size_t)b->n_heavy[o], parts = (size_t)b->hub_parts[o];
        GM_HIP(hipMemsetAsync(h2 + n_heavy + 1 + parts, 0, sizeof(int32_t) * n_heavy, s));
        b->d_hub2[o] = h2; b->d_hub_scratch2[o] = sc2;
    }
    if (waited) gm_batch_mark_use(b, s);
    return GM_OK;
}

int gm_balloc_bytes(gm_batch* b, void** p, size_t bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(b->slab_mu);          // (lazily built tables -- receptive-field levels, stream tables, gains -- may come from another thread than the build's)
    bytes = (bytes + 255) / 256 * 256;
    if (b->slabs.empty() || b->slabs.back().cap - b->slabs.back().used < bytes) {
        // slab size: what the big arrays of this batch will need in total when the sizes are known (rows / edges; measured on the arxiv query batch:
        // 139 MB at 1.14 M rows / 2.1 M edges), plus room for the level arrays of a two-layer receptive-field build (12 + 4 bytes per row, 8 per edge:
        // cone.hip) so that a batch is ONE block of the slab cache; else 8 MiB steps
        const size_t guess = (size_t)b->rows * 74 + (size_t)b->edges * 27 + ((size_t)2 << 20) + (size_t)b->rows * 20 + (size_t)b->edges * 8;
        gm_batch::slab sl{nullptr, 0, 0};
        GM_TRY(gm_slab_acquire(&sl.base, &sl.cap, std::max(bytes, b->slabs.empty() ? guess : std::max<size_t>(guess / 4, (size_t)8 << 20)), s));
        b->slabs.push_back(sl);
    }
    gm_batch::slab& sl = b->slabs.back();
    *p = sl.base + sl.used; sl.used += bytes;
    return GM_OK;
}

static void batch_free(gm_batch* b) {
    gm_phase_timer tm("batch-free");
    hipStream_t s = b->stream;
    for (int o = 0; o < 4; ++o) if (b->hub_ev[o]) { (void)hipEventDestroy(b->hub_ev[o]); b->hub_ev[o] = nullptr; }
    if (b->used_ev) {
        if (hipStreamWaitEvent(s, b->used_ev, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(b->used_ev); }
        (void)hipEventDestroy(b->used_ev); b->used_ev = nullptr;
    }
    tm.lap("events");
    for (auto& sl : b->slabs) {      // every array of the batch lives in these (gm_balloc)
        gm_slab_release(sl.base, sl.cap, s);
        if (gm_knob().timing) { char nm[64]; snprintf(nm, sizeof nm, "slab %zu MB (%zu used)", sl.cap >> 20, sl.used >> 20); tm.lap(nm); }
    }
    b->slabs.clear();
    tm.lap("slabs");
    for (int l = 0; l <= GM_MAX_GCN; ++l) { gm_cone_free(b->cone[l], s); b->cone[l] = nullptr; }
    tm.lap("cones");
}

// Launch tables derived from the set layout: GEMM row tiles never straddle two sets (each set has its own
// fast weights); weight-gradient chunks are sized by gm_wgrad_chunk_rows.
// Row gains of the two aggregates (gm_bound.h): only the opt-in two-piece kernels read them, so they are computed at first use (on `s`,
// ordered behind the batch's build) instead of in every batch finalisation (k_gains was the longest finalisation kernel: 0.2 ms on the 1.14 M-row
// query batch).  At least 1: an isolated row still passes its own magnitude on wherever a kernel adds a self term.
int gm_batch_gains(const gm_batch* cb, hipStream_t s) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    gm_batch* b = const_cast<gm_batch*>(cb);
    if (b->d_gain) return GM_OK;
    float* g = nullptr;
    GM_TRY(gm_balloc(b, &g, (size_t)2, b->stream));                                        // (the batch's slabs: freed with it, on its own stream)
    if (s != b->stream) { hipEvent_t e; GM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); GM_HIP(hipEventRecord(e, b->stream)); GM_HIP(hipStreamWaitEvent(s, e, 0)); GM_HIP(hipEventDestroy(e)); }
    GM_HIP(hipMemsetD32Async((hipDeviceptr_t)g, 0x3f800000, 2, s));                        // 1.0f, 1.0f
    if (b->rows > 0) {
        hipLaunchKernelGGL(k_gains, dim3((int)std::min<int64_t>(2048, (b->rows + 255) / 256)), dim3(256), 0, s, b->d_indptr, b->d_indices, b->d_indptr_t, b->d_norm,
                           (int64_t)b->rows, reinterpret_cast<unsigned*>(g));
        GM_HIP(hipGetLastError());
    }
    b->d_gain = g;
    gm_batch_mark_use(b, s);
    return GM_OK;
}

// Launch tables and derived per-batch tables.  ONE host round trip: the kernels whose results the host needs (hub-row lists, the fused launch's
// row / edge counts, the centres' in-degrees) are launched back to back, their results come back in one batch of copies into pinned memory,
// and everything the host derives from them goes up through pinned staging without waiting (gm_stager).
#define GM_HEAVY_FIRST 8192     // hub rows per orientation fetched with the first round trip (more: one more round trip)
// Hub rows come back in atomic-append order; the schedules want them ascending (and the build deterministic).  The rows are distinct ids below
// `rows`: an LSD radix sort, 11 bits a pass (two passes up to 4 M rows), carries the degrees along -- std::sort on the pairs cost 0.2 ms per
// orientation of the 1.14 M-row query batch, on the host, between the build's kernels.
static void sort_rows_with_degrees(std::vector<int32_t>& row, std::vector<int32_t>& deg, int64_t rows) {
    const size_t n = row.size();
    if (n < 2) return;
    std::vector<int32_t> row2(n), deg2(n);
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < rows) ++bits;
    int32_t* r0 = row.data(); int32_t* d0 = deg.data(); int32_t* r1 = row2.data(); int32_t* d1 = deg2.data();
    for (int shift = 0; shift < bits; shift += 11) {
        uint32_t cnt[2049] = {};
        for (size_t k = 0; k < n; ++k) ++cnt[(((uint32_t)r0[k] >> shift) & 2047u) + 1];
        for (int k = 0; k < 2048; ++k) cnt[k + 1] += cnt[k];
        for (size_t k = 0; k < n; ++k) { const uint32_t at = cnt[((uint32_t)r0[k] >> shift) & 2047u]++; r1[at] = r0[k]; d1[at] = d0[k]; }
        std::swap(r0, r1); std::swap(d0, d1);
    }
    if (r0 != row.data()) { row.swap(row2); deg.swap(deg2); }
}
// The finalisation in two halves around its host round trip, so that a caller that builds TWO batches (gm_extract_pair) queues both batches' kernels,
// waits once, and derives both batches' host-side tables while nothing is left to wait for.
struct FinalizeCtx {
    int cap = 0, first = 0, n_count_pairs = 0;
    char* scratch = nullptr; hipStream_t s = nullptr;            // device side of the round trip (finalize_launch); released by finalize_finish or on the way out
    const int32_t* h_cnt = nullptr; const int32_t* h_first[2] = {nullptr, nullptr};      // hub-row counts; the first (row, degree) pairs of both hub lists
    const unsigned long long* h_counts = nullptr; const int32_t* h_cdeg = nullptr; const int32_t* h_centre = nullptr;
    FinalizeCtx() = default;
    FinalizeCtx(const FinalizeCtx&) = delete;
    FinalizeCtx& operator=(const FinalizeCtx&) = delete;
    ~FinalizeCtx() { if (scratch) gm_dev_free(scratch, s); }
};
static int finalize_launch(gm_batch* b, hipStream_t s, gm_stager& sg, FinalizeCtx& fc);
static int finalize_finish(gm_batch* b, hipStream_t s, gm_stager& sg, FinalizeCtx& fc);
int gm_batch_finalize(gm_batch* b, hipStream_t s, gm_stager& sg) {
    FinalizeCtx fc;
    GM_TRY(finalize_launch(b, s, sg, fc));
    GM_HIP(hipStreamSynchronize(s));
    return finalize_finish(b, s, sg, fc);
}
// several small host tables into ONE region of the batch's slabs (batch_free releases nothing else) with ONE copy through pinned staging; each part
// starts on a 256-byte boundary.  (One hipMemcpyAsync per table before: ~20 of the ~43 copies of a meta-batch build.)
struct TabPart { int32_t** d; const std::vector<int32_t>* v; };
static int upload_tables(gm_batch* b, gm_stager& sg, hipStream_t s, std::initializer_list<TabPart> parts) {
    auto pad = [](size_t n) { return (std::max<size_t>(n, 1) + 63) / 64 * 64; };
    size_t tot = 0;
    for (const TabPart& p : parts) tot += pad(p.v->size());
    int32_t* base = nullptr;
    GM_TRY(gm_balloc(b, &base, tot, s));
    std::vector<int32_t> h(tot, 0);
    size_t o = 0;
    for (const TabPart& p : parts) { std::copy(p.v->begin(), p.v->end(), h.begin() + o); *p.d = base + o; o += pad(p.v->size()); }
    return sg.upload(base, h);
}
static int finalize_launch(gm_batch* b, hipStream_t s, gm_stager& sg, FinalizeCtx& fc) {
    gm_phase_timer tm("finalize-launch");
    std::vector<int32_t> sub_set(b->subs), tiles, chunks, set_chunk_off(b->sets + 1, 0);
    for (int t = 0; t < b->sets; ++t)
        for (int k = b->h_set_sub_off[t]; k < b->h_set_sub_off[t + 1]; ++k) sub_set[k] = t;
    const int64_t cr = gm_wgrad_chunk_rows(b->h_set_row_off);
    for (int t = 0; t < b->sets; ++t) {
        const int r0 = b->h_set_row_off[t], r1 = b->h_set_row_off[t + 1];
        for (int r = r0; r < r1; r += GM_GEMM_BM) { tiles.push_back(t); tiles.push_back(r); tiles.push_back(std::min(GM_GEMM_BM, r1 - r)); }
        for (int r = r0; r < r1; r += (int)cr) { chunks.push_back(t); chunks.push_back(r); chunks.push_back(std::min<int>((int)cr, r1 - r)); }
        set_chunk_off[t + 1] = (int32_t)(chunks.size() / 3);
    }
    b->n_tiles = (int32_t)(tiles.size() / 3); b->n_chunks = (int32_t)(chunks.size() / 3);
    GM_TRY(upload_tables(b, sg, s, {{&b->d_sub_set, &sub_set}, {&b->d_tiles, &tiles}, {&b->d_chunks, &chunks}, {&b->d_set_chunk_off, &set_chunk_off}}));
    tm.lap("tables");
    // ---- device side, nothing here waits for the host: hub-row lists of both orientations, per-edge tables, the fused launch's row table +
    // counts, centre rows with their in-degrees
    b->heavy_deg = gm_heavy_deg_for(b->rows, b->edges);
    const int cap = (int)(b->edges / b->heavy_deg + 1);
    // scratch of the round trip in ONE allocation, ONE memset, ONE download (eight copies before):
    // ints [4,6) hub-row counts | [8, 8 + n_c) centre in-degrees | n_c local centre ids | 2 x first (row, degree) pairs of the hub lists | k_row_tables' per-workgroup {rows, edges} partials (u64 pairs)
    const int nc = b->centres; b->n_c = b->subs * nc;
    const int first = std::min(cap, GM_HEAVY_FIRST);
    const int rt_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (b->rows + 255) / 256));        // k_row_tables' grid, one {rows, edges} partial pair per workgroup (512 .. 8,192 workgroups: the same 91 us on the query batch)
    const size_t o_cdeg = 8, o_centre = o_cdeg + b->n_c, o_first = (o_centre + b->n_c + 1) / 2 * 2, o_part = o_first + 4 * (size_t)first, scr_ints = o_part + 4 * (size_t)rt_blocks;
    char* scratch = nullptr;
    GM_TRY(gm_dev_alloc((void**)&scratch, 4 * scr_ints, s));
    fc.scratch = scratch; fc.s = s;
    GM_HIP(hipMemsetAsync(scratch, 0, 32, s));
    unsigned long long* d_counts = (unsigned long long*)((int32_t*)scratch + o_part);
    int32_t* d_cnt = (int32_t*)scratch + 4;
    int32_t* d_cdeg = (int32_t*)scratch + o_cdeg;
    int2* d_first[2] = {(int2*)((int32_t*)scratch + o_first), (int2*)((int32_t*)scratch + o_first) + first};
    for (int o = 0; o < 2; ++o) GM_TRY(gm_balloc(b, &b->d_heavy[o], 2 * (size_t)cap, s));
    const int edge_tables = gm_knob().agg_edge_tables;
    if (b->edges > 0 && edge_tables) {
        GM_TRY(gm_balloc(b, &b->d_enorm[0], (size_t)b->edges, s)); GM_TRY(gm_balloc(b, &b->d_enorm[1], (size_t)b->edges, s)); GM_TRY(gm_balloc(b, &b->d_efeat, (size_t)b->edges, s));
        hipLaunchKernelGGL(k_edge_tables, dim3((int)std::min<int64_t>(4096, (b->edges + 255) / 256)), dim3(256), 0, s, b->d_indices, b->d_indices_t, (int64_t)b->edges,
                           b->d_norm, b->d_feat_row, b->d_enorm[0], b->d_enorm[1], b->d_efeat);
    }
    GM_TRY(gm_balloc(b, &b->d_norm_c, b->rows, s));
    if (b->rows > 0) {
        int4 *f0 = nullptr, *ff = nullptr;
        GM_TRY(gm_balloc(b, &f0, (size_t)b->rows, s)); GM_TRY(gm_balloc(b, &ff, (size_t)b->rows, s));
        b->d_fuse2 = f0; b->d_fuse2_feat = ff;
        hipLaunchKernelGGL(k_row_tables, dim3(rt_blocks), dim3(256), 0, s, b->d_indptr, b->d_indices, b->d_indptr_t, (int64_t)b->rows,
                           b->d_norm, b->d_feat_row, f0, ff, d_counts, b->d_heavy[0], b->d_heavy[1], d_cnt, cap, b->heavy_deg, b->d_norm_c, d_first[0], d_first[1], first);
    }
    GM_TRY(gm_balloc(b, &b->d_crow, b->n_c, s)); GM_TRY(gm_balloc(b, &b->d_cnorm, b->n_c, s));
    hipLaunchKernelGGL(k_centre_rows, dim3((b->n_c + 255) / 256), dim3(256), 0, s, b->d_sub_off, b->d_centre, nc, b->n_c, b->d_indptr, b->d_norm,
                       b->d_crow, b->d_cnorm, d_cdeg, b->d_norm_c, (int32_t*)scratch + o_centre);
    GM_HIP(hipGetLastError());
    // ---- the one round trip
    const int32_t* h_scr = sg.download((const int32_t*)scratch, scr_ints);
    GM_REQUIRE(h_scr, GM_ENOMEM, "finalize: pinned staging failed");
    tm.lap("launches");
    fc.cap = cap; fc.first = first;
    fc.h_cnt = h_scr + 4; fc.h_counts = b->rows > 0 ? (const unsigned long long*)(h_scr + o_part) : nullptr; fc.n_count_pairs = rt_blocks; fc.h_cdeg = h_scr + o_cdeg; fc.h_centre = h_scr + o_centre;
    fc.h_first[0] = h_scr + o_first; fc.h_first[1] = h_scr + o_first + 2 * (size_t)first;
    return GM_OK;
}
static int finalize_finish(gm_batch* b, hipStream_t s, gm_stager& sg, FinalizeCtx& fc) {      // (the stream has passed finalize_launch's downloads)
    gm_phase_timer tm("finalize-finish");
    const int cap = fc.cap, first = fc.first, nc = b->centres;

    const int32_t* h_cnt = fc.h_cnt; const unsigned long long* h_counts = fc.h_counts; const int32_t* h_cdeg = fc.h_cdeg;
    gm_dev_free(fc.scratch, s); fc.scratch = nullptr;
    if (h_counts) {
        unsigned long long nr = 0, ne = 0;
        for (int k = 0; k < fc.n_count_pairs; ++k) { nr += h_counts[2 * k]; ne += h_counts[2 * k + 1]; }
        b->unfused_rows = (int64_t)nr; b->unfused_edges = (int64_t)ne;
    }
    b->h_centre.assign(fc.h_centre, fc.h_centre + b->n_c);
    b->sched_win = gm_agg_window(b->rows, b->edges);
    std::vector<int32_t> heavy0, tab0;                       // forward orientation: sorted hub rows and their part table (for the list schedule below)
    for (int o = 0; o < 2; ++o) {
        b->n_heavy[o] = std::min(h_cnt[o], cap);
        if (b->n_heavy[o] > 0) {         // deterministic order (atomic append order is not)
            const size_t nh = b->n_heavy[o];
            std::vector<int32_t> h(nh), hd(nh);
            if ((int)nh <= first) { for (size_t k = 0; k < nh; ++k) { h[k] = fc.h_first[o][2 * k]; hd[k] = fc.h_first[o][2 * k + 1]; } }
            else {                       // more hub rows than the first fetch carried: one more round trip for this orientation
                const int32_t* a = sg.download(b->d_heavy[o], nh); const int32_t* d = sg.download(b->d_heavy[o] + cap, nh);
                GM_REQUIRE(a && d, GM_ENOMEM, "finalize: pinned staging failed");
                GM_HIP(hipStreamSynchronize(s));
                std::copy(a, a + nh, h.begin()); std::copy(d, d + nh, hd.begin());
            }
            sort_rows_with_degrees(h, hd, b->rows);
            if (nh > 1) GM_TRY(sg.upload(b->d_heavy[o], h));
            tm.lap("hub-sort");
            gm_agg_sched sc;
            GM_TRY(gm_agg_schedule(b, b->rows, b->sched_win, h.data(), hd.data(), b->n_heavy[o], &sc, s, &sg));
            tm.lap("schedule");
            if (o == 0) { heavy0 = h; tab0 = sc.tab; }
            {   // stream tables: at first use (gm_agg_stream_args), from these host copies
                gm_batch::stream_pending& sp = b->spend[o];
                sp.pending = true; sp.has_tab = sc.d_hub != nullptr; sp.n_parts = sc.d_hub ? sc.parts : (int)nh;
                sp.hubs = h; sp.deg = hd; if (sp.has_tab) sp.tab = sc.tab;
            }
            b->d_sched[o] = sc.d_sched; b->sched_len[o] = sc.len; b->d_hub[o] = sc.d_hub; b->d_hub_scratch[o] = sc.d_hub_scratch; b->hub_part[o] = sc.hub_part; b->hub_words[o] = sc.hub_words; b->hub_parts[o] = sc.parts;
        } else b->spend[o].pending = true;                       // (no hub rows)
    }
    // ---- the window rows of the fused passes' partial aggregate launch as a compact ascending list + its block schedule (no host wait: the
    // list's length follows from counts the round trip above already brought: rows with more than GM_FUSE_MAXDEG in-edges minus the hub rows)
    if (b->d_fuse2 && gm_knob().agg_mid_list) {
        const int64_t n_mid = b->unfused_rows - (int64_t)b->n_heavy[0];
        if (n_mid > 0 && n_mid < b->rows && h_cnt[0] <= cap) {
            const int nb = (int)((b->rows + MID_BLOCK - 1) / MID_BLOCK);
            int32_t* d_bcnt = nullptr;
            GM_TRY(gm_alloc(&d_bcnt, (size_t)nb, s));
            GM_TRY(gm_balloc(b, &b->d_mid, (size_t)n_mid, s));
            hipLaunchKernelGGL(k_mid_count, dim3(nb), dim3(MID_BLOCK), 0, s, b->d_indptr, (int64_t)b->rows, GM_FUSE_MAXDEG + 1, b->heavy_deg, d_bcnt);
            hipLaunchKernelGGL(k_mid_scan, dim3(1), dim3(1024), 0, s, d_bcnt, nb);
            hipLaunchKernelGGL(k_mid_scatter, dim3(nb), dim3(MID_BLOCK), 0, s, b->d_indptr, (int64_t)b->rows, GM_FUSE_MAXDEG + 1, b->heavy_deg, d_bcnt, b->d_mid, (int)n_mid);
            GM_HIP(hipGetLastError());
            gm_dev_free(d_bcnt, s);
            b->n_mid = (int32_t)n_mid;
            // window size over the list: enough waves to fill the chip, at least two rows per wave (two rows in flight per lane group)
            int win = 64;
            while (win > 2 && n_mid / win < 16384) win >>= 1;
            if (gm_knob().agg_mid_win > 0) win = gm_knob().agg_mid_win;
            b->mid_win = win;
            if (b->n_heavy[0] > 0 && b->d_sched[0]) {
                // hub parts ride in the same launch: placed after the list block nearest to the hub row's position (row id scaled to the list)
                std::vector<int32_t> pos(heavy0.size());
                for (size_t k = 0; k < heavy0.size(); ++k) pos[k] = (int32_t)std::min<int64_t>(n_mid - 1, (int64_t)heavy0[k] * n_mid / b->rows);
                GM_TRY(gm_agg_schedule_flat(b, n_mid, win, pos.data(), (int)heavy0.size(), tab0, &b->d_sched_mid, &b->sched_len_mid, s, &sg));
            }
        }
    }
    tm.lap("mid-list");
    // ---- compact lists for the row-sparse backward: centre rows and the in-edges of centres
    std::vector<int32_t> eoff(b->n_c + 1, 0);
    for (int k = 0; k < b->n_c; ++k) eoff[k + 1] = eoff[k] + h_cdeg[k];
    b->n_e1 = eoff[b->n_c];
    int32_t* d_eoff = nullptr;
    GM_TRY(gm_alloc(&d_eoff, eoff.size(), s));
    GM_TRY(sg.upload(d_eoff, eoff));
    GM_TRY(gm_balloc(b, &b->d_e1_row, b->n_e1, s)); GM_TRY(gm_balloc(b, &b->d_e1_par, b->n_e1, s)); GM_TRY(gm_balloc(b, &b->d_e1_norm, b->n_e1, s));
    hipLaunchKernelGGL(k_centre_edges, dim3(b->n_c), dim3(64), 0, s, b->d_crow, d_eoff, b->n_c, b->d_indptr, b->d_indices, b->d_norm,
                       b->d_e1_row, b->d_e1_par, b->d_e1_norm);
    GM_HIP(hipGetLastError());
    std::vector<int32_t> ct, cc, ccoff(b->sets + 1, 0), ec, ecoff(b->sets + 1, 0), c_set_off(b->sets + 1), e_set_off(b->sets + 1);
    for (int t = 0; t <= b->sets; ++t) { c_set_off[t] = b->h_set_sub_off[t] * nc; e_set_off[t] = eoff[b->h_set_sub_off[t] * nc]; }
    const int ccr = gm_wgrad_chunk_rows(c_set_off), ecr = gm_wgrad_chunk_rows(e_set_off);
    for (int t = 0; t < b->sets; ++t) {
        const int k0 = b->h_set_sub_off[t] * nc, k1 = b->h_set_sub_off[t + 1] * nc;
        for (int k = k0; k < k1; k += GM_GEMM_BM) { ct.push_back(t); ct.push_back(k); ct.push_back(std::min(GM_GEMM_BM, k1 - k)); }
        for (int k = k0; k < k1; k += ccr) { cc.push_back(t); cc.push_back(k); cc.push_back(std::min(ccr, k1 - k)); }
        ccoff[t + 1] = (int32_t)(cc.size() / 3);
        for (int q = eoff[k0]; q < eoff[k1]; q += ecr) { ec.push_back(t); ec.push_back(q); ec.push_back(std::min(ecr, eoff[k1] - q)); }
        ecoff[t + 1] = (int32_t)(ec.size() / 3);
    }
    b->n_c_tiles = (int32_t)(ct.size() / 3); b->n_c_chunks = (int32_t)(cc.size() / 3); b->n_e1_chunks = (int32_t)(ec.size() / 3);
    GM_TRY(upload_tables(b, sg, s, {{&b->d_c_tiles, &ct}, {&b->d_c_chunks, &cc}, {&b->d_c_set_chunk_off, &ccoff}, {&b->d_e1_chunks, &ec}, {&b->d_e1_set_chunk_off, &ecoff}}));
    gm_dev_free(d_eoff, s);
    return GM_OK;
}

extern "C" void gm_batch_destroy(gm_batch_t* b) {
    if (!b) return;
    batch_free(b);
    delete b;
}

static int batch_alloc(gm_batch* b, hipStream_t s) {
    GM_TRY(gm_balloc(b, &b->d_sub_off, b->subs + 1, s)); GM_TRY(gm_balloc(b, &b->d_set_sub_off, b->sets + 1, s));
    GM_TRY(gm_balloc(b, &b->d_set_row_off, b->sets + 1, s)); GM_TRY(gm_balloc(b, &b->d_graph, b->subs, s));
    GM_TRY(gm_balloc(b, &b->d_parent, b->rows, s)); GM_TRY(gm_balloc(b, &b->d_feat_row, b->rows, s));
    GM_TRY(gm_balloc(b, &b->d_indptr, b->rows + 1, s)); GM_TRY(gm_balloc(b, &b->d_indices, b->edges, s));
    GM_TRY(gm_balloc(b, &b->d_indptr_t, b->rows + 1, s)); GM_TRY(gm_balloc(b, &b->d_indices_t, b->edges, s));
    GM_TRY(gm_balloc(b, &b->d_centre, (size_t)b->subs * b->centres, s)); GM_TRY(gm_balloc(b, &b->d_norm, b->rows, s));
    return GM_OK;
}

static int upload_small(gm_batch* b, gm_stager& sg) {
    GM_TRY(sg.upload(b->d_sub_off, b->h_sub_off)); GM_TRY(sg.upload(b->d_set_sub_off, b->h_set_sub_off));
    GM_TRY(sg.upload(b->d_set_row_off, b->h_set_row_off)); GM_TRY(sg.upload(b->d_graph, b->h_graph));
    return GM_OK;
}

// One build of ONE batch (n_parts = 1) or of the two batches of a meta-batch together (n_parts = 2: seeds = [part 0 | part 1]; gm_extract_pair): the
// node-set kernel runs over all subgraphs in one launch, so does the fill kernel (k_fill serves two batches), the two finalisations queue their kernels
// back to back and share one host round trip.
struct ExPart { const int32_t* set_offsets; int32_t n_sets; int32_t n_seeds; };
static int extract_impl(const gm_store_t* store, const gm_seed_t* seeds, int n_parts, const ExPart* parts, int32_t h, int32_t sample_nodes, uint64_t rng_seed,
                        int32_t link, const int32_t* nodes_flat, const int64_t* nodes_off, void* stream, gm_batch_t** outs) {
    GM_REQUIRE(outs && (n_parts == 1 || n_parts == 2), GM_EINVAL, "extract: out is NULL");
    for (int p = 0; p < n_parts; ++p) outs[p] = nullptr;
    gm_phase_timer tm("extract");
    GM_REQUIRE(store && seeds, GM_EINVAL, "extract: bad arguments");
    int32_t n_seeds = 0;
    for (int p = 0; p < n_parts; ++p) {
        const ExPart& q = parts[p];
        GM_REQUIRE(q.set_offsets && q.n_seeds >= 1 && q.n_sets >= 1, GM_EINVAL, "extract: bad arguments");
        GM_REQUIRE(q.set_offsets[0] == 0 && q.set_offsets[q.n_sets] == q.n_seeds, GM_EINVAL, "extract: set_offsets must span [0,n_seeds]");
        n_seeds += q.n_seeds;
    }
    const int32_t split = parts[0].n_seeds;                  // seeds [0, split): part 0
    const bool given = nodes_flat != nullptr;
    GM_REQUIRE(!given || n_parts == 1, GM_EINVAL, "extract: node lists are given per batch");
    if (!given) {
        GM_REQUIRE(link || (h >= 1 && h <= 3), GM_EINVAL, "extract: h=%d unsupported (the reference defines h in {1,2,3}, sdp.py:300-311)", h);
        GM_REQUIRE(sample_nodes >= 1, GM_EINVAL, "extract: sample_nodes must be >= 1");
    }
    int64_t cap = 1;
    for (int k = 0; k < n_seeds; ++k) {
        const gm_seed_t& sd = seeds[k];
        GM_REQUIRE(sd.graph >= 0 && sd.graph < store->n_graphs, GM_EINVAL, "extract: seed %d: graph %d out of range", k, sd.graph);
        const int64_t n = store->node_off[sd.graph + 1] - store->node_off[sd.graph];
        GM_REQUIRE(sd.i >= 0 && sd.i < n, GM_EINVAL, "extract: seed %d: node %d out of range", k, sd.i);
        GM_REQUIRE(!link || (sd.j >= 0 && sd.j < n), GM_EINVAL, "extract: seed %d: second node %d out of range", k, sd.j);
        if (given) {
            const int64_t a = nodes_off[k], b = nodes_off[k + 1];
            GM_REQUIRE(b > a, GM_EINVAL, "from_nodes: subgraph %d is empty", k);
            bool has_i = false, has_j = !link;
            for (int64_t q = a; q < b; ++q) {
                GM_REQUIRE(nodes_flat[q] >= 0 && nodes_flat[q] < n, GM_EINVAL, "from_nodes: node id out of range in subgraph %d", k);
                GM_REQUIRE(q == a || nodes_flat[q] > nodes_flat[q - 1], GM_EINVAL, "from_nodes: subgraph %d not strictly ascending", k);
                has_i |= nodes_flat[q] == sd.i; has_j |= nodes_flat[q] == sd.j;
            }
            GM_REQUIRE(has_i && has_j, GM_EINVAL, "from_nodes: subgraph %d does not contain its centre(s)", k);
            cap = std::max<int64_t>(cap, b - a);
        }
    }
    if (!given) cap = std::min<int64_t>(store->max_nodes, (int64_t)sample_nodes + 2);
    const int Wmax = (int)((store->max_nodes + 31) >> 5);
    size_t lds_a = sizeof(uint32_t) * (2 * (size_t)Wmax + EX_BLOCK + 256 + 16);
    const int force_global = gm_knob().extract_global_bitmap;
    // parent graphs beyond ~650k nodes do not fit the LDS bitmap pair: fall back to a per-workgroup slab in HBM
    const bool gpath = force_global || lds_a > 160 * 1024;
    if (gpath) lds_a = sizeof(uint32_t) * (EX_BLOCK + 256 + 16);
    // 16-bit prefix words wherever a subgraph stays below 65,536 nodes (GM_EXTRACT_PREF16=0: 32-bit as before); the BFS keeps its `expanded` bitmap from three hops on
    const bool p16 = !gpath && cap < 65536 && gm_knob().extract_pref16;
    const bool needx = !given && !link && h >= 3;
    const size_t Wp = p16 ? ((size_t)Wmax + 1) / 2 : (size_t)Wmax;              // words of the prefix region
    if (!gpath) lds_a = sizeof(uint32_t) * ((size_t)Wmax + ((p16 && !needx) ? Wp : (size_t)Wmax) + EX_BLOCK + 256 + 16);
    hipStream_t st = (hipStream_t)stream;
    GM_TRY(gm_func_full_lds((const void*)k_nodes<false>));
    GM_TRY(gm_func_full_lds((const void*)k_nodes<false, true, true>));
    GM_TRY(gm_func_full_lds((const void*)k_nodes<false, true, false>));
    GM_TRY(gm_func_full_lds((const void*)k_fill<false>));
    GM_TRY(gm_func_full_lds((const void*)k_fill<false, true>));
    ExStore S{store->d_node_off, store->d_in_ptr, store->d_in_idx, store->d_out_ptr, store->d_out_idx, store->symmetric ? 1 : 0};

    gm_seed_t* d_seeds = nullptr; int32_t *d_nodes = nullptr, *d_degi = nullptr, *d_dego = nullptr, *d_nsub = nullptr, *d_esub = nullptr;
    int32_t* d_given = nullptr; int64_t* d_given_off = nullptr;
    int32_t* d_eoff = nullptr;
    uint32_t* d_gbits = nullptr;
    gm_batch* bs[2] = {new gm_batch(), n_parts == 2 ? new gm_batch() : nullptr};
    int rc = GM_OK;
    auto cleanup = [&]() {
        gm_dev_free(d_gbits, st);
        gm_dev_free(d_seeds, st); gm_dev_free(d_nodes, st); gm_dev_free(d_degi, st); gm_dev_free(d_dego, st);
        gm_dev_free(d_nsub, st); gm_dev_free(d_esub, st); gm_dev_free(d_given, st); gm_dev_free(d_given_off, st); gm_dev_free(d_eoff, st);
    };
    auto drop = [&]() { cleanup(); for (int p = 0; p < n_parts; ++p) { batch_free(bs[p]); delete bs[p]; } };
#define EX_TRY(x) do { rc = (x); if (rc != GM_OK) { drop(); return rc; } } while (0)
#define EX_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gm_set_error("%s: %s", #x, hipGetErrorString(e_)); drop(); return GM_EHIP; } } while (0)
    tm.lap("validate");
    gm_stager sg(st);                          // pinned staging: the build makes TWO host round trips (subgraph sizes, finalisation)
    EX_TRY(gm_alloc(&d_seeds, n_seeds, st));
    EX_TRY(gm_alloc(&d_nodes, (size_t)n_seeds * cap, st)); EX_TRY(gm_alloc(&d_degi, (size_t)n_seeds * cap, st));
    EX_TRY(gm_alloc(&d_dego, (size_t)n_seeds * cap, st));
    EX_TRY(gm_alloc(&d_nsub, n_seeds, st)); EX_TRY(gm_alloc(&d_esub, n_seeds, st));
    EX_TRY(sg.upload(d_seeds, seeds, sizeof(gm_seed_t) * n_seeds));
    if (given) {
        const int64_t tot = nodes_off[n_seeds];
        EX_TRY(gm_alloc(&d_given, tot, st)); EX_TRY(gm_alloc(&d_given_off, n_seeds + 1, st));
        EX_TRY(sg.upload(d_given, nodes_flat, sizeof(int32_t) * tot));
        EX_TRY(sg.upload(d_given_off, nodes_off, sizeof(int64_t) * (n_seeds + 1)));
    }
    gm_prof_begin(GM_PROF_EX_NODES, st, n_seeds);
    if (gpath) {
        EX_TRY(gm_alloc(&d_gbits, (size_t)n_seeds * 2 * Wmax, st));
        hipLaunchKernelGGL(k_nodes<true>, dim3(n_seeds), dim3(EX_BLOCK), lds_a, st, S, d_seeds, n_seeds, h, sample_nodes, rng_seed, link ? 1 : 0,
                           d_given, d_given_off, (int)cap, d_nodes, d_degi, d_dego, d_nsub, d_esub, Wmax, d_gbits);
    } else if (p16 && !needx) {
        hipLaunchKernelGGL((k_nodes<false, true, false>), dim3(n_seeds), dim3(EX_BLOCK), lds_a, st, S, d_seeds, n_seeds, h, sample_nodes, rng_seed, link ? 1 : 0,
                           d_given, d_given_off, (int)cap, d_nodes, d_degi, d_dego, d_nsub, d_esub, Wmax, (uint32_t*)nullptr);
    } else if (p16) {
        hipLaunchKernelGGL((k_nodes<false, true, true>), dim3(n_seeds), dim3(EX_BLOCK), lds_a, st, S, d_seeds, n_seeds, h, sample_nodes, rng_seed, link ? 1 : 0,
                           d_given, d_given_off, (int)cap, d_nodes, d_degi, d_dego, d_nsub, d_esub, Wmax, (uint32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_nodes<false>, dim3(n_seeds), dim3(EX_BLOCK), lds_a, st, S, d_seeds, n_seeds, h, sample_nodes, rng_seed, link ? 1 : 0,
                           d_given, d_given_off, (int)cap, d_nodes, d_degi, d_dego, d_nsub, d_esub, Wmax, (uint32_t*)nullptr);
    }
    gm_prof_end(GM_PROF_EX_NODES, st);
    EX_HIP(hipGetLastError());
    const int32_t* nsub = sg.download(d_nsub, (size_t)n_seeds); const int32_t* esub = sg.download(d_esub, (size_t)n_seeds);
    if (!nsub || !esub) { gm_set_error("extract: pinned staging failed"); drop(); return GM_ENOMEM; }
    EX_HIP(hipStreamSynchronize(st));

    tm.lap("k_nodes+sizes");
    // per batch: host prefix sums, device arrays; the edge offsets of both batches share one upload ([part 0: n0 + 1 | part 1: n1 + 1])
    std::vector<int32_t> eoff;
    eoff.reserve(2 * (size_t)n_seeds + n_parts);          // (+ k_fill's workgroup order below: no reallocation under eoff_p)
    eoff.assign((size_t)n_seeds + n_parts, 0);
    int32_t* eoff_p[2] = {eoff.data(), eoff.data() + split + 1};
    FillOut fo[2] = {};
    for (int p = 0, k0 = 0; p < n_parts; k0 += parts[p].n_seeds, ++p) {
        gm_batch* b = bs[p]; const ExPart& q = parts[p];
        b->store = store; b->subs = q.n_seeds; b->sets = q.n_sets; b->centres = link ? 2 : 1; b->stream = st;
        b->h_sub_off.assign(q.n_seeds + 1, 0); b->h_graph.resize(q.n_seeds);
        int64_t rows = 0, edges = 0;
        for (int k = 0; k < q.n_seeds; ++k) {
            const int gk = k0 + k;
            if (nsub[gk] <= 0 || esub[gk] < 0) {
                gm_set_error("extract: subgraph %d failed on device (nodes=%d, edges=%d, cap=%lld)", gk, nsub[gk], esub[gk], (long long)cap);
                drop(); return GM_ERANGE;
            }
            rows += nsub[gk]; edges += esub[gk];
            if (rows > INT32_MAX - 2 || edges > INT32_MAX - 2) { gm_set_error("extract: batch exceeds 2^31 rows/edges; split the meta-batch"); drop(); return GM_ERANGE; }
            b->h_sub_off[k + 1] = (int32_t)rows; eoff_p[p][k + 1] = (int32_t)edges; b->h_graph[k] = seeds[gk].graph;
        }
        b->rows = rows; b->edges = edges;
        b->h_set_sub_off.assign(q.set_offsets, q.set_offsets + q.n_sets + 1);
        b->h_set_row_off.resize(q.n_sets + 1);
        for (int s = 0; s <= q.n_sets; ++s) b->h_set_row_off[s] = b->h_sub_off[q.set_offsets[s]];
        EX_TRY(batch_alloc(b, st));
        EX_TRY(upload_small(b, sg));
    }
    // k_fill's workgroup order rides in the same upload: subgraphs by edge count, largest first (64 linear buckets: coarse is enough, and O(n))
    const size_t o_order = eoff.size();
    const bool lpt = n_seeds >= 512;                      // (more subgraphs than workgroup slots in half a round: 0.156 -> 0.145 ms at the arxiv shape, profiles/r06_experiments_not_shipped.txt H)
    if (lpt) {
        int emax = 1;
        for (int k = 0; k < n_seeds; ++k) emax = std::max(emax, esub[k]);
        int cnt[65] = {0};
        auto bucket = [&](int e) { return 63 - (int)((int64_t)e * 63 / emax); };             // 0 = largest
        for (int k = 0; k < n_seeds; ++k) ++cnt[bucket(esub[k]) + 1];
        for (int q = 0; q < 64; ++q) cnt[q + 1] += cnt[q];
        eoff.resize(o_order + n_seeds);
        for (int k = 0; k < n_seeds; ++k) eoff[o_order + cnt[bucket(esub[k])]++] = k;
    }
    EX_TRY(gm_alloc(&d_eoff, eoff.size(), st));
    EX_TRY(sg.upload(d_eoff, eoff));
    const int32_t* d_order = lpt ? d_eoff + o_order : nullptr;
    for (int p = 0; p < n_parts; ++p) {
        gm_batch* b = bs[p];
        fo[p] = FillOut{b->d_sub_off, d_eoff + (p ? split + 1 : 0), b->d_parent, b->d_feat_row, b->d_norm, b->d_indptr, b->d_indices, b->d_indptr_t, b->d_indices_t, b->d_centre};
    }
    if (n_parts == 1) fo[1] = fo[0];
    gm_prof_begin(GM_PROF_EX_FILL, st, n_seeds);
    if (gpath) {
        hipLaunchKernelGGL(k_fill<true>, dim3(n_seeds), dim3(EX_BLOCK), sizeof(uint32_t) * (EX_BLOCK + 16), st, S, d_seeds, n_seeds, link ? 1 : 0, (int)cap,
                           d_nodes, d_degi, d_dego, fo[0], fo[1], (int)split, Wmax, d_gbits, d_order);
    } else {
        const size_t lds_b = sizeof(uint32_t) * ((size_t)Wmax + Wp + EX_BLOCK + 16);
        if (p16) hipLaunchKernelGGL((k_fill<false, true>), dim3(n_seeds), dim3(EX_BLOCK), lds_b, st, S, d_seeds, n_seeds, link ? 1 : 0, (int)cap, d_nodes, d_degi, d_dego,
                                    fo[0], fo[1], (int)split, Wmax, (uint32_t*)nullptr, d_order);
        else hipLaunchKernelGGL(k_fill<false>, dim3(n_seeds), dim3(EX_BLOCK), lds_b, st, S, d_seeds, n_seeds, link ? 1 : 0, (int)cap, d_nodes, d_degi, d_dego,
                                fo[0], fo[1], (int)split, Wmax, (uint32_t*)nullptr, d_order);
    }
    gm_prof_end(GM_PROF_EX_FILL, st);
    EX_HIP(hipGetLastError());
    tm.lap("alloc+k_fill");
    // finalisation: both batches' kernels and downloads queued, ONE wait, then the host halves
    gm_prof_begin(GM_PROF_EX_FINAL, st, 1);
    FinalizeCtx fc[2];
    for (int p = 0; p < n_parts; ++p) EX_TRY(finalize_launch(bs[p], st, sg, fc[p]));
    EX_HIP(hipStreamSynchronize(st));
    tm.lap("finalize-wait");
    for (int p = 0; p < n_parts; ++p) EX_TRY(finalize_finish(bs[p], st, sg, fc[p]));
    gm_prof_end(GM_PROF_EX_FINAL, st);
    tm.lap("finalize");
    cleanup();
#undef EX_TRY
#undef EX_HIP
    for (int p = 0; p < n_parts; ++p) outs[p] = bs[p];
    return GM_OK;
}

extern "C" int gm_extract(const gm_store_t* store, const gm_seed_t* seeds, int32_t n_seeds, const int32_t* set_offsets, int32_t n_sets,
                          int32_t h, int32_t sample_nodes, uint64_t rng_seed, int32_t link_pred, void* stream, gm_batch_t** out) {
    GM_REQUIRE(out, GM_EINVAL, "extract: out is NULL");
    const ExPart part{set_offsets, n_sets, n_seeds};
    return extract_impl(store, seeds, 1, &part, h, sample_nodes, rng_seed, link_pred, nullptr, nullptr, stream, out);
}

extern "C" int gm_extract_pair(const gm_store_t* store, const gm_seed_t* seeds_a, int32_t n_seeds_a, const int32_t* set_offsets_a, int32_t n_sets_a,
                               const gm_seed_t* seeds_b, int32_t n_seeds_b, const int32_t* set_offsets_b, int32_t n_sets_b,
                               int32_t h, int32_t sample_nodes, uint64_t rng_seed, int32_t link_pred, void* stream, gm_batch_t** out_a, gm_batch_t** out_b) {
    GM_REQUIRE(out_a && out_b && seeds_a && seeds_b && n_seeds_a >= 1 && n_seeds_b >= 1, GM_EINVAL, "extract_pair: bad arguments");
    *out_a = nullptr; *out_b = nullptr;
    std::vector<gm_seed_t> all((size_t)n_seeds_a + n_seeds_b);
    std::copy(seeds_a, seeds_a + n_seeds_a, all.begin()); std::copy(seeds_b, seeds_b + n_seeds_b, all.begin() + n_seeds_a);
    const ExPart parts[2] = {{set_offsets_a, n_sets_a, n_seeds_a}, {set_offsets_b, n_sets_b, n_seeds_b}};
    gm_batch_t* outs[2] = {nullptr, nullptr};
    const int rc = extract_impl(store, all.data(), 2, parts, h, sample_nodes, rng_seed, link_pred, nullptr, nullptr, stream, outs);
    *out_a = outs[0]; *out_b = outs[1];
    return rc;
}

extern "C" int gm_batch_from_nodes(const gm_store_t* store, const gm_seed_t* seeds, int32_t n_seeds, const int32_t* set_offsets,
                                   int32_t n_sets, const int32_t* nodes_flat, const int64_t* nodes_off, int32_t link_pred, void* stream,
                                   gm_batch_t** out) {
    GM_REQUIRE(out && nodes_flat && nodes_off, GM_EINVAL, "from_nodes: node lists are NULL");
    const ExPart part{set_offsets, n_sets, n_seeds};
    return extract_impl(store, seeds, 1, &part, 1, 1, 0, link_pred, nodes_flat, nodes_off, stream, out);
}

extern "C" int gm_batch_concat(const gm_batch_t* const* parts, int32_t n_parts, void* stream, gm_batch_t** out) {
    GM_REQUIRE(out, GM_EINVAL, "concat: out is NULL");
    *out = nullptr;
    GM_REQUIRE(parts && n_parts >= 1, GM_EINVAL, "concat: no parts");
    hipStream_t st = (hipStream_t)stream;
    gm_batch* b = new gm_batch();
    b->store = parts[0]->store; b->centres = parts[0]->centres; b->stream = st;
    int64_t rows = 0, edges = 0, subs = 0, sets = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (!parts[p] || parts[p]->store != b->store || parts[p]->centres != b->centres) {
            delete b; gm_set_error("concat: part %d has a different store or centre count", p); return GM_EINVAL;
        }
        rows += parts[p]->rows; edges += parts[p]->edges; subs += parts[p]->subs; sets += parts[p]->sets;
    }
    if (rows > INT32_MAX - 2 || edges > INT32_MAX - 2) { delete b; gm_set_error("concat: batch exceeds 2^31 rows/edges"); return GM_ERANGE; }
    b->rows = rows; b->edges = edges; b->subs = (int32_t)subs; b->sets = (int32_t)sets;
    b->h_sub_off.assign(1, 0); b->h_set_sub_off.assign(1, 0); b->h_set_row_off.assign(1, 0);
    int rc = batch_alloc(b, st);
    if (rc != GM_OK) { batch_free(b); delete b; return rc; }
    int64_t r0 = 0, e0 = 0; int32_t s0 = 0;
    auto cpy = [&](int32_t* dst, const int32_t* src, int64_t n, int32_t add) {
        if (n <= 0) return;
        const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
        hipLaunchKernelGGL(k_copy_add, dim3(blocks), dim3(256), 0, st, dst, src, n, add);
    };
    for (int p = 0; p < n_parts; ++p) {
        const gm_batch* q = parts[p];
        cpy(b->d_parent + r0, q->d_parent, q->rows, 0); cpy(b->d_feat_row + r0, q->d_feat_row, q->rows, 0);
        cpy((int32_t*)b->d_norm + r0, (const int32_t*)q->d_norm, q->rows, 0);
        cpy(b->d_indptr + r0, q->d_indptr, q->rows + (p == n_parts - 1 ? 1 : 0), (int32_t)e0);
        cpy(b->d_indptr_t + r0, q->d_indptr_t, q->rows + (p == n_parts - 1 ? 1 : 0), (int32_t)e0);
        cpy(b->d_indices + e0, q->d_indices, q->edges, (int32_t)r0); cpy(b->d_indices_t + e0, q->d_indices_t, q->edges, (int32_t)r0);
        cpy(b->d_centre + (int64_t)s0 * b->centres, q->d_centre, (int64_t)q->subs * b->centres, 0);
        for (int k = 1; k <= q->subs; ++k) b->h_sub_off.push_back((int32_t)(r0 + q->h_sub_off[k]));
        for (int k = 1; k <= q->sets; ++k) { b->h_set_sub_off.push_back(s0 + q->h_set_sub_off[k]); b->h_set_row_off.push_back((int32_t)(r0 + q->h_set_row_off[k])); }
        b->h_graph.insert(b->h_graph.end(), q->h_graph.begin(), q->h_graph.end());
        r0 += q->rows; e0 += q->edges; s0 += q->subs;
    }
    if (hipGetLastError() != hipSuccess) { batch_free(b); delete b; gm_set_error("concat: copy kernel launch failed"); return GM_EHIP; }
    gm_stager sg(st);
    rc = upload_small(b, sg);
    if (rc == GM_OK) rc = gm_batch_finalize(b, st, sg);
    if (rc != GM_OK) { batch_free(b); delete b; return rc; }
    *out = b;
    return GM_OK;
}

extern "C" int gm_batch_dims(const gm_batch_t* b, int64_t* rows, int64_t* edges, int32_t* subs, int32_t* sets, int32_t* centres) {
    GM_REQUIRE(b, GM_EINVAL, "batch_dims: NULL batch");
    if (rows) *rows = b->rows; if (edges) *edges = b->edges; if (subs) *subs = b->subs; if (sets) *sets = b->sets;
    if (centres) *centres = b->centres;
    return GM_OK;
}

static int field_ptr(const gm_batch_t* b, int32_t field, void** p, int64_t* bytes) {
    switch (field) {
        case GM_F_SUB_OFF: *p = b->d_sub_off; *bytes = 4ll * (b->subs + 1); break;
        case GM_F_SET_SUB_OFF: *p = b->d_set_sub_off; *bytes = 4ll * (b->sets + 1); break;
        case GM_F_PARENT: *p = b->d_parent; *bytes = 4ll * b->rows; break;
        case GM_F_GRAPH: *p = b->d_graph; *bytes = 4ll * b->subs; break;
        case GM_F_INDPTR: *p = b->d_indptr; *bytes = 4ll * (b->rows + 1); break;
        case GM_F_INDICES: *p = b->d_indices; *bytes = 4ll * b->edges; break;
        case GM_F_INDPTR_T: *p = b->d_indptr_t; *bytes = 4ll * (b->rows + 1); break;
        case GM_F_INDICES_T: *p = b->d_indices_t; *bytes = 4ll * b->edges; break;
        case GM_F_CENTRE: *p = b->d_centre; *bytes = 4ll * b->subs * b->centres; break;
        case GM_F_NORM: *p = b->d_norm; *bytes = 4ll * b->rows; break;
        case GM_F_FEAT_ROW: *p = b->d_feat_row; *bytes = 4ll * b->rows; break;
        default: gm_set_error("unknown batch field %d", field); return GM_EINVAL;
    }
    return GM_OK;
}

extern "C" int gm_batch_read(const gm_batch_t* b, int32_t field, void* host_dst, int64_t bytes) {
    GM_REQUIRE(b && host_dst, GM_EINVAL, "batch_read: NULL argument");
    void* p; int64_t need;
    GM_TRY(field_ptr(b, field, &p, &need));
    GM_REQUIRE(bytes >= need, GM_EINVAL, "batch_read: destination holds %lld bytes, field needs %lld", (long long)bytes, (long long)need);
    if (field == GM_F_CENTRE && (int64_t)b->h_centre.size() * 4 == need) { memcpy(host_dst, b->h_centre.data(), (size_t)need); return GM_OK; }
    GM_HIP(hipMemcpyAsync(host_dst, p, (size_t)need, hipMemcpyDeviceToHost, b->stream));
    GM_HIP(hipStreamSynchronize(b->stream));
    return GM_OK;
}

extern "C" int gm_batch_device_ptr(const gm_batch_t* b, int32_t field, void** dptr) {
    GM_REQUIRE(b && dptr, GM_EINVAL, "batch_device_ptr: NULL argument");
    int64_t bytes;
    return field_ptr(b, field, dptr, &bytes);
}

int gm_gather_rows(const gm_store* store, const int32_t* feat_row, int64_t n, int F, float* out, hipStream_t st) {
    if (n <= 0) return GM_OK;          // F: columns copied (feat_dim, or feat_ld for the padded internal model)
    const int blocks = (int)std::min<int64_t>(256 * 8, (n * F + 255) / 256);
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, st, store->d_feat, (int64_t)store->feat_ld, feat_row, out, n, F);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

extern "C" int gm_gather_features(const gm_batch_t* b, float* x_out, void* stream) {
    GM_REQUIRE(b && x_out, GM_EINVAL, "gather_features: NULL argument");
    const int F = b->store->feat_dim;
    const int64_t total = b->rows * F;
    const int blocks = (int)std::min<int64_t>(256 * 8, (total + 255) / 256);
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b->store->d_feat, (int64_t)b->store->feat_ld, b->d_feat_row, x_out, b->rows, F);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
