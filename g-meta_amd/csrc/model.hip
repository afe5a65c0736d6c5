// Classifier forward/backward (learner.py:134-175), prototypical losses (meta.py:14-79) and the
// first-order ProtoMAML inner/outer step (meta.py:101-173, 175-234) for ALL tasks of a meta-batch
// at once.  Each task (set) owns its fast weights; the K-loop runs here so that one FFI call is one
// meta-step and nothing returns to the host until the accuracies are read back.
#include <algorithm>
#include <atomic>
#include <map>
#include <stdlib.h>
#include "gm_internal.h"


// ================================================================================ small kernels
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct HeadK {
    const float* H; int64_t ldh; int Hd;            // last GCN activation [rows, Hd]
    const int32_t* sub_off; const int32_t* centre; int nc; const int32_t* sub_set; const int32_t* set_sub_off;
    const float* params; int64_t pstride; int64_t wl_off, bl_off; int hc, C; int subs;
    int compact;                                    // 1: H holds only the centre rows, [subs*nc, Hd] in centre order (cone schedule)
    unsigned* dq_amax;                              // optional [sets * GM_BOUND_PAD] (zeroed): receives max |dQ| written for the set (gm_bound.h)
    int subs_per_set;                               // > 0: every set holds this many subgraphs (set t starts at t * subs_per_set: no load of set_sub_off ahead of everything else)
};

__device__ __forceinline__ int64_t centre_row(const HeadK& k, int s, int which) {
    return k.compact ? (int64_t)s * k.nc + which : (int64_t)k.sub_off[s] + k.centre[s * k.nc + which];
}

// Optional fused inner-loop SGD (meta.py:126,151): next_t[j] = cur_t[j] - lr * grad_t[j], written with the gradient.
struct SgdK { const float* cur; int64_t cur_stride; float* next; int64_t next_stride; float lr; };

// ---- address-space policies of the head kernels.  The fused kernel (k_head_loss) keeps a set's centre rows, head weights, logits and dlogits in LDS; the
// unfused kernels (k_head_fwd / k_proto / k_head_bwd: the public per-phase entry points) work on the global arrays.  Written with plain `float*` and a
// run-time `staged ? lds : global` choice, every access became a FLAT instruction (159 flat_load in the fused kernel: LDS data through the vector-memory
// path, several times an ds_read's latency, and both wait counters to drain) -- in a kernel that is one workgroup per task and nothing but latency.  The LDS
// pointers therefore carry their address space in the type, and the two variants are separate instantiations.
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int lds_int;
struct MemG { typedef float* F; typedef const float* CF; typedef const int32_t* CI; static constexpr bool lds = false; };
struct MemL { typedef lds_float* F; typedef const lds_float* CF; typedef const lds_int* CI; static constexpr bool lds = true; };
// Barrier between phases that exchange data through LDS only (MemL): lgkmcnt(0) + s_barrier.  __syncthreads() also drains the vector-memory counter, i.e.
// it waits for the acknowledgement of every global STORE issued before it (the prototypes, the loss / accuracy, the phase stamps: ~1 us each) -- results nobody in
// the block reads.  MemG exchanges the logits / dlogits through global memory and keeps the full barrier.
template <typename M> __device__ __forceinline__ void block_sync() {
    if constexpr (M::lds) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
}

// logits[s,:] = F.linear(cat(h[c0], h[c1]), Wl, bl)  (learner.py:165-175); one wave per subgraph, a 16-lane group per output class: four classes
// are reduced at once with four shuffle steps (a wave-wide reduction per class was 6 dependent ds_bpermute round trips per logit).
// The association is the wave-wide version's, bit for bit: lane l of a group carries the partial sums of the former lanes l, l + 16, l + 32, l + 48
// (columns h = lane mod 64, ascending) and joins them the way the 32- and 16-lane butterfly steps did.  (Not pedantry: fixture g1_sampled_h2 holds rows
// whose layer-2 pre-activation is the bias alone, so the SIGN of a 1e-9-sized bias decides their relu' -- another rounding of the logits moved one
// meta-gradient entry of the flagged schedules by 150 % against the reference's golden value.)
// MemL: hs = the set's centre rows (subgraphs from s0, centre order: cat(h[c0], h[c1]) of a subgraph is contiguous), wl_s = [Wl | bl]; logits holds
// subgraphs [lbase, ..).  MemG: rows of k.H, the set's parameters.
template <typename M>
__device__ __forceinline__ void head_fwd_sub(const HeadK& k, int s, int lane, typename M::F logits, typename M::CF hs, int s0, typename M::CF wl_s, int lbase) {
    const int grp = lane >> 4, l = lane & 15;
    if constexpr (M::lds) {
        typename M::CF x = hs + (s - s0) * k.nc * k.Hd;
        for (int c0 = 0; c0 < k.C; c0 += 4) {
            const int c = c0 + grp, cc = min(c, k.C - 1);
            typename M::CF w = wl_s + cc * k.hc;
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            if ((k.hc & 63) == 0) {                    // the reads of four 64-column steps (all of hc = 256) in flight together
#pragma unroll 4
                for (int hb = l; hb < k.hc; hb += 64) {
                    p[0] += x[hb] * w[hb]; p[1] += x[hb + 16] * w[hb + 16]; p[2] += x[hb + 32] * w[hb + 32]; p[3] += x[hb + 48] * w[hb + 48];
                }
            } else {
                for (int hb = l; hb < k.hc; hb += 64) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) { const int h = hb + 16 * m; if (h < k.hc) p[m] += x[h] * w[h]; }
                }
            }
            float acc = (p[0] + p[2]) + (p[1] + p[3]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (l == 0 && c < k.C) logits[(s - lbase) * k.C + c] = acc + wl_s[k.C * k.hc + c];
        }
    } else {
        const float* P = k.params + (int64_t)k.sub_set[s] * k.pstride;
        const float* h0 = k.H + centre_row(k, s, 0) * k.ldh;
        const float* h1 = k.nc == 2 ? k.H + centre_row(k, s, 1) * k.ldh : h0;
        for (int c0 = 0; c0 < k.C; c0 += 4) {
            const int c = c0 + grp, cc = min(c, k.C - 1);
            const float* w = P + k.wl_off + (int64_t)cc * k.hc;
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            for (int hb = l; hb < k.hc; hb += 64) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { const int h = hb + 16 * m; if (h < k.hc) p[m] += (h < k.Hd ? h0[h] : h1[h - k.Hd]) * w[h]; }
            }
            float acc = (p[0] + p[2]) + (p[1] + p[3]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (l == 0 && c < k.C) logits[(int64_t)(s - lbase) * k.C + c] = acc + P[k.bl_off + c];
        }
    }
}
__global__ __launch_bounds__(256) void k_head_fwd(HeadK k, float* logits) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s < k.subs) head_fwd_sub<MemG>(k, s, threadIdx.x & 63, logits, nullptr, 0, nullptr, 0);
}

// Backward of the head for one set per block: dWl, dbl into dparams; dQ_L (pre-zeroed) at the centre rows,
// already multiplied by relu'(H_L).  With Gc != NULL the rows go to a compact [subs*centres, Hd] matrix instead.
// dlogits holds subgraphs [dbase, ..).  MemL: hs / wl_s as above, crow_s = the rows of the set's centres; pre: u.cur of this thread's first Wl / bl entry was
// loaded by the caller.  s0 / s1: the set's subgraph range.  dbg: phase stamp 5 of block 0 (gm_head_loss_debug).
template <int NT, typename M>
__device__ __forceinline__ void head_bwd_set(const HeadK& k, int set, int tid, typename M::CF dlogits, float* dparams, int64_t dstride, float* dQ,
                                             float* Gc, const SgdK& u, typename M::CF hs, typename M::CF wl_s, int dbase, typename M::CI crow_s,
                                             bool pre, float pre_w, float pre_b, int s0, int s1, unsigned long long* dbg) {
    const float* P = k.params + (int64_t)set * k.pstride;
    auto wl = [&](int i) -> float { if constexpr (M::lds) return wl_s[i]; else return P[k.wl_off + i]; };                   // Wl[C, hc]
    auto feat = [&](int s, int which, int col) -> float {                                                                  // H_L at centre `which` of subgraph s
        if constexpr (M::lds) return hs[((s - s0) * k.nc + which) * k.Hd + col]; else return k.H[centre_row(k, s, which) * k.ldh + col];
    };
    auto crow = [&](int s, int which) -> int64_t { if constexpr (M::lds) return crow_s[(s - s0) * k.nc + which]; else return centre_row(k, s, which); };
    float* D = dparams + (int64_t)set * dstride;
    const int S = s1 - s0;
    for (int id = tid; id < k.C * k.hc; id += NT) {
        const int c = id / k.hc, h = id - c * k.hc;
        float acc = 0.f;
        if constexpr (M::lds) {     // column h of cat(h[c0], h[c1]) of subgraph s0 + j is hs[j * hc + h]: 32-bit index arithmetic, reads issued ahead
            typename M::CF hcol = hs + h;
            typename M::CF dcol = dlogits + (s0 - dbase) * k.C + c;
#pragma unroll 8
            for (int j = 0; j < S; ++j) acc += dcol[j * k.C] * hcol[j * k.hc];
        } else {
            for (int s = s0; s < s1; ++s) acc += dlogits[(int64_t)(s - dbase) * k.C + c] * (h < k.Hd ? feat(s, 0, h) : feat(s, 1, h - k.Hd));
        }
        D[k.wl_off + id] = acc;
        if (u.next) u.next[(int64_t)set * u.next_stride + k.wl_off + id] = ((pre && id == tid) ? pre_w : u.cur[(int64_t)set * u.cur_stride + k.wl_off + id]) - u.lr * acc;
    }
    for (int c = tid; c < k.C; c += NT) {
        float acc = 0.f;
#pragma unroll 8
        for (int s = s0; s < s1; ++s) acc += dlogits[(s - dbase) * k.C + c];
        D[k.bl_off + c] = acc;
        if (u.next) u.next[(int64_t)set * u.next_stride + k.bl_off + c] = ((pre && c == tid) ? pre_b : u.cur[(int64_t)set * u.cur_stride + k.bl_off + c]) - u.lr * acc;
    }
    if (dbg && set == 0 && tid == 0) dbg[5] = wall_clock64();
    // one thread per (subgraph, column); both centres of a pair stay in one thread (they may share a row).  dQ is all zeros when this runs and a row is the
    // centre of ONE subgraph, so the "+=" of learner.py's index_select backward is a plain store (no dependent read of dQ); the two centres of a pair may be
    // the same row: (0 + v0) + v1
    float dq_max = 0.f;
    constexpr int CMAX = 8;
    if (M::lds && !Gc && NT % k.Hd == 0 && k.C <= CMAX) {
        // dense dQ from the staged rows: a thread keeps ONE column (NT is a multiple of Hd) and walks the subgraphs NT / Hd apart -- no division in the loop,
        // the column's head weights in registers; per element the same operations in the same order as the general loop below
        const int col = tid % k.Hd, sstep = NT / k.Hd;
        float wv[2][CMAX];
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int c = 0; c < CMAX; ++c) wv[which][c] = (which < k.nc && c < k.C) ? wl(c * k.hc + which * k.Hd + col) : 0.f;
        for (int j = tid / k.Hd; j < S; j += sstep) {
            typename M::CF dl_j = dlogits + (s0 - dbase + j) * k.C;
            float vv[2] = {0.f, 0.f};
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (which < k.nc) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) if (c < k.C) v += dl_j[c] * wv[which][c];
                    vv[which] = feat(s0 + j, which, col) > 0.f ? v : 0.f;
                }
            }
            const int64_t r0 = crow(s0 + j, 0);
            if (k.nc == 2) {
                const int64_t r1 = crow(s0 + j, 1);
                if (r1 == r0) vv[0] = vv[0] + vv[1]; else { dQ[r1 * k.ldh + col] = vv[1]; dq_max = fmaxf(dq_max, fabsf(vv[1])); }
            }
            dQ[r0 * k.ldh + col] = vv[0];
            dq_max = fmaxf(dq_max, fabsf(vv[0]));
        }
    } else {
        for (int id = tid; id < S * k.Hd; id += NT) {
            const int s = s0 + id / k.Hd, col = id % k.Hd;
            float vv[2] = {0.f, 0.f};
            for (int which = 0; which < k.nc; ++which) {
                float v = 0.f;
                for (int c = 0; c < k.C; ++c) v += dlogits[(s - dbase) * k.C + c] * wl(c * k.hc + which * k.Hd + col);
                const float hval = feat(s, which, col);
                if (Gc) Gc[((int64_t)s * k.nc + which) * k.Hd + col] = hval > 0.f ? v : 0.f;     // compact rows (sparse backward / cone)
                else vv[which] = hval > 0.f ? v : 0.f;
            }
            if (!Gc) {
                const int64_t r0 = crow(s, 0);
                if (k.nc == 2) {
                    const int64_t r1 = crow(s, 1);
                    if (r1 == r0) vv[0] = vv[0] + vv[1]; else { dQ[r1 * k.ldh + col] = vv[1]; dq_max = fmaxf(dq_max, fabsf(vv[1])); }
                }
                dQ[r0 * k.ldh + col] = vv[0];
                dq_max = fmaxf(dq_max, fabsf(vv[0]));
            }
        }
    }
    if (k.dq_amax && !Gc) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dq_max = fmaxf(dq_max, __shfl_xor(dq_max, o));
        if ((tid & 63) == 0 && dq_max > 0.f) atomicMax(k.dq_amax + (int64_t)set * GM_BOUND_PAD, __float_as_uint(dq_max));      // (a handful per set)
    }
}
__global__ __launch_bounds__(256) void k_head_bwd(HeadK k, const float* dlogits, float* dparams, int64_t dstride, float* dQ, float* Gc) {
    const int set = blockIdx.x;
    head_bwd_set<256, MemG>(k, set, threadIdx.x, dlogits, dparams, dstride, dQ, Gc, SgdK{nullptr, 0, nullptr, 0, 0.f}, nullptr, nullptr, 0, nullptr, false, 0.f, 0.f,
                            k.set_sub_off[set], k.set_sub_off[set + 1], nullptr);
}

// Prototypical loss for one set per block (meta.py:28-79).  rows: [sets, Ct, n] global subgraph ids grouped by
// sorted class.  mode 0 = support (prototypes = class means of the same rows), 1 = query (prototypes given).
struct ProtoK {
    const float* logits; int D; const int32_t* rows; int Ct, n; int mode;      // Ct, n: the LARGEST class count / rows per class over the sets
    const float* protos_in; float* protos_out; float* loss; float* acc; int64_t ld_out; int col_out;
    float* dlogits; float* dprotos;
    int row_base;                         // logits / dlogits hold subgraphs [row_base, ...): 0 for the global arrays, the set's first
                                          // subgraph when the fused kernel keeps them in LDS
    const int32_t* tab;                   // [sets*3] per set: offset into rows, classes, rows per class (every task keeps its own class
                                          // layout, like the per-task proto_loss calls of meta.py:118-157); prototypes are strided by Ct
    int uniform;                          // every set has Ct classes of n rows at offset set * Ct * n (the usual case): the kernels take the layout from the arguments
};                                        // instead of a dependent load of `tab` ahead of the class rows

template <typename XP>
__device__ __forceinline__ float sqdist(XP x, const lds_float* p, int D) {
    float d = 0.f;
    for (int k = 0; k < D; ++k) { const float t = x[k] - p[k]; d += t * t; }
    return d;
}

// LDS of a set: prototypes [Ct*D] | lse [Ct*n] | block-sum scratch [2*NT] | A [Ct*n*Ct] (only when it is at most PROTO_A_MAX floats).
// A holds -dist(q, c) and then G[q,c] = (softmax(-d)[q,c] - [c == class(q)]) / Q: every later phase reads it instead of redoing the distance and the
// exponential, and the loops that must stay serial (a class sum over its rows, a prototype's gradient over the queries: their order is part of the
// result) carry nothing but LDS reads that can be issued ahead.  The kernel is one workgroup per task and pure latency.
// logits / dlogits / rows: the arrays the set is scored on -- k's global ones (MemG; rows already offset to the set) or the fused kernel's LDS copies (MemL).
#define PROTO_A_MAX 8192
template <int NT, typename M>
__device__ __forceinline__ void proto_set(const ProtoK& k, int set, int tid, lds_float* sm, typename M::CF logits, typename M::F dlogits, typename M::CI rows,
                                          bool have_proto = false, float proto_pre = 0.f) {      // have_proto: protos_in[set][tid] was loaded by the caller (tid < Ct * D)
    const int Ct = k.uniform ? k.Ct : k.tab[set * 3 + 1], n = k.uniform ? k.n : k.tab[set * 3 + 2];
    const int Q = Ct * n, D = k.D, rb = k.row_base;
    lds_float* protos = sm;             // [Ct*D]   (LDS sized for the largest set)
    lds_float* lse = sm + k.Ct * D;     // [Q]
    lds_float* red = lse + k.Ct * k.n;  // [2 * NT]
    lds_float* A = red + 2 * NT;        // [Q * Ct]
    const bool useA = (int64_t)k.Ct * k.n * k.Ct <= PROTO_A_MAX;
    for (int id = tid; id < Ct * D; id += NT) {
        const int c = id / D, d = id - c * D;
        float p;
        if (k.mode == 0) {
            p = 0.f;
#pragma unroll 4
            for (int r = 0; r < n; ++r) p += logits[(rows[c * n + r] - rb) * D + d];
            p /= (float)n;                                                          // .mean(0) (meta.py:41)
            if (k.protos_out) k.protos_out[(int64_t)set * k.Ct * D + id] = p;
        } else {
            p = (have_proto && id == tid) ? proto_pre : k.protos_in[(int64_t)set * k.Ct * D + id];
        }
        protos[id] = p;
    }
    block_sync<M>();
    if (useA) {
        for (int id = tid; id < Q * Ct; id += NT) {
            const int q = id / Ct, c = id - q * Ct;
            A[id] = -sqdist(logits + (rows[q] - rb) * D, protos + c * D, D);      // -dists (meta.py:44-45)
        }
        block_sync<M>();
    }
    float lpart = 0.f, apart = 0.f;
    for (int q = tid; q < Q; q += NT) {
        typename M::CF x = logits + (rows[q] - rb) * D;
        const int tgt = q / n;
        float m = -INFINITY, at = 0.f;
        for (int c = 0; c < Ct; ++c) {
            const float a = useA ? A[q * Ct + c] : -sqdist(x, protos + c * D, D);
            m = fmaxf(m, a);
            if (c == tgt) at = a;
        }
        float se = 0.f;
        for (int c = 0; c < Ct; ++c) se += expf((useA ? A[q * Ct + c] : -sqdist(x, protos + c * D, D)) - m);
        const float lg = logf(se);
        lse[q] = m + lg;
        lpart += -((at - m) - lg);                                                   // -log_p[q, class(q)], log_softmax = (x - max) - log(sum exp(x - max))
        // y_hat = log_p_y.max(2) (meta.py:52,76): the prediction is the FIRST maximum of the fp32 LOG-PROBABILITIES, not of the distances --
        // when the logits are tiny (distances below one ulp of log(sum)) every class rounds to the same log-probability and the reference
        // predicts class 0; reproduced (fixture g8_wide_scales)
        float bl = -INFINITY; int best = 0;
        for (int c = 0; c < Ct; ++c) {
            const float lp = ((useA ? A[q * Ct + c] : -sqdist(x, protos + c * D, D)) - m) - lg;
            if (lp > bl) { bl = lp; best = c; }
        }
        apart += (best == tgt) ? 1.f : 0.f;
    }
    // block sum of (loss, correct): wave shuffles, then the first wave adds the NT / 64 wave partials (two barriers instead of log2 NT)
    lpart = wave_sumf(lpart); apart = wave_sumf(apart);
    if ((tid & 63) == 0) { red[tid >> 6] = lpart; red[NT + (tid >> 6)] = apart; }
    block_sync<M>();
    if (tid < 64) {
        float l2 = tid < NT / 64 ? red[tid] : 0.f, a2 = tid < NT / 64 ? red[NT + tid] : 0.f;
        l2 = wave_sumf(l2); a2 = wave_sumf(a2);
        if (tid == 0) {
            k.loss[(int64_t)set * k.ld_out + k.col_out] = l2 / (float)Q;
            k.acc[(int64_t)set * k.ld_out + k.col_out] = a2 / (float)Q;
        }
    }
    if (!dlogits) return;
    // G[q,c] = (softmax(-d)[q,c] - [c == tgt(q)]) / Q ;  d(-d_qc)/dx_q = -2 (x_q - p_c) ; d(-d_qc)/dp_c = +2 (x_q - p_c)
    const float invQ = 1.f / (float)Q;
    if (useA) {
        for (int id = tid; id < Q * Ct; id += NT) {
            const int q = id / Ct, c = id - q * Ct;
            A[id] = (expf(A[id] - lse[q]) - (c == q / n ? 1.f : 0.f)) * invQ;
        }
        block_sync<M>();
    }
    for (int id = tid; id < Q * D; id += NT) {
        const int q = id / D, d = id - q * D;
        typename M::CF x = logits + (rows[q] - rb) * D;
        const int tgt = q / n;
        float s = 0.f;
        for (int c = 0; c < Ct; ++c) {
            const float g = useA ? A[q * Ct + c] : (expf(-sqdist(x, protos + c * D, D) - lse[q]) - (c == tgt ? 1.f : 0.f)) * invQ;
            s += g * -2.f * (x[d] - protos[c * D + d]);
        }
        dlogits[(rows[q] - rb) * D + d] = s;
    }
    block_sync<M>();
    for (int id = tid; id < Ct * D; id += NT) {
        const int c = id / D, d = id - c * D;
        const float pd = protos[id];
        float s = 0.f;
        if (useA) {
#pragma unroll 8
            for (int q = 0; q < Q; ++q) s += A[q * Ct + c] * 2.f * (logits[(rows[q] - rb) * D + d] - pd);
        } else {
            for (int q = 0; q < Q; ++q) {
                typename M::CF x = logits + (rows[q] - rb) * D;
                const float g = (expf(-sqdist(x, protos + c * D, D) - lse[q]) - (c == q / n ? 1.f : 0.f)) * invQ;
                s += g * 2.f * (x[d] - pd);
            }
        }
        if (k.mode == 0) {
#pragma unroll 4
            for (int r = 0; r < n; ++r) dlogits[(rows[c * n + r] - rb) * D + d] += s / (float)n;
        } else if (k.dprotos) {
            k.dprotos[(int64_t)set * k.Ct * D + id] = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_proto(ProtoK k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int set = blockIdx.x;
    proto_set<256, MemG>(k, set, threadIdx.x, (lds_float*)sm, k.logits, k.dlogits, k.rows + (k.uniform ? set * k.Ct * k.n : k.tab[set * 3]));
}

// Head forward + prototypical loss (+ head backward and the SGD of the head's own parameters) of one set per block: the
// five launches between the last GCN layer of a forward and the first weight gradient of its backward, in one.
#define HL_THREADS 1024     // largest workgroup of k_head_loss (LDS sizing)
template <int NT>
__global__ __launch_bounds__(NT) void k_head_loss(HeadK hk, float* logits, ProtoK pk, int do_bwd, float* dparams, int64_t dstride, float* dQ,
                                                          float* Gc, SgdK u, int stage, int proto_floats, int copy_out, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float sm_generic[];
    lds_float* sm = (lds_float*)sm_generic;
    const int set = blockIdx.x, tid = threadIdx.x;
#define HL_STAMP(K) do { if (dbg && set == 0 && tid == 0) dbg[K] = wall_clock64(); } while (0)      // phase timeline of block 0 (tools/head_loss_probe.py)
    HL_STAMP(0);
    const long long cyc0 = dbg ? clock64() : 0;      // shader-clock cycles next to the constant-clock stamps: stamp 7 = cycles spent (what clock the probe ran at)
    const int s0 = hk.subs_per_set ? set * hk.subs_per_set : hk.set_sub_off[set], s1 = hk.subs_per_set ? s0 + hk.subs_per_set : hk.set_sub_off[set + 1], S = s1 - s0, D = pk.D;
    // the head's own SGD inputs (w - lr * g, meta.py:126,151) go out with the first round trip instead of behind the loss
    const bool pre = do_bwd && u.next != nullptr;
    float pre_w = 0.f, pre_b = 0.f;
    if (pre) {
        if (tid < hk.C * hk.hc) pre_w = u.cur[(int64_t)set * u.cur_stride + hk.wl_off + tid];
        if (tid < hk.C) pre_b = u.cur[(int64_t)set * u.cur_stride + hk.bl_off + tid];
    }
    // ... and so do the prototypes a query loss is scored against (mode 1; uniform class layout: the set's are at set * Ct * D)
    const bool have_proto = stage && pk.mode == 1 && pk.uniform && pk.Ct * D <= NT;
    float proto_pre = 0.f;
    if (have_proto && tid < pk.Ct * D) proto_pre = pk.protos_in[(int64_t)set * pk.Ct * D + tid];
    if (!stage) {
        // everything in the global arrays (sets too large for LDS): the three phases with L2 round trips in between
        if (pk.dlogits) for (int id = tid; id < S * D; id += NT) pk.dlogits[(int64_t)s0 * D + id] = 0.f;   // rows outside the class tables
        for (int s = s0 + (tid >> 6); s < s1; s += NT / 64) head_fwd_sub<MemG>(hk, s, tid & 63, logits, nullptr, 0, nullptr, 0);
        __syncthreads();          // workgroup-scope fence: the logits / zeros written above are visible to the whole block
        HL_STAMP(3);
        proto_set<NT, MemG>(pk, set, tid, sm, logits, pk.dlogits, pk.rows + (pk.uniform ? set * pk.Ct * pk.n : pk.tab[set * 3]));
        HL_STAMP(4);
        if (!do_bwd) return;
        __syncthreads();
        head_bwd_set<NT, MemG>(hk, set, tid, pk.dlogits, dparams, dstride, dQ, Gc, u, nullptr, nullptr, 0, nullptr, pre, pre_w, pre_b, s0, s1, dbg);
        HL_STAMP(6);
        return;
    }
    // staged: everything the three phases share stays in LDS -- the set's centre rows of H_L, its head weights, its
    // logits and dlogits -- so that a phase costs LDS latency instead of an L2 round trip (the kernel is pure latency)
    lds_float* hs = sm + proto_floats;                                      // [S * nc, Hd] centre rows of H_L, centre order
    lds_float* wl_s = hs + S * hk.nc * hk.Hd;                               // [C * hc | C]  Wl | bl
    lds_float* lg_s = wl_s + hk.C * hk.hc + hk.C;                           // [S, D] logits
    lds_float* dl_s = lg_s + S * D;                                         // [S, D] dlogits
    lds_int* crow = (lds_int*)(dl_s + S * D);                               // [S * nc] row of every centre
    lds_int* rows_l = crow + S * hk.nc;                                     // [Ct * n] the set's class rows (subgraph ids)
    const float* P = hk.params + (int64_t)set * hk.pstride;
    const int nq = S * hk.nc;
    // first round trip, everything that needs no other load: centre rows (two dependent loads each), class rows, head weights
    for (int q = tid; q < nq; q += NT) crow[q] = (int)centre_row(hk, s0 + q / hk.nc, q % hk.nc);
    {
        const int off = pk.uniform ? set * pk.Ct * pk.n : pk.tab[set * 3], Qs = pk.uniform ? pk.Ct * pk.n : pk.tab[set * 3 + 1] * pk.tab[set * 3 + 2];
        for (int q = tid; q < Qs; q += NT) rows_l[q] = pk.rows[off + q];
    }
    for (int id = tid; id < hk.C * hk.hc; id += NT) wl_s[id] = P[hk.wl_off + id];
    for (int id = tid; id < hk.C; id += NT) wl_s[hk.C * hk.hc + id] = P[hk.bl_off + id];
    block_sync<MemL>();
    HL_STAMP(1);
    // second round trip: the centre rows of H_L, all loads of a thread in flight together (16-byte loads when the rows allow it).  (The first version was a
    // serial chain of 18 dependent global loads per thread.)
    if ((hk.Hd & 3) == 0 && (hk.ldh & 3) == 0 && ((uintptr_t)hk.H & 15) == 0) {
        const int hd4 = hk.Hd >> 2, total4 = nq * hd4;
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) f4v lds_f4v;
        for (int id0 = tid; id0 < total4; id0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = min(id0 + j * NT, total4 - 1), q = id / hd4, c4 = id - q * hd4;
                v[j] = *reinterpret_cast<const float4*>(hk.H + (int64_t)crow[q] * hk.ldh + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int id = id0 + j * NT; if (id < total4) { const f4v t = {v[j].x, v[j].y, v[j].z, v[j].w}; *(lds_f4v*)(hs + id * 4) = t; } }
        }
    } else {
        const int total = nq * hk.Hd;
        for (int id0 = tid; id0 < total; id0 += 8 * NT) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = min(id0 + j * NT, total - 1), q = id / hk.Hd, col = id - q * hk.Hd;
                v[j] = hk.H[(int64_t)crow[q] * hk.ldh + col];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int id = id0 + j * NT; if (id < total) hs[id] = v[j]; }
        }
    }
    lds_float* dl = pk.dlogits ? dl_s : nullptr;        // (pk.dlogits only says WHETHER the gradient is wanted; the global array is written with copy_out)
    if (dl) for (int id = tid; id < S * D; id += NT) dl_s[id] = 0.f;
    block_sync<MemL>();
    HL_STAMP(2);
    if (dbg && set == 0 && (tid & 63) == 0) dbg[8 + (tid >> 6)] = wall_clock64();
    for (int s = s0 + (tid >> 6); s < s1; s += NT / 64) head_fwd_sub<MemL>(hk, s, tid & 63, lg_s, hs, s0, wl_s, s0);
    if (dbg && set == 0 && (tid & 63) == 0) dbg[24 + (tid >> 6)] = wall_clock64();      // per-wave start / end of the logits phase
    block_sync<MemL>();
    HL_STAMP(3);
    ProtoK pl = pk;
    pl.row_base = s0;                                   // the LDS copies hold the set's subgraphs only
    proto_set<NT, MemL>(pl, set, tid, sm, lg_s, dl, rows_l, have_proto, proto_pre);
    HL_STAMP(4);
    if (copy_out) {  // the global copies (the public gm_proto_loss_* shape; nobody inside gm_meta_step reads them): logits always, dlogits when requested
        block_sync<MemL>();
        for (int id = tid; id < S * D; id += NT) {
            logits[(int64_t)s0 * D + id] = lg_s[id];
            if (dl) pk.dlogits[(int64_t)s0 * D + id] = dl_s[id];
        }
    }
    if (!do_bwd) return;
    block_sync<MemL>();
    head_bwd_set<NT, MemL>(hk, set, tid, dl_s, dparams, dstride, dQ, Gc, u, hs, wl_s, s0, crow, pre, pre_w, pre_b, s0, s1, dbg);
    HL_STAMP(6);
    if (dbg && set == 0 && tid == 0) dbg[7] = (unsigned long long)(clock64() - cyc0);
#undef HL_STAMP
}

// Prototype path back into the support logits (prototype_c = mean of the class's first n rows).
__global__ void k_protos_to_dlogits(const float* dprotos, const int32_t* rows, const int32_t* tab, int CtMax, int D, float* dlogits) {
    const int set = blockIdx.x, Ct = tab[set * 3 + 1], n = tab[set * 3 + 2];
    rows += tab[set * 3];
    for (int id = threadIdx.x; id < Ct * n * D; id += blockDim.x) {
        const int q = id / D, d = id - q * D, c = q / n;
        dlogits[(int64_t)rows[q] * D + d] = dprotos[((int64_t)set * CtMax + c) * D + d] / (float)n;
    }
}

// Row-sparse backward, expansion through the transposed aggregate: for every in-edge e = (u -> centre k)
//   G1[e,:] = norm[u] * relu'(H1[u,:]) * T2[k,:]      (dQ_{L-1} restricted to the rows that can be non-zero)
__global__ void k_expand_edges(const float* T2, const float* H1, int F, const int32_t* e_row, const int32_t* e_par, const float* e_norm,
                               int n_e, float* G1) {
    const int64_t tot = (int64_t)n_e * F;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < tot; id += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(id / F), f = (int)(id - (int64_t)e * F);
        const float h = H1[(int64_t)e_row[e] * F + f];
        G1[id] = h > 0.f ? e_norm[e] * T2[(int64_t)e_par[e] * F + f] : 0.f;
    }
}

// WT[t][k][n] = W_t[n][k]: the dZ GEMM (dQ @ W^T) then runs as a plain row-major product on the DMA-fed kernel.
// W_t = params + t*pstride + w_off, [fi, fo] row-major; WT_t [fo, fi].  32x32 tiles through LDS, grid (fo/32, fi/32, sets).
__global__ __launch_bounds__(256) void k_transpose_w(const float* params, int64_t pstride, int64_t w_off, int fi, int fo, float* WT) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const float* W = params + (int64_t)t * pstride + w_off;
    float* O = WT + (int64_t)t * fi * fo;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) { const int n = n0 + r, k = k0 + tx; tile[r][tx] = (n < fi && k < fo) ? W[(int64_t)n * fo + k] : 0.f; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int k = k0 + r, n = n0 + tx; if (k < fo && n < fi) O[(int64_t)k * fi + n] = tile[tx][r]; }
}

// db[set, n] = sum over the set's rows [set_off[set], set_off[set+1]) of G[row, n]   (cone schedule, multiply-first layers)
__global__ void k_colsum_rows(const float* G, int64_t ldg, int N, const int32_t* set_off, float* db, int64_t db_stride, SgdK u, int64_t b_off) {
    const int set = blockIdx.x, r0 = set_off[set], r1 = set_off[set + 1];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += G[(int64_t)r * ldg + n];
        db[(int64_t)set * db_stride + n] = s;
        if (u.next) u.next[(int64_t)set * u.next_stride + b_off + n] = u.cur[(int64_t)set * u.cur_stride + b_off + n] - u.lr * s;
    }
}

// out = [ sum_t (gq+gp) | sum_t lq[t,:] | sum_t aq[t,:] | T | aq[t,:] ... ]
// (cut, shift): the kernels' parameter layout has `shift` extra floats after position `cut` (zero-padded W_1 rows when the store pads
// its feature columns); `out` uses the caller's layout.
// The last float is the step's violation word (gm_bound.h GM_VIOL_*; 0 unless the opt-in two-piece kernels found one of their bounds broken).  A step
// with a violation also reports losses_q[K] = NaN, so that every rank of a sharded meta-batch skips the optimiser step on the REDUCED loss
// (meta.py:163) and the host mirror can re-run the step with the three-piece kernels.
__global__ void k_finalize(const float* gq, const float* gp, int64_t stride, int64_t P, int T, const float* lq, const float* aq, int K1, float* out,
                           int64_t cut, int64_t shift, const unsigned* viol) {
    const int64_t tot = P + 2 * K1 + 1 + (int64_t)T * K1;
    const unsigned vw = viol ? *viol : 0u;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id <= tot; id += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (id == tot) { out[id] = (float)vw; continue; }
        if (id < P) { const int64_t src = id < cut ? id : id + shift; if (gq) for (int t = 0; t < T; ++t) s += gq[t * stride + src] + gp[t * stride + src]; }
        else if (id < P + K1) { for (int t = 0; t < T; ++t) s += lq[t * K1 + (id - P)]; if (vw && id == P + K1 - 1) s = __uint_as_float(0x7fc00000u); }
        else if (id < P + 2 * K1) { for (int t = 0; t < T; ++t) s += aq[t * K1 + (id - P - K1)]; }
        else if (id == P + 2 * K1) s = (float)T;
        else s = aq[id - P - 2 * K1 - 1];
        out[id] = s;
    }
}

// ================================================================================ workspace carving
struct Carver {
    char* base; int64_t used = 0, cap;
    Carver(void* p, int64_t c) : base((char*)p), cap(c) {}
    template <class T> T* take(int64_t n) {
        const int64_t bytes = ((n > 0 ? n : 1) * (int64_t)sizeof(T) + 255) / 256 * 256;
        T* r = base ? (T*)(base + used) : nullptr;
        used += bytes;
        return r;
    }
    bool ok() const { return !base || used <= cap; }
};

// Split-bf16 operand planes of the fast weights fw_1..fw_K, written by the weight-gradient reduction that produces each vector
// (gm_wgrad_args::pl_fwd / pl_dz) and looked up by the GEMMs that consume it -- on either stream; the vector's own ready event
// orders them.  One slot per (k, layer, orientation); theta (shared by all tasks) is split on the fly: one set, one tiny launch.
#define GM_W_HEADROOM 1024.f
struct PlaneDir {
    const float* fw0 = nullptr; int64_t TP = 0; int K = 0;                  // fw_k = fw0 + (k-1) * TP, k = 1..K
    uint16_t* base = nullptr; int64_t per_k = 0;                            // planes of fw_k at base + (k-1) * per_k
    int64_t off[GM_MAX_GCN][2] = {};                                        // [layer][0 fwd, 1 dz] inside a k block; -1 = not kept
    bool valid[64][GM_MAX_GCN][2] = {};
    int index_of(const float* params) const {                               // 1..K, or 0
        if (!base || !fw0 || params < fw0) return 0;
        const int64_t d = params - fw0;
        if (d % TP) return 0;
        const int64_t k = d / TP + 1;
        return (k >= 1 && k <= K && k < 64) ? (int)k : 0;
    }
    uint16_t* slot(int k, int l, int o) const { return (k && off[l][o] >= 0) ? base + (int64_t)(k - 1) * per_k + off[l][o] : nullptr; }
    uint16_t* lookup(const float* params, int l, int o) const { const int k = index_of(params); return (k && off[l][o] >= 0 && valid[k][l][o]) ? slot(k, l, o) : nullptr; }
    // Weight bound for the two-piece fp16 planes (gm_bound.h): wam[l * GM_BOUND_PAD] = bit pattern of max |W_l| of THETA, taken once per
    // meta-step; every weight vector of the step (theta and the fast weights fw_1..fw_K of every task) is split under GM_W_HEADROOM x that
    // bound -- the planes of fw_k are written by the reduction that produces fw_k, before its own maximum could be known.  A weight that
    // outgrows theta's largest by that factor inside one inner loop turns into inf / NaN losses, which the caller sees.
    unsigned* wam = nullptr; const float* theta = nullptr;
    unsigned* viol = nullptr;                                               // the step's violation word (gm_bound.h)
    bool w_bound(const float* params, int l, gm_bound& bd) const {
        bd = gm_no_bound();
        if (!wam || !(index_of(params) || params == theta)) return false;
        bd.amax = wam + (int64_t)l * GM_BOUND_PAD; bd.stride = 0; bd.gain = nullptr; bd.hgain = GM_W_HEADROOM; bd.viol = viol;
        return true;
    }
};

struct GcnCtx {
    const gm_batch* b; gm_layout L;
    PlaneDir* pd;                // non-NULL inside gm_meta_step
    bool is_support = false;     // gm_meta_step's support-chain context (the serial dependency of the step; the query contexts carry the bulk work)
    int dq_zeroed = 0;           // the last forward GEMM already zero-filled bufA (= dQ) for the head/loss launch that follows
    bool zfused[GM_MAX_GCN] = {};    // the last forward left Z[l] written at the rows of three or more sources only (fused aggregate + GEMM in a pass that IS
                                     // differentiated): the backward's weight gradient forms the other rows from the per-row source table (gm_wgrad_args::fuse2)
    float* Z[GM_MAX_GCN]; float* H[GM_MAX_GCN]; float* X0; float* bufA; float* bufB; float* partial;
    float* partial_l[GM_MAX_GCN];    // dense backward: own partials for the layers above the first, whose reductions are held back and run
    gm_wgrad_hold hold;              // together with the first layer's (one launch less per layer and backward pass)
    uint16_t* Wsplit;            // per-task weights of the GEMM being launched as three bf16 planes (split-bf16 kernel, gemm_split.h)
    float* WTl[GM_MAX_GCN];      // per-task transposed weights of layer l >= 1, [set][fo][fi]: B of the dZ GEMM (dense backward)
    const float* wt_of[GM_MAX_GCN]; int64_t wt_stride[GM_MAX_GCN];      // the parameter vector (pointer, per-set stride) WTl[l] is the transpose of
    uint8_t* M[GM_MAX_GCN];      // packed relu' bits of H[l] (one byte per 4 columns): what the backward reads instead of H[l] (dense schedule)
    float* cG2; float* cT2; float* cG1; float* partial_c;      // compact matrices of the row-sparse backward
    const float* x0_user; const int32_t* centre; int z1_valid;
    int zw[GM_MAX_GCN];
    const gm_cone* cone;       // non-NULL: receptive-field schedule, every buffer is compact (gm_hparams_t.cone)
    SgdK sgd;                  // next != NULL: the backward also writes the SGD-updated parameters (inner loop)
    // Two-piece fp16 split kernels (np == 2, gm_meta_step's dense schedule; gm_bound.h): per-pass slots am[pass][2 Lg + 1][sets] (zeroed once per
    // meta-step) receive the per-set maxima of H_l (forward GEMM epilogues), T_l (dZ GEMM epilogues) and dQ_L (head backward); hv / tv / dqv:
    // recorded in the current pass.  A launch whose bounds are not all there runs the three-piece bf16 kernels.
    int hub_set = 0;           // which of the batch's two hub counter / partial-row sets this context's aggregate launches use (gm_batch_hub_alt)
    int np = 3; unsigned* am = nullptr; int am_passes = 0, am_pass = -1;
    const unsigned* feat_bound = nullptr;      // [sets] per-task bound of the layer-1 operand: the largest |feature| of the graphs the task draws from (NULL: loose table, three-piece)
    bool hv[GM_MAX_GCN] = {}, tv[GM_MAX_GCN] = {}; bool dqv = false;
    unsigned* am_slot(int i) const { return am + ((int64_t)am_pass * (2 * L.n_gcn + 1) + i) * b->sets * GM_BOUND_PAD; }
    unsigned* amH(int l) const { return am_slot(l); }
    unsigned* amT(int l) const { return am_slot(L.n_gcn + l); }
    unsigned* amdQ() const { return am_slot(2 * L.n_gcn); }
};
// bound of Z_l, the aggregate of layer l's input (features or H_{l-1}): the A operand of the forward GEMM and of the weight gradient
static bool in_bound(const GcnCtx& c, int l, gm_bound& bd) {
    bd = gm_no_bound();
    if (c.np != 2 || c.am_pass < 0 || !c.b->d_gain) return false;
    if (l == 0) {
        if (c.x0_user || !c.feat_bound) return false;
        bd.amax = c.feat_bound; bd.stride = 1;
    } else {
        if (!c.hv[l - 1]) return false;
        bd.amax = c.amH(l - 1); bd.stride = GM_BOUND_PAD;
    }
    bd.gain = c.b->d_gain; bd.hgain = 1.f;
    return true;
}
// bound of dQ_l (the gradient at layer l's output): from the head for the last layer, else relu' * norm * A^T T_{l+1}
static bool dq_bound(const GcnCtx& c, int l, gm_bound& bd) {
    bd = gm_no_bound();
    if (c.np != 2 || c.am_pass < 0 || !c.b->d_gain) return false;
    if (l == c.L.n_gcn - 1) { if (!c.dqv) return false; bd.amax = c.amdQ(); bd.gain = nullptr; }
    else { if (!c.tv[l + 1]) return false; bd.amax = c.amT(l + 1); bd.gain = c.b->d_gain + 1; }
    bd.stride = GM_BOUND_PAD; bd.hgain = 1.f;
    return true;
}
// The weight planes of a split GEMM of layer l (o = 0: X @ W, 1: dQ @ W^T) and their bound: the planes a reduction left for `params`, else
// split now into c.Wsplit.  want16: the launch has its A bound and would take two-piece planes.  *np = the pieces (2 / 3) of *pl.
static int weight_planes(GcnCtx& c, const float* params, int64_t pstride, int l, int o, int K, int N, bool want16, hipStream_t st, const uint16_t** pl, gm_bound* bb, int* np) {
    const gm_layout& L = c.L;
    uint16_t* have = (c.pd && pstride) ? c.pd->lookup(params, l, o) : nullptr;
    *bb = gm_no_bound();
    if (have && c.np == 2) {                                               // stored planes are two-piece in this context
        if (want16 && c.pd->w_bound(params, l, *bb)) { *pl = have; *np = 2; return GM_OK; }
        have = nullptr;
    }
    if (have) { *pl = have; *np = 3; return GM_OK; }
    const bool f16 = want16 && c.pd && c.pd->w_bound(params, l, *bb);
    if (!f16) *bb = gm_no_bound();
    GM_TRY(gm_split_weights(params, pstride, L.w_off[l], K, N, o, pstride ? c.b->sets : 1, c.Wsplit, st, f16 ? 2 : 3, *bb));
    *pl = c.Wsplit; *np = f16 ? 2 : 3;
    return GM_OK;
}

static void wgrad_sgd(gm_wgrad_args& w, const GcnCtx& c, int l) {
    w.sgd_cur = c.sgd.cur; w.sgd_cur_stride = c.sgd.cur_stride; w.sgd_next = c.sgd.next; w.sgd_next_stride = c.sgd.next_stride; w.sgd_lr = c.sgd.lr;
    w.w_off = c.L.w_off[l]; w.b_off = c.L.b_off[l];
}

static void gcn_carve(GcnCtx& c, Carver& cv) {
    const gm_layout& L = c.L; const int64_t rows = c.b->rows;
    int maxd = 0; int64_t maxkn = 0;
    if (c.cone) {
        const gm_cone* cn = c.cone;
        int64_t maxn = 1; int maxc = 1;
        for (int l = 0; l <= L.n_gcn; ++l) { maxn = std::max<int64_t>(maxn, cn->lv[l].n); maxc = std::max(maxc, cn->lv[l].n_chunks); }
        for (int l = 0; l < L.n_gcn; ++l) {
            const int fi = L.dims[l], fo = L.dims[l + 1];
            c.zw[l] = fi > fo ? fo : fi;
            c.Z[l] = cv.take<float>((int64_t)(fi > fo ? cn->lv[l].n : cn->lv[l + 1].n) * c.zw[l]);   // multiply-first: Y lives on the source level
            c.H[l] = cv.take<float>((int64_t)cn->lv[l + 1].n * fo);
            maxd = std::max(maxd, std::max(fi, fo));
            maxkn = std::max<int64_t>(maxkn, (int64_t)(fi + 1) * fo);
        }
        c.X0 = (L.dims[0] > L.dims[1]) ? cv.take<float>((int64_t)cn->lv[0].n * L.dims[0]) : nullptr;
        c.bufA = cv.take<float>(maxn * maxd);
        c.bufB = cv.take<float>(maxn * maxd);
        c.partial = cv.take<float>((int64_t)maxc * maxkn);
        c.cG2 = c.cT2 = c.cG1 = c.partial_c = nullptr;
        return;
    }
    for (int l = 0; l < L.n_gcn; ++l) {
        const int fi = L.dims[l], fo = L.dims[l + 1];
        c.zw[l] = fi > fo ? fo : fi;
        c.Z[l] = cv.take<float>(rows * c.zw[l]);
        c.H[l] = cv.take<float>(rows * fo);
        c.M[l] = (fo % 4 == 0 && l + 1 < L.n_gcn) ? cv.take<uint8_t>(rows * fo / 4) : nullptr;
        maxd = std::max(maxd, std::max(fi, fo));
        maxkn = std::max<int64_t>(maxkn, (int64_t)(fi + 1) * fo);
    }
    c.X0 = (L.dims[0] > L.dims[1]) ? cv.take<float>(rows * L.dims[0]) : nullptr;
    c.bufA = cv.take<float>(rows * maxd);
    c.bufB = cv.take<float>(rows * maxd);
    c.partial = cv.take<float>((int64_t)c.b->n_chunks * maxkn);
    for (int l = 1; l < L.n_gcn; ++l) c.partial_l[l] = cv.take<float>((int64_t)c.b->n_chunks * (L.dims[l] + 1) * L.dims[l + 1]);
    for (int l = 1; l < L.n_gcn; ++l) c.WTl[l] = cv.take<float>((int64_t)c.b->sets * L.dims[l] * L.dims[l + 1]);
    c.Wsplit = cv.take<uint16_t>((int64_t)c.b->sets * 3 * maxd * maxd);
    c.cG2 = cv.take<float>((int64_t)c.b->n_c * maxd); c.cT2 = cv.take<float>((int64_t)c.b->n_c * maxd);
    c.cG1 = cv.take<float>((int64_t)c.b->n_e1 * maxd);
    c.partial_c = cv.take<float>((int64_t)std::max(c.b->n_c_chunks, c.b->n_e1_chunks) * maxkn);
}

extern "C" int64_t gm_gcn_ws_bytes(const gm_batch_t* b, const gm_model_t* m) {
    GcnCtx c{}; c.b = b;
    if (!b || gm_make_layout(m, &c.L) != GM_OK) return -1;
    Carver cv(nullptr, 0);
    gcn_carve(c, cv);
    return cv.used + 256;
}

static HeadK make_head(const GcnCtx& c, const float* params, int64_t pstride) {
    const gm_layout& L = c.L; const gm_batch* b = c.b;
    HeadK k{};
    k.H = c.H[L.n_gcn - 1]; k.ldh = L.dims[L.n_gcn]; k.Hd = L.dims[L.n_gcn];
    k.sub_off = b->d_sub_off; k.centre = c.centre ? c.centre : b->d_centre; k.nc = b->centres; k.sub_set = b->d_sub_set;
    k.set_sub_off = b->d_set_sub_off; k.params = params; k.pstride = pstride; k.wl_off = L.wl_off; k.bl_off = L.bl_off;
    k.hc = L.hc; k.C = L.n_out; k.subs = b->subs; k.compact = c.cone ? 1 : 0;
    k.dq_amax = (c.np == 2 && c.am_pass >= 0) ? c.amdQ() : nullptr;
    k.subs_per_set = b->sets > 0 ? b->subs / b->sets : 0;
    for (int t = 0; t <= b->sets && k.subs_per_set; ++t) if (b->h_set_sub_off[t] != t * k.subs_per_set) k.subs_per_set = 0;
    return k;
}

int gm_gather_rows(const gm_store* store, const int32_t* feat_row, int64_t n, int F, float* out, hipStream_t st);
static int cone_forward(GcnCtx& c, const float* params, int64_t pstride, float* logits, hipStream_t st, int reuse_z1, int skip_head);
static int cone_backward(GcnCtx& c, const float* params, int64_t pstride, const float* dlogits, float* dparams, int64_t dstride, hipStream_t st, int skip_head);

// skip_head: the head (centre gather + linear) is evaluated by the fused k_head_loss launch that follows
// fwd_only: nobody differentiates this pass (query evaluations of the inner steps, finetunning's query passes): aggregate-first layers
// whose update runs on the split-bf16 kernel take the FUSED aggregate + GEMM -- the aggregate of a row with one or two sources is formed
// in the GEMM's A feeders (same fma order), only rows of other degrees go through the aggregate kernel and HBM; Z_l of the other
// rows and the relu' bits are never written.  Same floats as the unfused pass, bit for bit.
static std::atomic<int> g_fuse_agg_override{-1};
extern "C" void gm_set_fuse_agg(int32_t on) { g_fuse_agg_override.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); }
extern "C" int32_t gm_get_fuse_agg(void) { const int o = g_fuse_agg_override.load(std::memory_order_relaxed); return o >= 0 ? o : gm_knob().fuse_agg; }

static int gcn_forward(GcnCtx& c, const float* params, int64_t pstride, float* logits, hipStream_t st, int reuse_z1, int skip_head = 0, int fwd_only = 0) {
    if (c.cone) return cone_forward(c, params, pstride, logits, st, reuse_z1, skip_head);
    const gm_layout& L = c.L; const gm_batch* b = c.b;
    GM_REQUIRE(L.dims[0] == b->store->feat_dim || L.dims[0] == b->store->feat_ld || c.x0_user, GM_EINVAL, "forward: dims[0]=%d but the store has %d features", L.dims[0], b->store->feat_dim);
    GM_REQUIRE((L.link != 0) == (b->centres == 2), GM_EINVAL, "forward: link_pred model needs a 2-centre batch and vice versa");
    const float* xin = c.x0_user;           // NULL = gather rows of the store through feat_row
    if (c.np == 2) {                        // a new pass: its own bound slots
        ++c.am_pass;
        GM_REQUIRE(c.am_pass < c.am_passes, GM_EINVAL, "forward: more passes than bound slots (%d)", c.am_passes);
        for (int l = 0; l < GM_MAX_GCN; ++l) c.hv[l] = c.tv[l] = false;
        c.dqv = false;
    }
    for (int l = 0; l < L.n_gcn; ++l) {
        const int fi = L.dims[l], fo = L.dims[l + 1];
        const bool gather = (l == 0 && !c.x0_user);
        if (fi > fo) {                      // learner.py:34-40: multiply first, then aggregate
            const float* A = xin; int64_t lda = fi;
            if (gather) { GM_TRY(gm_gather_rows(b->store, b->d_feat_row, b->rows, fi, c.X0, st)); A = c.X0; }
            gm_gemm_args g{}; g.A = A; g.lda = lda; g.B = params + L.w_off[l]; g.b_stride = pstride; g.C = c.Z[l]; g.ldc = fo; g.K = fi; g.N = fo;
            g.row_scale = b->d_norm; g.tiles = b->d_tiles; g.n_tiles = b->n_tiles; g.rows = b->rows;
            GM_TRY(gm_launch_gemm_nn(g, st));
            gm_agg_args a{}; a.indptr = b->d_indptr; a.indices = b->d_indices; a.heavy = b->d_heavy[0]; a.n_heavy = b->n_heavy[0]; a.heavy_deg = b->heavy_deg; a.sched = b->d_sched[0]; a.sched_len = b->sched_len[0]; a.sched_win = b->sched_win; GM_TRY(gm_agg_hub(a, b, 0, st, c.hub_set)); a.x = c.Z[l]; a.ldx = fo; a.s_out = b->d_norm;
            a.bias = params + L.b_off[l]; a.bias_stride = pstride; a.set_row_off = b->d_set_row_off; a.n_sets = b->sets; a.relu = 1;
            a.out = c.H[l]; a.rows = b->rows; a.width = fo; a.relu_bits = c.M[l];
            gm_prof_agg_begin(st, gm_aggregate_bytes(b, fo)); gm_prof_note(GM_PROF_AGG_STRICT, gm_aggregate_bytes(b, fo));
            GM_TRY(gm_launch_aggregate(a, st));
            gm_prof_agg_end(st);
        } else {                            // learner.py:41-47: aggregate first, then multiply
            const bool split_ok = c.Wsplit && gm_gemm_split_ok(b->n_tiles, fi, fo) && ((uintptr_t)(params + L.b_off[l]) & 15) == 0 && pstride % 4 == 0;
            // ... and, round 6, the passes that ARE differentiated by the dense backward (fwd_only == 2: the support passes, the last query pass): their only
            // other reader of Z_l is the weight gradient, whose split kernel forms the same rows from the same table (gm_wgrad_gather_ok; three-piece
            // arithmetic only) -- the differentiated passes no longer write and re-read Z_l either (GM_FUSE_DIFF=0: as before)
            // GM_FUSE_DIFF=2: ... except where the pass's full launches would take the stream aggregate (a support batch with stream tables, agg_stream.hip):
            // that kernel runs INSIDE the CUs the query stream's GEMM occupies, a partial window launch competes with it for them
            const bool keeps_stream = gm_knob().fuse_diff == 2 && c.is_support && gm_knob().agg_stream && gm_stream_batch_ok(b, 0);
            const bool diff_ok = fwd_only == 2 && gm_knob().fuse_diff && !keeps_stream && c.np != 2 && !c.cone && gm_wgrad_gather_ok(b->n_chunks, fi, fo);
            const bool fuse = (fwd_only == 1 || diff_ok) && gm_get_fuse_agg() && split_ok && !(l == 0 && reuse_z1) && fi >= 64 && fi % 4 == 0 && b->d_fuse2 && b->d_enorm[0] &&
                              (!gather || (b->store->feat_ld % 4 == 0 && b->store->feat_ld >= fi)) &&
                              // worth it only where a good part of the rows has one or two sources: on dense batches (Tissue shape: ~30 in-edges per
                              // row) nearly every row still goes through the ordinary aggregate and the gather feeders only cost (3.30 -> 3.21 ms)
                              2 * b->unfused_rows <= b->rows;
            c.zfused[l] = fuse && fwd_only == 2;
            if (!(l == 0 && reuse_z1 && c.z1_valid)) {
                gm_agg_args a{}; a.indptr = b->d_indptr; a.indices = b->d_indices; a.heavy = b->d_heavy[0]; a.n_heavy = b->n_heavy[0]; a.heavy_deg = b->heavy_deg; a.sched = b->d_sched[0]; a.sched_len = b->sched_len[0]; a.sched_win = b->sched_win; GM_TRY(gm_agg_hub(a, b, 0, st, c.hub_set)); a.s_in = b->d_norm; a.e_w = b->d_enorm[0]; a.out = c.Z[l]; a.rows = b->rows; a.width = fi;
                if (gather) { a.x = b->store->d_feat; a.x_row = b->d_feat_row; a.x_idx = b->d_efeat; a.ldx = b->store->feat_ld; }
                else { a.x = xin; a.ldx = fi; }
                if (fuse) {
                    // only the rows the fused kernel does not form itself (more than GM_FUSE_MAXDEG sources): a partial launch
                    a.skip_on = 1; a.skip_lo = 0; a.skip_hi = GM_FUSE_MAXDEG;
                    if (b->d_mid && c.hub_set == 0 && gm_knob().agg_mid_list && (b->n_heavy[0] == 0 || b->d_sched_mid)) {
                        // ... walking the compact list of those rows (hub rows keep their blocks; the schedule is the list's)
                        a.rowlist = b->d_mid; a.n_list = b->n_mid; a.list_win = b->mid_win;
                        if (b->n_heavy[0] > 0) { a.sched = b->d_sched_mid; a.sched_len = b->sched_len_mid; a.sched_win = b->mid_win; }
                        else { a.sched = nullptr; a.sched_len = 0; }
                    }
                    // SURVEY 8(d)'s B_agg restricted to what this launch touches: every indptr entry, the indices / norms of the rows it writes,
                    // those rows (written once) and their sources (read once: the high-degree rows of a subgraph reach ~all of its rows)
                    const int64_t pb0 = 4 * (b->rows + 1) + 4 * b->unfused_edges + 4 * b->unfused_rows + 4 * b->unfused_rows * (int64_t)fi;
                    // (their DISTINCT sources when the launch accounting is on: counted once per batch; the bound min(edges, rows) otherwise -- nobody reads it then)
                    const int64_t src = gm_prof_enabled() ? gm_batch_unfused_sources(b, st) : std::min<int64_t>(b->unfused_edges, b->rows);
                    const int64_t pb = pb0 + 4 * src * (int64_t)fi;
                    // strict HBM pricing as for the full launches: a gather launch reads at most the whole (cache-resident) feature table
                    const int64_t pbs = pb0 + 4 * (gather ? std::min<int64_t>(src, b->store->total_nodes) : src) * (int64_t)fi;
                    gm_prof_agg_begin(st, pb); gm_prof_note(GM_PROF_AGG_STRICT, pbs);
                    gm_prof_note(GM_PROF_AGG_BOUND, pb0 + 4 * std::min<int64_t>(b->unfused_edges, b->rows) * (int64_t)fi - pb);      // (difference to the exact count)
                    GM_TRY(gm_launch_aggregate(a, st));
                    gm_prof_agg_end(st);
                } else {
                    if (a.e_w) GM_TRY(gm_agg_stream_args(a, b, 0, gather, st));      // full launches may take the LDS-DMA stream kernel (agg_stream.hip)
                    gm_prof_agg_begin(st, gm_aggregate_bytes(b, fi));
                    {   // compulsory HBM bytes: a gather launch reads rows of the (cache-resident) feature table, at most all of it
                        int64_t strict = gm_aggregate_bytes(b, fi);
                        if (gather) strict += 4 * b->rows - 4 * b->rows * (int64_t)fi + std::min<int64_t>(4 * b->rows * (int64_t)fi, 4 * b->store->total_nodes * (int64_t)fi);
                        gm_prof_note(GM_PROF_AGG_STRICT, strict);
                    }
                    GM_TRY(gm_launch_aggregate(a, st));
                    gm_prof_agg_end(st);
                    if (l == 0) c.z1_valid = 1;
                }
            }
            gm_gemm_args g{}; g.A = c.Z[l]; g.lda = fi; g.B = params + L.w_off[l]; g.b_stride = pstride; g.C = c.H[l]; g.ldc = fo; g.K = fi; g.N = fo;
            g.row_scale = b->d_norm; g.bias = params + L.b_off[l]; g.bias_stride = pstride; g.relu = 1; g.tiles = b->d_tiles; g.n_tiles = b->n_tiles; g.rows = b->rows;
            g.relu_bits = fwd_only == 1 ? nullptr : c.M[l];
            // last layer: only the head reads H_L, and only its centre rows (h[to_fetch]; the backward pass takes relu' from the bits and the
            // weight gradient from Z_L): the other rows are computed, their relu' bits written, their values not stored
            if (l == L.n_gcn - 1 && !c.centre && b->d_norm_c && (gm_knob().centre_store >= 2 || (gm_knob().centre_store == 1 && fwd_only == 1))) { g.row_scale_keep = b->d_norm_c; g.n_keep = b->n_c; }
            if (split_ok) {
                gm_bound ab = gm_no_bound(), bb;
                const bool want16 = in_bound(c, l, ab);
                const uint16_t* pl = nullptr;                                                 // left by the reduction that wrote these weights, else split now
                int np = 3;
                GM_TRY(weight_planes(c, params, pstride, l, 0, fi, fo, want16, st, &pl, &bb, &np));
                g.Bsplit = pl; g.bsplit_stride = pstride ? (int64_t)3 * fi * fo : 0;
                g.np = np; g.a_bound = ab; g.b_bound = bb;
                if (np == 2) { g.amax_out = c.amH(l); c.hv[l] = true; }        // (the two-piece kernels record it)
            }
            // fwd_only == 2: the head + loss + backward follow (gm_meta_step): the last layer's GEMM zero-fills dQ on its way out instead of a memset launch
            if (fwd_only == 2 && l == L.n_gcn - 1 && fo == L.dims[L.n_gcn]) { g.zero_out = c.bufA; c.dq_zeroed = 1; }
            if (fuse) {
                g.zside = c.Z[l]; g.ldz = fi;
                if (gather) { g.A = b->store->d_feat; g.lda = b->store->feat_ld; g.fuse2 = b->d_fuse2_feat; }
                else { g.A = xin; g.lda = fi; g.fuse2 = b->d_fuse2; }
            }
            GM_TRY(gm_launch_gemm_nn(g, st));
        }
        xin = c.H[l];
    }
    if (skip_head) return GM_OK;
    HeadK k = make_head(c, params, pstride);
    hipLaunchKernelGGL(k_head_fwd, dim3((b->subs + 3) / 4), dim3(256), 0, st, k, logits);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

static int gcn_backward_sparse(GcnCtx& c, const float* params, int64_t pstride, const float* dlogits, float* dparams, int64_t dstride, hipStream_t st, int skip_head);
static bool sparse_bwd_ok(const gm_layout& L);

// skip_head: dQ_L / the compact G2 and the head's own gradients were already produced by k_head_loss (head_loss below)
static int gcn_backward(GcnCtx& c, const float* params, int64_t pstride, const float* dlogits, float* dparams, int64_t dstride, hipStream_t st, int sparse = 0,
                        int skip_head = 0) {
    if (c.cone) return cone_backward(c, params, pstride, dlogits, dparams, dstride, st, skip_head);
    if (sparse && sparse_bwd_ok(c.L)) return gcn_backward_sparse(c, params, pstride, dlogits, dparams, dstride, st, skip_head);
    const gm_layout& L = c.L; const gm_batch* b = c.b;
    const int Lg = L.n_gcn;
    float* dQ = c.bufA; float* T = c.bufB;
    c.hold.n = 0;
    if (!skip_head) {
        GM_HIP(hipMemsetAsync(dQ, 0, sizeof(float) * b->rows * L.dims[Lg], st));
        HeadK k = make_head(c, params, pstride);
        hipLaunchKernelGGL(k_head_bwd, dim3(b->sets), dim3(256), 0, st, k, dlogits, dparams, dstride, dQ, (float*)nullptr);
        GM_HIP(hipGetLastError());
        if (k.dq_amax) c.dqv = true;
    }
    for (int l = Lg - 1; l >= 0; --l) {
        const int fi = L.dims[l], fo = L.dims[l + 1];
        const float* Xprev = l > 0 ? c.H[l - 1] : (c.x0_user ? c.x0_user : c.X0);
        const float* maskprev = l > 0 ? c.H[l - 1] : nullptr;
        const uint8_t* maskbits = l > 0 ? c.M[l - 1] : nullptr;           // packed relu'(H_{l-1}): 1/16 of the bytes of H_{l-1}
        if (maskbits) maskprev = nullptr;
        gm_wgrad_args w{}; w.chunks = b->d_chunks; w.n_chunks = b->n_chunks; w.set_chunk_off = b->d_set_chunk_off; w.sets = b->sets; w.rows = b->rows;
        w.partial = c.partial; w.dW = dparams + L.w_off[l]; w.dw_stride = dstride; w.db = dparams + L.b_off[l]; w.db_stride = dstride;
        w.a_scale = b->d_norm; w.K = fi; w.N = fo;
        wgrad_sgd(w, c, l);
        // the reduction (+ SGD step, weight planes) of a layer above the first is held back and launched with the first layer's: the new
        // W_l has no reader before the next forward
        w.hold = &c.hold; w.hold_this = l > 0 ? 1 : 0;
        if (l > 0) w.partial = c.partial_l[l];
        if (fi > fo) {
            // dY = A^T (norm * dQ) ; dW = (norm*X)^T dY ; db = colsum(dQ) ; dQ_prev = relu'(H_prev) * norm * (dY W^T)
            gm_agg_args a{}; a.indptr = b->d_indptr_t; a.indices = b->d_indices_t; a.heavy = b->d_heavy[1]; a.n_heavy = b->n_heavy[1]; a.heavy_deg = b->heavy_deg; a.sched = b->d_sched[1]; a.sched_len = b->sched_len[1]; a.sched_win = b->sched_win; GM_TRY(gm_agg_hub(a, b, 1, st, c.hub_set)); a.x = dQ; a.ldx = fo; a.s_in = b->d_norm; a.e_w = b->d_enorm[1]; a.out = T; a.rows = b->rows; a.width = fo;
            gm_prof_agg_begin(st, gm_aggregate_bytes(b, fo)); gm_prof_note(GM_PROF_AGG_STRICT, gm_aggregate_bytes(b, fo));
            GM_TRY(gm_launch_aggregate(a, st));
            gm_prof_agg_end(st);
            w.A = Xprev; w.lda = fi; w.G = T; w.ldg = fo; w.Gb = dQ; w.ldgb = fo;
            GM_TRY(gm_launch_wgrad(w, st));
            if (l > 0) {
                gm_gemm_args g{}; g.A = T; g.lda = fo; g.B = params + L.w_off[l]; g.b_stride = pstride; g.transB = 1; g.C = dQ; g.ldc = fi; g.K = fo; g.N = fi;
                g.row_scale = b->d_norm; g.mask_h = maskprev; g.mask_b = maskbits; g.tiles = b->d_tiles; g.n_tiles = b->n_tiles; g.rows = b->rows;
                GM_TRY(gm_launch_gemm_nn(g, st));
            }
        } else {
            // dW = (norm*Z)^T dQ ; db = colsum(dQ) ; dZ = norm * (dQ W^T) ; dQ_prev = relu'(H_prev) * norm * A^T dZ
            // Order: dZ GEMM (reads dQ and the CURRENT weights) -> weight gradient (reads dQ; its reduction writes the updated
            // weights and, for the next step's dZ GEMM, their transpose) -> transposed aggregate (overwrites dQ).
            w.A = c.Z[l]; w.lda = fi; w.G = dQ; w.ldg = fo;
            if (c.zfused[l]) {                 // the forward ran fused: Z[l] holds the rows of three or more sources, the table forms the others
                const bool gather = l == 0 && !c.x0_user;
                w.fuse2 = gather ? b->d_fuse2_feat : b->d_fuse2;
                w.gx = gather ? b->store->d_feat : (l > 0 ? c.H[l - 1] : c.x0_user); w.ldgx = gather ? b->store->feat_ld : fi;
            }
            if (l > 0) {
                gm_gemm_args g{}; g.A = dQ; g.lda = fo; g.C = T; g.ldc = fi; g.K = fo; g.N = fi;
                g.row_scale = b->d_norm; g.tiles = b->d_tiles; g.n_tiles = b->n_tiles; g.rows = b->rows;
                const int dz_glds = gm_knob().dz_glds;
                const bool use_split = c.Wsplit && gm_gemm_split_ok(b->n_tiles, fo, fi);
                const bool use_wt = !use_split && dz_glds && c.WTl[l] && fi % 64 == 0 && fo % 16 == 0;
                if (use_split) {
                    // B = W^T with W stored [fi][fo]: the planes are W's own rows (no transpose), K = fo, N = fi
                    gm_bound ab = gm_no_bound(), bb;
                    const bool want16 = dq_bound(c, l, ab);
                    const uint16_t* pl = nullptr; int np = 3;
                    GM_TRY(weight_planes(c, params, pstride, l, 1, fo, fi, want16, st, &pl, &bb, &np));
                    g.Bsplit = pl; g.bsplit_stride = pstride ? (int64_t)3 * fi * fo : 0;
                    g.np = np; g.a_bound = ab; g.b_bound = bb;
                    if (np == 2) { g.amax_out = c.amT(l); c.tv[l] = true; }
                    g.B = params + L.w_off[l]; g.b_stride = pstride; g.transB = 1;
                } else if (use_wt) {
                    // dZ = dQ @ W^T through the direct-to-LDS kernel on transposed weights: left there by the previous step's
                    // weight-gradient reduction (which wrote these very weights), else transposed now (T x 256 KB)
                    if (!(c.wt_of[l] == params && c.wt_stride[l] == pstride)) {
                        const int nt = pstride ? b->sets : 1;         // shared theta (step 0): one transpose serves every task
                        hipLaunchKernelGGL(k_transpose_w, dim3((fo + 31) / 32, (fi + 31) / 32, nt), dim3(256), 0, st, params, pstride, L.w_off[l], fi, fo, c.WTl[l]);
                        GM_HIP(hipGetLastError());
                        c.wt_of[l] = params; c.wt_stride[l] = pstride;
                    }
                    g.B = c.WTl[l]; g.b_stride = pstride ? (int64_t)fi * fo : 0; g.transB = 0;
                } else { g.B = params + L.w_off[l]; g.b_stride = pstride; g.transB = 1; }
                GM_TRY(gm_launch_gemm_nn(g, st));
                if (use_wt && c.sgd.next) { w.wt_next = c.WTl[l]; }
            }
            int kn = 0;
            if (c.pd && c.sgd.next && (kn = c.pd->index_of(c.sgd.next)) != 0) { w.pl_fwd = c.pd->slot(kn, l, 0); w.pl_dz = c.pd->slot(kn, l, 1); }
            if (c.np == 2) {
                gm_bound ab, gb;
                if (in_bound(c, l, ab) && dq_bound(c, l, gb)) { w.np = 2; w.a_bound = ab; w.g_bound = gb; }
                if (kn) {                                                  // two-piece planes of the updated weights, under the step's weight bound
                    if (c.pd->w_bound(c.sgd.next, l, w.pl_bound)) w.pl_np = 2; else { w.pl_fwd = nullptr; w.pl_dz = nullptr; }
                }
            }
            GM_TRY(gm_launch_wgrad(w, st));
            if (kn) { c.pd->valid[kn][l][0] = w.pl_fwd != nullptr; c.pd->valid[kn][l][1] = w.pl_dz != nullptr; }
            if (w.wt_next) { c.wt_of[l] = c.sgd.next; c.wt_stride[l] = c.sgd.next_stride; }
            if (l > 0) {
                gm_agg_args a{}; a.indptr = b->d_indptr_t; a.indices = b->d_indices_t; a.heavy = b->d_heavy[1]; a.n_heavy = b->n_heavy[1]; a.heavy_deg = b->heavy_deg; a.sched = b->d_sched[1]; a.sched_len = b->sched_len[1]; a.sched_win = b->sched_win; GM_TRY(gm_agg_hub(a, b, 1, st, c.hub_set)); a.x = T; a.ldx = fi; a.s_out = b->d_norm; a.mask_h = maskprev; a.mask_b = maskbits;
                a.out = dQ; a.rows = b->rows; a.width = fi;
                gm_prof_agg_begin(st, gm_aggregate_bytes(b, fi)); gm_prof_note(GM_PROF_AGG_STRICT, gm_aggregate_bytes(b, fi));
                GM_TRY(gm_launch_aggregate(a, st));
                gm_prof_agg_end(st);
            }
        }
    }
    return GM_OK;
}

static bool sparse_bwd_ok(const gm_layout& L) {
    if (L.n_gcn > 2) return false;
    for (int l = 0; l < L.n_gcn; ++l) if (L.dims[l] > L.dims[l + 1]) return false;      // aggregate-first layers only
    return true;
}

// Exact row-sparse backward (see gm_hparams_t.sparse_bwd).  The head is the only consumer of the last GCN layer, so
// dQ_L lives on the centre rows; one transposed-aggregate step spreads it along the in-edges of the centres.  Every
// product below is the dense backward's product with the structurally-zero terms removed.
static int gcn_backward_sparse(GcnCtx& c, const float* params, int64_t pstride, const float* dlogits, float* dparams, int64_t dstride, hipStream_t st, int skip_head) {
    const gm_layout& L = c.L; const gm_batch* b = c.b;
    const int Lg = L.n_gcn, fiL = L.dims[Lg - 1], foL = L.dims[Lg];
    if (!skip_head) {
        HeadK k = make_head(c, params, pstride);
        hipLaunchKernelGGL(k_head_bwd, dim3(b->sets), dim3(256), 0, st, k, dlogits, dparams, dstride, (float*)nullptr, c.cG2);
        GM_HIP(hipGetLastError());
    }
    // dW_L = sum_k norm[c_k] Z_L[c_k]^T G2[k] ; db_L = sum_k G2[k]
    gm_wgrad_args w{}; w.A = c.Z[Lg - 1]; w.lda = fiL; w.K = fiL; w.a_row = b->d_crow; w.a_scale = b->d_cnorm; w.G = c.cG2; w.ldg = foL; w.N = foL;
    w.rows = b->n_c; w.chunks = b->d_c_chunks; w.n_chunks = b->n_c_chunks; w.set_chunk_off = b->d_c_set_chunk_off; w.sets = b->sets; w.partial = c.partial_c;
    w.dW = dparams + L.w_off[Lg - 1]; w.dw_stride = dstride; w.db = dparams + L.b_off[Lg - 1]; w.db_stride = dstride;
    wgrad_sgd(w, c, Lg - 1);
    GM_TRY(gm_launch_wgrad(w, st));
    if (Lg == 1) return GM_OK;
    // T2[k] = norm[c_k] * (G2[k] W_L^T)
    gm_gemm_args g{}; g.A = c.cG2; g.lda = foL; g.B = params + L.w_off[Lg - 1]; g.b_stride = pstride; g.transB = 1; g.C = c.cT2; g.ldc = fiL; g.K = foL; g.N = fiL;
    g.row_scale = b->d_cnorm; g.tiles = b->d_c_tiles; g.n_tiles = b->n_c_tiles; g.rows = b->n_c;
    GM_TRY(gm_launch_gemm_nn(g, st));
    if (b->n_e1 > 0) {
        const int64_t tot = (int64_t)b->n_e1 * fiL;
        hipLaunchKernelGGL(k_expand_edges, dim3((int)std::min<int64_t>(2048, (tot + 255) / 256)), dim3(256), 0, st, c.cT2, c.H[0], fiL, b->d_e1_row, b->d_e1_par,
                           b->d_e1_norm, b->n_e1, c.cG1);
        GM_HIP(hipGetLastError());
    }
    // dW_1 = sum_e norm[u_e] Z_1[u_e]^T G1[e] ; db_1 = sum_e G1[e]   (a set without any centre in-edge gets zeros: empty chunk range)
    const int f0 = L.dims[0];
    gm_wgrad_args w1{}; w1.A = c.Z[0]; w1.lda = f0; w1.K = f0; w1.a_row = b->d_e1_row; w1.a_scale = b->d_e1_norm; w1.G = c.cG1; w1.ldg = fiL; w1.N = fiL;
    w1.rows = b->n_e1; w1.chunks = b->d_e1_chunks; w1.n_chunks = b->n_e1_chunks; w1.set_chunk_off = b->d_e1_set_chunk_off; w1.sets = b->sets; w1.partial = c.partial_c;
    w1.dW = dparams + L.w_off[0]; w1.dw_stride = dstride; w1.db = dparams + L.b_off[0]; w1.db_stride = dstride;
    wgrad_sgd(w1, c, 0);
    GM_TRY(gm_launch_wgrad(w1, st));
    return GM_OK;
}


// ================================================================================ receptive-field ("cone") schedule
// The same layer formulas as gcn_forward/gcn_backward, evaluated only on the rows that can reach a centre
// (cone.hip): layer l maps level l (sources) to level l+1 (destinations); all matrices are compact.

// SURVEY 8(d)'s B_agg on the rows a level-to-level aggregate actually touches: the destination level's row bounds + norms, the edges between the
// two levels, every source-level row read once (by construction each of them is the source of at least one of those edges), every
// destination row written once
static int64_t cone_agg_bytes(int64_t n_dst, int64_t n_src, int64_t nnz, int width) {
    return 4 * (n_dst + 1) + 4 * nnz + 4 * n_dst + 4 * (n_src + n_dst) * (int64_t)width;
}
static int cone_launch_agg(const gm_agg_args& a, int64_t n_src, int64_t nnz, hipStream_t st) {
    const int64_t by = cone_agg_bytes(a.rows, n_src, nnz, a.width);
    gm_prof_agg_begin(st, by); gm_prof_note(GM_PROF_AGG_STRICT, by);
    const int rc = gm_launch_aggregate(a, st);
    gm_prof_agg_end(st);
    return rc;
}
static gm_agg_args cone_agg(const gm_cone* cn, const gm_cone_level& up, int transposed) {
    gm_agg_args a{};
    a.indptr = transposed ? up.d_indptr_t : up.d_indptr; a.indices = transposed ? up.d_indices_t : up.d_indices;
    a.heavy = up.d_heavy[transposed]; a.n_heavy = up.n_heavy[transposed]; a.heavy_deg = cn->heavy_deg;
    return a;
}

static int cone_forward(GcnCtx& c, const float* params, int64_t pstride, float* logits, hipStream_t st, int reuse_z1, int skip_head) {
    const gm_layout& L = c.L; const gm_batch* b = c.b; const gm_cone* cn = c.cone;
    GM_REQUIRE(L.dims[0] == b->store->feat_dim || L.dims[0] == b->store->feat_ld, GM_EINVAL, "forward: dims[0]=%d but the store has %d features", L.dims[0], b->store->feat_dim);
    GM_REQUIRE((L.link != 0) == (b->centres == 2), GM_EINVAL, "forward: link_pred model needs a 2-centre batch and vice versa");
    GM_REQUIRE(!c.x0_user && !c.centre, GM_EINVAL, "forward: the cone schedule reads features and centres from the batch");
    const float* xin = nullptr;
    for (int l = 0; l < L.n_gcn; ++l) {
        const gm_cone_level& lo = cn->lv[l]; const gm_cone_level& up = cn->lv[l + 1];
        const int fi = L.dims[l], fo = L.dims[l + 1];
        if (fi > fo) {                      // multiply on the source level, then aggregate into the destination level
            const float* A = xin;
            if (l == 0) { GM_TRY(gm_gather_rows(b->store, lo.d_feat_row, lo.n, fi, c.X0, st)); A = c.X0; }
            gm_gemm_args g{}; g.A = A; g.lda = fi; g.B = params + L.w_off[l]; g.b_stride = pstride; g.C = c.Z[l]; g.ldc = fo; g.K = fi; g.N = fo;
            g.row_scale = lo.d_norm; g.tiles = lo.d_tiles; g.n_tiles = lo.n_tiles; g.rows = lo.n;
            GM_TRY(gm_launch_gemm_nn(g, st));
            gm_agg_args a = cone_agg(cn, up, 0);
            a.x = c.Z[l]; a.ldx = fo; a.s_out = up.d_norm; a.bias = params + L.b_off[l]; a.bias_stride = pstride; a.set_row_off = up.d_set_off; a.n_sets = b->sets;
            a.relu = 1; a.out = c.H[l]; a.rows = up.n; a.width = fo;
            GM_TRY(cone_launch_agg(a, lo.n, up.nnz, st));
        } else {
            if (!(l == 0 && reuse_z1 && c.z1_valid)) {
                gm_agg_args a = cone_agg(cn, up, 0);
                a.s_in = lo.d_norm; a.out = c.Z[l]; a.rows = up.n; a.width = fi; a.ldx = fi;
                if (l == 0) { a.x = b->store->d_feat; a.x_row = lo.d_feat_row; a.ldx = b->store->feat_ld; } else a.x = xin;
                GM_TRY(cone_launch_agg(a, lo.n, up.nnz, st));
                if (l == 0) c.z1_valid = 1;
            }
            gm_gemm_args g{}; g.A = c.Z[l]; g.lda = fi; g.B = params + L.w_off[l]; g.b_stride = pstride; g.C = c.H[l]; g.ldc = fo; g.K = fi; g.N = fo;
            g.row_scale = up.d_norm; g.bias = params + L.b_off[l]; g.bias_stride = pstride; g.relu = 1; g.tiles = up.d_tiles; g.n_tiles = up.n_tiles; g.rows = up.n;
            GM_TRY(gm_launch_gemm_nn(g, st));
        }
        xin = c.H[l];
    }
    if (skip_head) return GM_OK;
    HeadK k = make_head(c, params, pstride);
    hipLaunchKernelGGL(k_head_fwd, dim3((b->subs + 3) / 4), dim3(256), 0, st, k, logits);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

static int cone_backward(GcnCtx& c, const float* params, int64_t pstride, const float* dlogits, float* dparams, int64_t dstride, hipStream_t st, int skip_head) {
    const gm_layout& L = c.L; const gm_batch* b = c.b; const gm_cone* cn = c.cone;
    const int Lg = L.n_gcn;
    float* dQ = c.bufA; float* T = c.bufB;
    if (!skip_head) {
        HeadK k = make_head(c, params, pstride);
        hipLaunchKernelGGL(k_head_bwd, dim3(b->sets), dim3(256), 0, st, k, dlogits, dparams, dstride, (float*)nullptr, dQ);   // dQ_L on the centre rows
        GM_HIP(hipGetLastError());
    }
    for (int l = Lg - 1; l >= 0; --l) {
        const gm_cone_level& lo = cn->lv[l]; const gm_cone_level& up = cn->lv[l + 1];
        const int fi = L.dims[l], fo = L.dims[l + 1];
        const float* maskprev = l > 0 ? c.H[l - 1] : nullptr;
        gm_wgrad_args w{}; w.sets = b->sets; w.partial = c.partial; w.K = fi; w.N = fo;
        w.dW = dparams + L.w_off[l]; w.dw_stride = dstride; w.db = dparams + L.b_off[l]; w.db_stride = dstride;
        wgrad_sgd(w, c, l);
        if (fi > fo) {
            // dY = A^T (norm * dQ) on the source level ; dW = (norm*X)^T dY ; db = colsum(dQ) ; dQ_prev = relu'(H_prev) * norm * (dY W^T)
            gm_agg_args a = cone_agg(cn, up, 1);
            a.x = dQ; a.ldx = fo; a.s_in = up.d_norm; a.out = T; a.rows = lo.n; a.width = fo;
            GM_TRY(cone_launch_agg(a, up.n, up.nnz, st));
            w.A = l > 0 ? c.H[l - 1] : c.X0; w.lda = fi; w.a_scale = lo.d_norm; w.G = T; w.ldg = fo; w.db = nullptr;
            w.rows = lo.n; w.chunks = lo.d_chunks; w.n_chunks = lo.n_chunks; w.set_chunk_off = lo.d_set_chunk_off;
            GM_TRY(gm_launch_wgrad(w, st));
            hipLaunchKernelGGL(k_colsum_rows, dim3(b->sets), dim3(256), 0, st, dQ, (int64_t)fo, fo, up.d_set_off, dparams + L.b_off[l], dstride, c.sgd, L.b_off[l]);
            GM_HIP(hipGetLastError());
            if (l > 0) {
                gm_gemm_args g{}; g.A = T; g.lda = fo; g.B = params + L.w_off[l]; g.b_stride = pstride; g.transB = 1; g.C = dQ; g.ldc = fi; g.K = fo; g.N = fi;
                g.row_scale = lo.d_norm; g.mask_h = maskprev; g.tiles = lo.d_tiles; g.n_tiles = lo.n_tiles; g.rows = lo.n;
                GM_TRY(gm_launch_gemm_nn(g, st));
            }
        } else {
            // dW = (norm*Z)^T dQ ; db = colsum(dQ) ; dZ = norm * (dQ W^T) ; dQ_prev = relu'(H_prev) * norm * A^T dZ
            w.A = c.Z[l]; w.lda = fi; w.a_scale = up.d_norm; w.G = dQ; w.ldg = fo;
            w.rows = up.n; w.chunks = up.d_chunks; w.n_chunks = up.n_chunks; w.set_chunk_off = up.d_set_chunk_off;
            GM_TRY(gm_launch_wgrad(w, st));
            if (l > 0) {
                gm_gemm_args g{}; g.A = dQ; g.lda = fo; g.B = params + L.w_off[l]; g.b_stride = pstride; g.transB = 1; g.C = T; g.ldc = fi; g.K = fo; g.N = fi;
                g.row_scale = up.d_norm; g.tiles = up.d_tiles; g.n_tiles = up.n_tiles; g.rows = up.n;
                GM_TRY(gm_launch_gemm_nn(g, st));
                gm_agg_args a = cone_agg(cn, up, 1);
                a.x = T; a.ldx = fi; a.s_out = lo.d_norm; a.mask_h = maskprev; a.out = dQ; a.rows = lo.n; a.width = fi;
                GM_TRY(cone_launch_agg(a, up.n, up.nnz, st));
            }
        }
    }
    return GM_OK;
}

extern "C" int gm_gcn_forward(const gm_batch_t* b, const gm_model_t* m, const float* params, int64_t param_stride, const float* x0,
                              const int32_t* centre_local, float* logits, void* ws, int64_t ws_bytes, void* stream) {
    GM_REQUIRE(b && m && params && logits && ws, GM_EINVAL, "gcn_forward: NULL argument");
    GcnCtx c{}; c.b = b;
    GM_TRY(gm_make_layout(m, &c.L));
    Carver cv(ws, ws_bytes);
    gcn_carve(c, cv);
    GM_REQUIRE(cv.ok(), GM_ENOMEM, "gcn_forward: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)cv.used);
    c.x0_user = x0; c.centre = centre_local;
    const int rc = gcn_forward(c, params, param_stride, logits, (hipStream_t)stream, 0);
    gm_batch_mark_use(b, (hipStream_t)stream);
    return rc;
}

extern "C" int gm_gcn_backward(const gm_batch_t* b, const gm_model_t* m, const float* params, int64_t param_stride, const float* x0,
                               const int32_t* centre_local, const float* dlogits, float* dparams, int64_t dparam_stride, void* ws,
                               int64_t ws_bytes, void* stream) {
    GM_REQUIRE(b && m && params && dlogits && dparams && ws, GM_EINVAL, "gcn_backward: NULL argument");
    GcnCtx c{}; c.b = b;
    GM_TRY(gm_make_layout(m, &c.L));
    GM_REQUIRE(dparam_stride >= c.L.P, GM_EINVAL, "gcn_backward: dparam_stride %lld < P=%lld", (long long)dparam_stride, (long long)c.L.P);
    Carver cv(ws, ws_bytes);
    gcn_carve(c, cv);
    GM_REQUIRE(cv.ok(), GM_ENOMEM, "gcn_backward: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)cv.used);
    c.x0_user = x0; c.centre = centre_local;
    const int rc = gcn_backward(c, params, param_stride, dlogits, dparams, dparam_stride, (hipStream_t)stream);
    gm_batch_mark_use(b, (hipStream_t)stream);
    return rc;
}

// ================================================================================ prototypical losses
// Host side of meta.py:32-42,60-65: sorted unique classes, rows of each class (first n_support for the
// support loss; all rows, equal counts required, for the query loss), as global subgraph ids.  Every set (task) keeps its
// own class layout -- the reference calls proto_loss_* once per task (meta.py:118-157), so a Shared dataset whose graphs
// carry different label sets, or evaluation tasks of different shapes, are fine.
struct ClassTables {
    std::vector<int32_t> rows;          // concatenated per set: [Ct_t][n_t] subgraph ids
    std::vector<int32_t> tab;           // [sets*3]: offset into rows, Ct_t, n_t
    int Ct = 0, n = 0;                  // maxima over the sets (LDS sizing, prototype stride)
    bool uniform = false;               // every set: Ct classes x n rows, stored at set * Ct * n
};
static int class_tables(const gm_batch* b, const int32_t* y, int limit, ClassTables& ct) {
    ct.rows.clear(); ct.tab.assign((size_t)b->sets * 3, 0); ct.Ct = 0; ct.n = 0;
    for (int t = 0; t < b->sets; ++t) {
        std::map<int32_t, std::vector<int32_t>> by;
        for (int s = b->h_set_sub_off[t]; s < b->h_set_sub_off[t + 1]; ++s) by[y[s]].push_back(s);
        GM_REQUIRE(!by.empty(), GM_EINVAL, "proto loss: set %d is empty", t);
        int cnt = -1;
        for (auto& kv : by) {
            int c = (int)kv.second.size();
            if (limit > 0) {
                GM_REQUIRE(c >= limit, GM_EINVAL, "proto loss: class %d of set %d has %d rows < n_support=%d (torch.stack at meta.py:42 fails)", kv.first, t, c, limit);
                c = limit;
            }
            GM_REQUIRE(cnt < 0 || cnt == c, GM_EINVAL, "proto loss: classes of set %d have unequal row counts (torch.stack at meta.py:65 fails)", t);
            cnt = c;
        }
        ct.tab[t * 3] = (int32_t)ct.rows.size(); ct.tab[t * 3 + 1] = (int32_t)by.size(); ct.tab[t * 3 + 2] = cnt;
        ct.Ct = std::max(ct.Ct, (int)by.size()); ct.n = std::max(ct.n, cnt);
        for (auto& kv : by) ct.rows.insert(ct.rows.end(), kv.second.begin(), kv.second.begin() + cnt);
    }
    GM_REQUIRE((int64_t)ct.Ct * ct.n <= 8192 && ct.Ct <= 256, GM_ERANGE, "proto loss: %d classes x %d rows per set is outside the kernel's range", ct.Ct, ct.n);
    ct.uniform = true;
    for (int t = 0; t < b->sets; ++t) ct.uniform = ct.uniform && ct.tab[t * 3] == t * ct.Ct * ct.n && ct.tab[t * 3 + 1] == ct.Ct && ct.tab[t * 3 + 2] == ct.n;
    return GM_OK;
}

static size_t proto_lds(int Ct, int n, int D, int nt = 256) {
    const size_t a = (size_t)Ct * n * Ct;
    return sizeof(float) * ((size_t)Ct * D + (size_t)Ct * n + 2 * (size_t)nt + (a <= PROTO_A_MAX ? a : 0));
}

static int launch_proto(const gm_batch* b, ProtoK k, hipStream_t st) {
    GM_TRY(gm_func_full_lds((const void*)k_proto));
    hipLaunchKernelGGL(k_proto, dim3(b->sets), dim3(256), proto_lds(k.Ct, k.n, k.D), st, k);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// rows + tab of a ClassTables on the device (one allocation: [rows | tab]); freed by the caller with gm_dev_free
static int upload_tables(const ClassTables& ct, int32_t** d_out, hipStream_t st) {
    *d_out = nullptr;
    int32_t* d = nullptr;
    GM_TRY(gm_alloc(&d, ct.rows.size() + ct.tab.size(), st));
    if (hipMemcpyAsync(d, ct.rows.data(), 4 * ct.rows.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d + ct.rows.size(), ct.tab.data(), 4 * ct.tab.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {             // pageable sources: complete before `ct` can go away
        gm_set_error("proto loss: class-table upload failed"); gm_dev_free(d, st); return GM_EHIP;
    }
    *d_out = d;
    return GM_OK;
}

extern "C" int gm_proto_loss_spt(const gm_batch_t* b, const float* logits, int32_t n_out, const int32_t* y, int32_t n_support, float* loss,
                                 float* acc, float* protos, float* dlogits, void* stream) {
    GM_REQUIRE(b && logits && y && loss && acc && n_support >= 1, GM_EINVAL, "proto_loss_spt: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ClassTables ct;
    GM_TRY(class_tables(b, y, n_support, ct));
    int32_t* d_rows = nullptr;
    GM_TRY(upload_tables(ct, &d_rows, st));
    int rc = GM_OK;
    if (dlogits && hipMemsetAsync(dlogits, 0, sizeof(float) * b->subs * n_out, st) != hipSuccess) { gm_set_error("proto_loss_spt: memset failed"); rc = GM_EHIP; }
    if (rc == GM_OK) {
        ProtoK k{logits, n_out, d_rows, ct.Ct, ct.n, 0, nullptr, protos, loss, acc, 1, 0, dlogits, nullptr, 0, d_rows + ct.rows.size()};
        rc = launch_proto(b, k, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == GM_OK) { gm_set_error("proto_loss_spt: kernel failed"); rc = GM_EHIP; }
    gm_dev_free(d_rows, st);
    return rc;
}

extern "C" int gm_proto_loss_qry(const gm_batch_t* b, const float* logits, int32_t n_out, const int32_t* y, const float* protos, int32_t c_task,
                                 float* loss, float* acc, float* dlogits, float* dprotos, void* stream) {
    GM_REQUIRE(b && logits && y && protos && loss && acc, GM_EINVAL, "proto_loss_qry: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ClassTables ct;
    GM_TRY(class_tables(b, y, 0, ct));
    // prototypes are indexed by sorted-class position with stride c_task: EVERY set must carry exactly c_task query classes (a set with
    // fewer would be scored against another class's prototype); gm_meta_step checks support vs query class counts per task itself
    for (int t = 0; t < b->sets; ++t)
        GM_REQUIRE(ct.tab[t * 3 + 1] == c_task, GM_EINVAL, "proto_loss_qry: set %d has %d query classes but the prototypes hold %d per set", t, ct.tab[t * 3 + 1], c_task);
    int32_t* d_rows = nullptr;
    GM_TRY(upload_tables(ct, &d_rows, st));
    int rc = GM_OK;
    if (dlogits && hipMemsetAsync(dlogits, 0, sizeof(float) * b->subs * n_out, st) != hipSuccess) { gm_set_error("proto_loss_qry: memset failed"); rc = GM_EHIP; }
    if (rc == GM_OK) {
        ProtoK k{logits, n_out, d_rows, ct.Ct, ct.n, 1, protos, nullptr, loss, acc, 1, 0, dlogits, dprotos, 0, d_rows + ct.rows.size()};
        rc = launch_proto(b, k, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == GM_OK) { gm_set_error("proto_loss_qry: kernel failed"); rc = GM_EHIP; }
    gm_dev_free(d_rows, st);
    return rc;
}

#ifdef GM_PROBES
// Probe build only (build.py --probes; include/gmeta_hip_probes.h)
static unsigned long long* g_head_dbg = nullptr;
// Phase timeline of k_head_loss (block 0, device constant clock): enable allocates 64 stamps that every later launch overwrites; out != NULL copies them
extern "C" int gm_head_loss_debug(int32_t enable, unsigned long long* out) {
    if (enable && !g_head_dbg) { GM_HIP(hipMalloc((void**)&g_head_dbg, 512)); GM_HIP(hipMemset(g_head_dbg, 0, 512)); }
    if (out && g_head_dbg) GM_HIP(hipMemcpy(out, g_head_dbg, 512, hipMemcpyDeviceToHost));
    if (!enable && g_head_dbg) { (void)hipFree(g_head_dbg); g_head_dbg = nullptr; }
    return GM_OK;
}
#else
static unsigned long long* const g_head_dbg = nullptr;      // the product library carries no probe state
#endif

// Head forward + loss (+ head backward) in one launch (k_head_loss) after a gcn_forward(..., skip_head = 1).  With
// bwd != 0 the matching gcn_backward(..., skip_head = 1) continues from dQ_L / the compact G2 written here; c.sgd (if
// set) makes the head's own parameters take their SGD step in the same launch.
static int head_loss(GcnCtx& c, const float* params, int64_t pstride, float* logits, const ProtoK& pk, int bwd, float* dparams, int64_t dstride, int sparse,
                     hipStream_t st) {
    const gm_batch* b = c.b; const gm_layout& L = c.L;
    float* dQ = nullptr; float* Gc = nullptr;
    if (bwd) {
        if (c.cone) Gc = c.bufA;
        else if (sparse && sparse_bwd_ok(L)) Gc = c.cG2;
        else {
            dQ = c.bufA;
            if (c.dq_zeroed) c.dq_zeroed = 0;
            else GM_HIP(hipMemsetAsync(dQ, 0, sizeof(float) * b->rows * L.dims[L.n_gcn], st));
        }
    }
    HeadK hk = make_head(c, params, pstride);
    const size_t proto_bytes = (proto_lds(pk.Ct, pk.n, pk.D, HL_THREADS) + 15) / 16 * 16;
    int max_subs = 0;
    for (int t = 0; t < b->sets; ++t) max_subs = std::max(max_subs, b->h_set_sub_off[t + 1] - b->h_set_sub_off[t]);
    const size_t hs_bytes = sizeof(float) * ((size_t)max_subs * b->centres * L.dims[L.n_gcn] + (size_t)L.n_out * (L.hc + 1) + 2 * (size_t)max_subs * L.n_out +
                                             (size_t)max_subs * b->centres + (size_t)pk.Ct * pk.n) + 16;      // + the centre-row scratch + the class rows
    const int stage_on = gm_knob().head_stage;
    const int stage_h = stage_on && proto_bytes + hs_bytes <= 150 * 1024;
    const size_t lds = proto_bytes + (stage_h ? hs_bytes : 0);
    // Workgroup size (GM_HEAD_THREADS, default 1024): one workgroup per task.  Measured: 256 threads -- which could start on a CU that a persistent
    // GEMM workgroup of the other stream fills -- lose more inside the kernel than they gain at its start (FirstMM shape 1.62 -> 1.89 ms per
    // meta-step, 4-task arxiv shard 4.59 -> 4.86; 512: 1.71 / 4.67).
    int nt = gm_knob().head_threads;
    if (nt != 256 && nt != 512) nt = 1024;
    const SgdK sg = bwd ? c.sgd : SgdK{nullptr, 0, nullptr, 0, 0.f};
    gm_prof_begin(GM_PROF_HEAD, st, b->subs);
    if (nt == 256) { GM_TRY(gm_func_full_lds((const void*)k_head_loss<256>)); hipLaunchKernelGGL(k_head_loss<256>, dim3(b->sets), dim3(256), lds, st, hk, logits, pk, bwd, dparams, dstride, dQ, Gc, sg, stage_h, (int)(proto_bytes / sizeof(float)), 0, g_head_dbg); }
    else if (nt == 512) { GM_TRY(gm_func_full_lds((const void*)k_head_loss<512>)); hipLaunchKernelGGL(k_head_loss<512>, dim3(b->sets), dim3(512), lds, st, hk, logits, pk, bwd, dparams, dstride, dQ, Gc, sg, stage_h, (int)(proto_bytes / sizeof(float)), 0, g_head_dbg); }
    else { GM_TRY(gm_func_full_lds((const void*)k_head_loss<1024>)); hipLaunchKernelGGL(k_head_loss<1024>, dim3(b->sets), dim3(1024), lds, st, hk, logits, pk, bwd, dparams, dstride, dQ, Gc, sg, stage_h, (int)(proto_bytes / sizeof(float)), 0, g_head_dbg); }
#ifdef GM_PROBES
    {   // tools/head_loss_probe.py: the same launch again -- it is idempotent -- to see what a warm instruction cache / warm L2 is worth
        static const int twice = getenv("GM_HEAD_TWICE") ? atoi(getenv("GM_HEAD_TWICE")) : 0;
        if (twice && nt != 256 && nt != 512)
            hipLaunchKernelGGL(k_head_loss<1024>, dim3(b->sets), dim3(1024), lds, st, hk, logits, pk, bwd, dparams, dstride, dQ, Gc, sg, stage_h, (int)(proto_bytes / sizeof(float)), 0, g_head_dbg);
    }
#endif
    GM_HIP(hipGetLastError());
    gm_prof_end(GM_PROF_HEAD, st);
    if (bwd && dQ && hk.dq_amax) c.dqv = true;
    return GM_OK;
}

// ================================================================================ the fused meta-step
// theta in the caller's layout -> the kernels' layout: `shift` zeros inserted at `cut` (the zero weight rows of the padded feature columns)
__global__ void k_pad_params(const float* theta, int64_t P, int64_t cut, int64_t shift, float* out) {
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < P + shift; id += (int64_t)gridDim.x * blockDim.x)
        out[id] = id < cut ? theta[id] : (id < cut + shift ? 0.f : theta[id - shift]);
}

// The model the kernels run: layer 1 reads the store's padded feature rows (gm_store::feat_ld columns, the extra ones zero), so W_1
// gets matching zero rows -- same sums, but the layer takes the vectorised aggregate / DMA GEMM / fast weight-gradient kernels
// whatever the dataset's feature width (50 and 5 in the reference's Tissue-PPI / FirstMM-DB configs).
static gm_model_t internal_model(const gm_model_t* m, const gm_store* store, int64_t* cut, int64_t* shift) {
    gm_model_t mp = *m; *cut = 0; *shift = 0;
    if (store->feat_ld != store->feat_dim && m->dims[0] == store->feat_dim) {
        mp.dims[0] = store->feat_ld;
        *cut = (int64_t)store->feat_dim * m->dims[1]; *shift = (int64_t)(store->feat_ld - store->feat_dim) * m->dims[1];
    }
    return mp;
}

struct MetaStreams {
    hipStream_t side = nullptr, side2 = nullptr;      // query streams (side2: GM_QUERY_STREAMS = 2)
    hipStream_t main = nullptr;      // CU-partitioned mode only (GM_CU_MASK_SUPPORT): the support chain's own stream, masked to its CUs
    std::vector<hipEvent_t> ev;
    int ensure(int n) {
        const int per_xcd = gm_knob().cu_mask_support;
        if (!side && per_xcd > 0 && per_xcd * GM_NXCD < gm_num_cus()) {
            // CU-partitioned streams: the support chain owns `per_xcd` CUs of every XCD, the query evaluations the rest, so that kernels of
            // the two chains CO-RESIDE on the chip instead of time-slicing it (a persistent 1024-thread GEMM workgroup needs a completely
            // empty CU, which it never gets while the other queue keeps feeding small workgroups).  Mask bit i = CU i / 8 of XCD i % 8
            // (the driver deals queue mask bits round-robin over the XCDs), so [0, 8 s) is s CUs on each XCD.
            const int cus = gm_num_cus(), ns = per_xcd * GM_NXCD, words = (cus + 31) / 32;
            std::vector<uint32_t> ms(words, 0u), mq(words, 0u);
            for (int i = 0; i < cus; ++i) (i < ns ? ms : mq)[i >> 5] |= 1u << (i & 31);
            GM_HIP(hipExtStreamCreateWithCUMask(&main, (uint32_t)words, ms.data()));
            GM_HIP(hipExtStreamCreateWithCUMask(&side, (uint32_t)words, mq.data()));
            gm_stream_set_cus(main, ns); gm_stream_set_cus(side, cus - ns);
        }
        if (!side) {
            // The query stream carries bulk, throughput-bound work; the caller's stream carries the latency-critical
            // support chain.  A lower priority (a) lets support kernels win CUs when both have work and (b) gives this
            // stream a hardware queue of its own: HIP multiplexes same-priority streams onto a small pool of HW queues
            // (GPU_MAX_HW_QUEUES, default 4), and once RCCL/torch have created their streams an ordinary stream created
            // here was observed to alias the caller's queue, silently serialising the whole step.
            int lo = 0, hi = 0;
            const int use_prio = gm_knob().side_stream_priority;
            if (use_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
                GM_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));      // `lo` = least priority
            } else {
                (void)hipGetLastError();
                GM_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            }
        }
        if (!side2 && !main) {
            int lo = 0, hi = 0;
            if (gm_knob().side_stream_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) GM_HIP(hipStreamCreateWithPriority(&side2, hipStreamNonBlocking, lo));
            else { (void)hipGetLastError(); GM_HIP(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking)); }
        }
        while ((int)ev.size() < n) { hipEvent_t e; GM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev.push_back(e); }
        return GM_OK;
    }
};
// One side stream / event pool / staging ring per (thread, device): a process may drive several GPUs.
static MetaStreams& meta_streams() {
    static thread_local std::map<int, MetaStreams> m;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    return m[dev];
}

// Pinned host staging for the per-step class tables: the H2D copy is truly asynchronous (no stream synchronisation in
// gm_meta_step); a slot is reused only after the copy that read it has completed (4 slots: never waits in practice).
struct StageRing {
    static const int N = 4;
    void* buf[N] = {}; size_t cap[N] = {}; hipEvent_t ev[N] = {}; bool busy[N] = {}; int next = 0;
    int acquire(size_t bytes, void** out, int* slot) {
        const int k = next; next = (next + 1) % N;
        if (busy[k]) { GM_HIP(hipEventSynchronize(ev[k])); busy[k] = false; }
        if (!ev[k]) GM_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        if (cap[k] < bytes) {
            if (buf[k]) (void)hipHostFree(buf[k]);
            buf[k] = nullptr; cap[k] = 0;
            const size_t want = std::max<size_t>(bytes * 2, 64 * 1024);
            GM_HIP(hipHostMalloc(&buf[k], want, hipHostMallocDefault));
            cap[k] = want;
        }
        *out = buf[k]; *slot = k;
        return GM_OK;
    }
    int release_after(int slot, hipStream_t st) { GM_HIP(hipEventRecord(ev[slot], st)); busy[slot] = true; return GM_OK; }
};
static StageRing& stage_ring() {
    static thread_local std::map<int, StageRing> m;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    return m[dev];
}

struct MetaPlan {
    gm_layout L; int T, K; int64_t Pp;       // Pp = P padded to 64 floats: per-task weight vectors stay 16-B aligned
    GcnCtx S, Q, Q2;                         // Q2: second query context (own activations) when two query streams are used
    int nq_ctx;
    float *logit_q2;
    float *fw, *g, *gq, *gp, *logit_s, *logit_q, *dlog_s, *dlog_q, *protos, *dprotos, *ls, *as_, *lq, *aq, *theta_p;
    PlaneDir pd;
    int64_t TP, proto_sz;               // fw holds K vectors-of-tasks fw_1..fw_K (distinct buffers: the support chain may run ahead)
    int32_t *rows_s, *rows_q, *tab_s, *tab_q;     // class tables: one contiguous block [rows_s | rows_q | tab_s | tab_q | featb_s | featb_q]
    unsigned *featb_s, *featb_q;                  // [T] each: per-task bound of the layer-1 operand (fp32 bit patterns; rides in the class-table copy)
    unsigned* viol;                               // the step's violation word (gm_bound.h), zeroed with the bound slots; NULL without two-piece kernels
    int Ct, ns, nq;
    int uni_s = 0, uni_q = 0;                     // class layouts uniform over the tasks (ProtoK::uniform)
    unsigned* bound_ws; int64_t bound_words;      // gm_bound.h slots of this step ([S passes | Q passes | weights]), zeroed by ONE memset; NULL: three-piece kernels
};

static int meta_plan(MetaPlan& p, const gm_batch* spt, const gm_batch* qry, const gm_model_t* m, const gm_hparams_t* hp, void* ws, int64_t ws_bytes,
                     int Ct, int ns, int nq, int64_t* need) {
    GM_TRY(gm_make_layout(m, &p.L));
    p.T = spt->sets; p.K = hp->update_step; p.Pp = (p.L.P + 63) / 64 * 64;
    p.S = GcnCtx{}; p.Q = GcnCtx{}; p.Q2 = GcnCtx{}; p.S.is_support = true; p.S.b = spt; p.Q.b = qry; p.Q2.b = qry; p.S.L = p.L; p.Q.L = p.L; p.Q2.L = p.L;
    if (hp->cone) {                      // receptive-field tables: built on first use, cached in the batch
        const gm_cone *cs = nullptr, *cq = nullptr;
        GM_TRY(gm_batch_cone(spt, p.L.n_gcn, spt->stream, &cs));
        GM_TRY(gm_batch_cone(qry, p.L.n_gcn, qry->stream, &cq));
        if (cs->ok && cq->ok) { p.S.cone = cs; p.Q.cone = cq; p.Q2.cone = cq; }      // else: a self pair among the centres -> dense schedule
    }
    Carver cv(ws, ws_bytes);
    const int64_t TP = (int64_t)p.T * p.Pp; const int C = p.L.n_out; const int K1 = p.K + 1;
    p.TP = TP; p.proto_sz = (int64_t)p.T * 256 * C;
    p.theta_p = cv.take<float>(p.Pp);
    p.fw = cv.take<float>(TP * p.K); p.g = cv.take<float>(TP); p.gq = cv.take<float>(TP); p.gp = cv.take<float>(TP);
    p.logit_s = cv.take<float>((int64_t)spt->subs * C); p.logit_q = cv.take<float>((int64_t)qry->subs * C);
    p.dlog_s = cv.take<float>((int64_t)spt->subs * C); p.dlog_q = cv.take<float>((int64_t)qry->subs * C);
    p.protos = cv.take<float>(p.proto_sz * p.K); p.dprotos = cv.take<float>(p.proto_sz);
    p.ls = cv.take<float>((int64_t)p.T * K1); p.as_ = cv.take<float>((int64_t)p.T * K1);
    p.lq = cv.take<float>((int64_t)p.T * K1); p.aq = cv.take<float>((int64_t)p.T * K1);
    {   // rows of a set never exceed its subgraphs: [rows_s (spt->subs) | rows_q (qry->subs) | tab_s (3T) | tab_q (3T)]
        int32_t* blk = cv.take<int32_t>((int64_t)spt->subs + qry->subs + 8 * (int64_t)p.T);
        p.rows_s = blk; p.rows_q = blk ? blk + spt->subs : nullptr;
        p.tab_s = blk ? p.rows_q + qry->subs : nullptr; p.tab_q = blk ? p.tab_s + 3 * p.T : nullptr;
        p.featb_s = blk ? reinterpret_cast<unsigned*>(p.tab_q + 3 * p.T) : nullptr; p.featb_q = blk ? p.featb_s + p.T : nullptr;
    }
    // Two query streams (GM_QUERY_STREAMS=2, off by default): evaluations k and k + 1 are independent of each other (each needs only fw_k and the
    // prototypes of step k - 1), so they can run in two contexts on two streams.  Measured: no gain anywhere -- 4-task arxiv shard 4.68 vs 4.67 ms,
    // 8 tasks 8.22 vs 8.07, Tissue shape 3.46 vs 3.55, FirstMM shape 1.61 vs 1.66, task_num 32 28.09 vs 28.17 (same box): kernels of
    // different queues time-slice the chip, they do not fill each other's stalls.  Kept as a knob (bitwise the one-stream result; tested).
    {
        const int qs = gm_knob().query_streams;
        p.nq_ctx = (hp->serialize || hp->cone || hp->sparse_bwd) ? 1 : (qs == 2 ? 2 : 1);
    }
    p.logit_q2 = p.nq_ctx == 2 ? cv.take<float>((int64_t)qry->subs * C) : nullptr;
    gcn_carve(p.S, cv); gcn_carve(p.Q, cv);
    if (p.nq_ctx == 2) { gcn_carve(p.Q2, cv); p.Q2.hub_set = 1; }
    // split-bf16 planes of every fast-weight vector (dense schedule, layers the split GEMM can take): forward planes for every such
    // layer, dZ planes for layers >= 1; ~1 MB per task and inner step at 128/256/256
    p.pd = PlaneDir{};
    if (gm_gemm_mode() == 1 && !p.S.cone && p.K < 64) {
        int64_t per_k = 0;
        for (int l = 0; l < p.L.n_gcn; ++l) {
            const int fi = p.L.dims[l], fo = p.L.dims[l + 1];
            p.pd.off[l][0] = p.pd.off[l][1] = -1;
            if (fi > fo) continue;                                           // multiply-first layers keep the on-the-fly path
            const int64_t sz = (int64_t)p.T * 3 * fi * fo;
            if ((fo == 256 || fo == 128) && fi % 16 == 0 && fi >= 32) { p.pd.off[l][0] = per_k; per_k += sz; }                 // X @ W: K = fi, N = fo
            if (l > 0 && (fi == 256 || fi == 128) && fo % 16 == 0 && fo >= 32) { p.pd.off[l][1] = per_k; per_k += sz; }        // dQ @ W^T: K = fo, N = fi
        }
        if (per_k > 0) {
            p.pd.base = cv.take<uint16_t>(per_k * p.K); p.pd.per_k = per_k; p.pd.fw0 = p.fw; p.pd.TP = TP; p.pd.K = p.K;
            if (!p.pd.base) p.pd.base = reinterpret_cast<uint16_t*>(1);      // sizing pass (no workspace yet): keep the layout decisions identical
        }
    }
    // two-piece fp16 split kernels: when the weight planes are kept, every GCN layer is aggregate-first and the dense schedule runs
    p.bound_ws = nullptr; p.bound_words = 0; p.viol = nullptr;
    bool agg_first = true;
    for (int l = 0; l < p.L.n_gcn; ++l) agg_first = agg_first && p.L.dims[l] <= p.L.dims[l + 1];
    if (p.pd.base && gm_split_np() == 2 && agg_first && !hp->sparse_bwd && !p.S.cone && spt->store->d_feat_amax &&
        spt->rows + qry->rows >= gm_knob().split16_min_rows) {
        const int per_pass = 2 * p.L.n_gcn + 1;
        const int64_t ws_s = (int64_t)p.K * per_pass * p.T * GM_BOUND_PAD, ws_q = (int64_t)K1 * per_pass * p.T * GM_BOUND_PAD, ws_w = (int64_t)p.L.n_gcn * GM_BOUND_PAD;
        p.bound_words = ws_s + ws_q + ws_w + GM_BOUND_PAD;                   // (+ the violation word, on a line of its own)
        p.bound_ws = cv.take<unsigned>(p.bound_words);
        if (!p.bound_ws) p.bound_ws = reinterpret_cast<unsigned*>(16);       // sizing pass
        p.S.np = p.Q.np = p.Q2.np = 2;
        p.S.am = p.bound_ws; p.S.am_passes = p.K; p.Q.am = p.bound_ws + ws_s; p.Q.am_passes = K1;
        p.Q2.am = p.Q.am; p.Q2.am_passes = K1;       // (the two query contexts share the slot array: evaluation j uses pass j, set by the step)
        p.pd.wam = p.bound_ws + ws_s + ws_q;
        p.viol = p.pd.wam + ws_w; p.pd.viol = p.viol;
    }
    p.Ct = Ct; p.ns = ns; p.nq = nq;
    if (need) *need = cv.used + 256;
    GM_REQUIRE(cv.ok(), GM_ENOMEM, "meta_step: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)cv.used);
    return GM_OK;
}

extern "C" int64_t gm_meta_ws_bytes(const gm_batch_t* spt, const gm_batch_t* qry, const gm_model_t* m, const gm_hparams_t* hp) {
    if (!spt || !qry || !m || !hp) return -1;
    MetaPlan p; int64_t need = 0, cut, shift;
    const gm_model_t mp = internal_model(m, spt->store, &cut, &shift);
    if (meta_plan(p, spt, qry, &mp, hp, nullptr, 0, 1, 1, 1, &need) != GM_OK) return -1;
    return need;
}

extern "C" int64_t gm_meta_out_floats(const gm_batch_t* spt, const gm_model_t* m, const gm_hparams_t* hp) {
    gm_layout L;
    if (!spt || !hp || gm_make_layout(m, &L) != GM_OK) return -1;
    return L.P + 2 * (int64_t)(hp->update_step + 1) + 1 + (int64_t)spt->sets * (hp->update_step + 1) + 1;
}

extern "C" int gm_meta_step(const gm_batch_t* spt, const gm_batch_t* qry, const int32_t* y_spt, const int32_t* y_qry, const gm_model_t* m,
                            const gm_hparams_t* hp, const float* theta, float* out, int64_t out_floats, void* ws, int64_t ws_bytes, void* stream) {
    GM_REQUIRE(spt && qry && y_spt && y_qry && m && hp && theta && out && ws, GM_EINVAL, "meta_step: NULL argument");
    GM_REQUIRE(out_floats >= gm_meta_out_floats(spt, m, hp), GM_ENOMEM, "meta_step: out holds %lld floats, the step writes %lld", (long long)out_floats,
               (long long)gm_meta_out_floats(spt, m, hp));
    GM_REQUIRE(spt->sets == qry->sets, GM_EINVAL, "meta_step: %d support sets but %d query sets", spt->sets, qry->sets);
    GM_REQUIRE(spt->store == qry->store, GM_EINVAL, "meta_step: support and query batches come from different stores");
    const int K = hp->update_step;
    GM_REQUIRE(K >= 1, GM_EINVAL, "meta_step: update_step must be >= 1");
    GM_REQUIRE(!hp->need_meta_grad || K >= 2, GM_EINVAL,
               "meta_step: update_step must be >= 2 for training (losses_q[0..1] are computed under no_grad, meta.py:129-141)");
    GM_REQUIRE(((uintptr_t)theta & 15) == 0, GM_EINVAL, "meta_step: theta must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    gm_phase_timer tm("meta_step");
    ClassTables cs, cq;
    GM_TRY(class_tables(spt, y_spt, hp->k_spt, cs));
    GM_TRY(class_tables(qry, y_qry, 0, cq));
    for (int t = 0; t < spt->sets; ++t)
        GM_REQUIRE(cs.tab[t * 3 + 1] == cq.tab[t * 3 + 1], GM_EINVAL, "meta_step: task %d has %d support classes but %d query classes", t, cs.tab[t * 3 + 1], cq.tab[t * 3 + 1]);
    const int Ct = cs.Ct, ns = cs.n, nq = cq.n;
    MetaPlan p;
    int64_t cut = 0, shift = 0;
    const gm_model_t mp = internal_model(m, spt->store, &cut, &shift);
    gm_layout Lu;                                          // the caller's parameter layout (theta, out)
    GM_TRY(gm_make_layout(m, &Lu));
    GM_TRY(meta_plan(p, spt, qry, &mp, hp, ws, ws_bytes, Ct, ns, nq, nullptr));
    const gm_layout& L = p.L; const int T = p.T, C = L.n_out, K1 = K + 1; const int64_t Pp = p.Pp;
    p.uni_s = cs.uniform ? 1 : 0; p.uni_q = cq.uniform ? 1 : 0;
    p.S.pd = p.Q.pd = p.Q2.pd = p.pd.base ? &p.pd : nullptr;
    if (shift) {
        hipLaunchKernelGGL(k_pad_params, dim3((int)std::min<int64_t>(512, (L.P + 255) / 256)), dim3(256), 0, st, theta, Lu.P, cut, shift, p.theta_p);
        GM_HIP(hipGetLastError());
        theta = p.theta_p;
    }
    {   // class tables -> pinned staging -> ONE asynchronous copy (no host synchronisation in the meta-step)
        const size_t n_tab = (size_t)spt->subs + qry->subs + 8 * (size_t)T;
        void* h = nullptr; int slot = 0;
        StageRing& ring = stage_ring();
        GM_TRY(ring.acquire(4 * n_tab, &h, &slot));
        int32_t* hp32 = (int32_t*)h;
        memset(hp32, 0, 4 * n_tab);
        memcpy(hp32, cs.rows.data(), 4 * cs.rows.size());
        memcpy(hp32 + spt->subs, cq.rows.data(), 4 * cq.rows.size());
        memcpy(hp32 + spt->subs + qry->subs, cs.tab.data(), 4 * cs.tab.size());
        memcpy(hp32 + spt->subs + qry->subs + 3 * (size_t)T, cq.tab.data(), 4 * cq.tab.size());
        if (p.bound_ws) {
            // two-piece kernels: the layer-1 operand of task t is bounded by the largest feature of the graphs its subgraphs come from.  A step
            // that touches a LOOSE table (gm_store::h_feat_mean: largest entry more than 2^14 above the typical one, or not finite) runs
            // entirely on the three-piece kernels
            const gm_store* sto = spt->store;
            bool loose = false;
            int k = 0;
            for (const gm_batch* bb : {spt, qry}) {
                float* fb = reinterpret_cast<float*>(hp32 + spt->subs + qry->subs + 6 * (size_t)T) + (size_t)(k++) * T;
                for (int t = 0; t < T; ++t) {
                    float mx = 0.f;
                    for (int sg = bb->h_set_sub_off[t]; sg < bb->h_set_sub_off[t + 1]; ++sg) {
                        const int gi = bb->h_graph[sg];
                        const float a = sto->h_feat_amax[gi], mean = sto->h_feat_mean[gi];
                        if (!(a <= 1.125899906842624e15f) || a > 16384.f * mean) loose = true;          // (2^50; 2^14 above the typical entry)
                        mx = a > mx ? a : mx;
                    }
                    fb[t] = mx;
                }
            }
            if (loose) {
                p.S.np = p.Q.np = p.Q2.np = 3; p.S.am = p.Q.am = p.Q2.am = nullptr; p.S.am_passes = p.Q.am_passes = p.Q2.am_passes = 0;
                p.pd.wam = nullptr; p.pd.viol = nullptr; p.viol = nullptr; p.bound_ws = nullptr;
            } else { p.S.feat_bound = p.featb_s; p.Q.feat_bound = p.Q2.feat_bound = p.featb_q; }
        }
        GM_HIP(hipMemcpyAsync(p.rows_s, h, 4 * n_tab, hipMemcpyHostToDevice, st));
        GM_TRY(ring.release_after(slot, st));
    }
    if (p.bound_ws) {
        GM_TRY(gm_batch_gains(spt, st)); GM_TRY(gm_batch_gains(qry, st));      // (computed once per batch, at its first two-piece step)
        // bound slots of this step: zero (the producers use atomicMax), then the maxima of theta's weight matrices (slot k = 0)
        GM_HIP(hipMemsetAsync(p.bound_ws, 0, sizeof(unsigned) * p.bound_words, st));
        p.pd.theta = theta;
        int64_t off[GM_MAX_GCN], n[GM_MAX_GCN];
        for (int l = 0; l < L.n_gcn; ++l) { off[l] = L.w_off[l]; n[l] = (int64_t)L.dims[l] * L.dims[l + 1]; }
        GM_TRY(gm_amax_segs(theta, off, n, L.n_gcn, p.pd.wam, GM_BOUND_PAD, st));
    }
    gm_prof_reset(GM_PROF_STEP_CATS);
    gm_prof_reset_cat(GM_PROF_GEMM_SPLIT_BYTES); gm_prof_reset_cat(GM_PROF_AGG_BOUND);
    gm_prof_reset_cat(GM_PROF_GEMM_BYTES); gm_prof_reset_cat(GM_PROF_WGRAD_BYTES); gm_prof_reset_cat(GM_PROF_HEAD);
    tm.lap("plan");
    // Two streams: `st` carries the support chain (the serial dependency through the fast weights: forward -> loss ->
    // backward -> SGD, K times), `sq` carries the K+1 query evaluations, each of which only needs fw_k and the
    // prototypes of step k-1.  The small latency-bound support kernels thus overlap the throughput-bound query work.
    MetaStreams& ms = meta_streams();
    GM_TRY(ms.ensure(2 * K + 10));
    hipStream_t sq = hp->serialize ? st : ms.side;
    const bool two_q = p.nq_ctx == 2 && !hp->serialize && ms.side2;
    hipStream_t sq2 = two_q ? ms.side2 : sq;
    if (two_q) GM_TRY(gm_batch_hub_alt(qry, st));          // private hub counters / partial rows for the second query stream
    hipStream_t const st0 = st;                            // the caller's stream: inputs arrive on it, the output leaves on it
    if (!hp->serialize && ms.main) st = ms.main;           // CU-partitioned mode: the support chain runs on its own masked stream
    int ev = 0;
    auto signal = [&](hipStream_t from) -> hipEvent_t { hipEvent_t e = ms.ev[ev++]; (void)hipEventRecord(e, from); return e; };
    auto wait = [&](hipStream_t who, hipEvent_t e) { (void)hipStreamWaitEvent(who, e, 0); };
    {
        hipEvent_t e_in = signal(st0);                     // inputs (theta, class tables, batches) are ready
        wait(sq, e_in);
        if (two_q) wait(sq2, e_in);
        if (st != st0) wait(st, e_in);
    }

    auto fw = [&](int k) -> float* { return p.fw + (int64_t)(k - 1) * p.TP; };       // fw_k, k = 1..K
    auto protos = [&](int k) -> float* { return p.protos + (int64_t)k * p.proto_sz; };   // prototypes of support step k
    const int hoist = hp->hoist_z1, sparse = hp->sparse_bwd;
    const SgdK no_sgd{nullptr, 0, nullptr, 0, 0.f};
    // One support step (meta.py:122-126,145-151): forward -> [head + proto_loss_spt + head backward, one launch] -> backward,
    // with the SGD step w_next = w - lr * grad written by the kernels that produce each gradient.
    auto spt_step = [&](int k, const float* w, int64_t wstride, float* w_next) -> int {
        GM_TRY(gcn_forward(p.S, w, wstride, p.logit_s, st, hoist, 1, (sparse && sparse_bwd_ok(p.L)) ? 0 : 2));
        p.S.sgd = SgdK{w, wstride, w_next, Pp, hp->update_lr};
        ProtoK pk{p.logit_s, C, p.rows_s, Ct, ns, 0, nullptr, protos(k), p.ls, p.as_, K1, k, p.dlog_s, nullptr, 0, p.tab_s, p.uni_s};
        GM_TRY(head_loss(p.S, w, wstride, p.logit_s, pk, 1, p.g, Pp, sparse, st));
        return GM_OK;
    };
    auto spt_step_bwd = [&](const float* w, int64_t wstride) -> int {
        GM_TRY(gcn_backward(p.S, w, wstride, p.dlog_s, p.g, Pp, st, sparse, 1));
        p.S.sgd = no_sgd;
        return GM_OK;
    };
    // One query evaluation (meta.py:129-141,152-154): forward on sq, then head + proto_loss_qry (+ head backward when the
    // meta-gradient is wanted) once the prototypes / weights it needs are ready.
    // Evaluation j (j = 0: theta, j >= 1: fw_j) runs in query context j % 2 on that context's stream when two query streams are used.
    auto q_ctx = [&](int j) -> GcnCtx& { return (two_q && (j & 1)) ? p.Q2 : p.Q; };
    auto q_str = [&](int j) -> hipStream_t { return (two_q && (j & 1)) ? sq2 : sq; };
    auto q_log = [&](int j) -> float* { return (two_q && (j & 1)) ? p.logit_q2 : p.logit_q; };
    auto qry_fwd = [&](int j, const float* w, int64_t wstride, int fwd_only) -> int {
        GcnCtx& c = q_ctx(j);
        if (c.np == 2) c.am_pass = j - 1;                  // (gcn_forward advances it: evaluation j records its bounds in pass j, whichever context runs it)
        return gcn_forward(c, w, wstride, q_log(j), q_str(j), hoist, 1, fwd_only);
    };
    auto qry_loss = [&](int j, const float* w, int64_t wstride, int col, int kproto, bool grad) -> int {
        ProtoK pk{q_log(j), C, p.rows_q, Ct, nq, 1, protos(kproto), nullptr, p.lq, p.aq, K1, col, grad ? p.dlog_q : nullptr, grad ? p.dprotos : nullptr, 0, p.tab_q, p.uni_q};
        return head_loss(q_ctx(j), w, wstride, q_log(j), pk, grad ? 1 : 0, p.gq, Pp, sparse, q_str(j));
    };
    // ---- support step 0 (meta.py:122-126) on st ; query evaluations 0 and 1 (meta.py:129-141) on sq.  Host enqueue order matters at the
    // start of a step (the GPU is idle and a launch costs the host ~5 us): the first query forward needs nothing but theta and is the head
    // of the longer chain on small shards, so it goes out first -- behind the support step's ~17 launches it started ~110 us late.
    // (Where the support chain is the longer one -- small query batches: the FirstMM shape lost 2.5 % -- its launches keep the lead.)
    const bool query_first = qry->rows >= 100000;
    if (query_first) GM_TRY(qry_fwd(0, theta, 0, 1));
    GM_TRY(spt_step(0, theta, 0, fw(1)));
    hipEvent_t e_proto0 = signal(st);
    if (!query_first) GM_TRY(qry_fwd(0, theta, 0, 1));
    wait(q_str(0), e_proto0);
    GM_TRY(qry_loss(0, theta, 0, 0, 0, false));
    GM_TRY(spt_step_bwd(theta, 0));
    hipEvent_t e_fw = signal(st);                          // fw_1 ready (and, being later on st, the prototypes of step 0)
    wait(q_str(1), e_fw);
    GM_TRY(qry_fwd(1, fw(1), Pp, 1));
    GM_TRY(qry_loss(1, fw(1), Pp, 1, 0, false));
    bool have_grad = false;
    for (int k = 1; k < K; ++k) {            // meta.py:143-157
        GM_TRY(spt_step(k, fw(k), Pp, fw(k + 1)));
        GM_TRY(spt_step_bwd(fw(k), Pp));
        e_fw = signal(st);                                 // fw_{k+1} and the prototypes of step k are ready
        const int j = k + 1;
        wait(q_str(j), e_fw);
        const bool last = hp->need_meta_grad && k == K - 1;
        GM_TRY(qry_fwd(j, fw(k + 1), Pp, last ? ((sparse && sparse_bwd_ok(p.L)) ? 0 : 2) : 1));       // only the last evaluation is differentiated
        GM_TRY(qry_loss(j, fw(k + 1), Pp, k + 1, k, last));
        if (last) {
            // first-order meta-gradient (no create_graph anywhere, meta.py:125,149): d L_q / d fw_K through the
            // query forward (on its query stream) plus d L_q / d fw_{K-1} through the prototypes of the last support forward (on st).
            hipEvent_t e_dp = signal(q_str(j));            // dprotos ready
            GM_TRY(gcn_backward(q_ctx(j), fw(k + 1), Pp, p.dlog_q, p.gq, Pp, q_str(j), sparse, 1));
            wait(st, e_dp);
            GM_HIP(hipMemsetAsync(p.dlog_s, 0, sizeof(float) * spt->subs * C, st));
            hipLaunchKernelGGL(k_protos_to_dlogits, dim3(T), dim3(256), 0, st, p.dprotos, p.rows_s, p.tab_s, Ct, C, p.dlog_s);
            GM_TRY(gcn_backward(p.S, fw(k), Pp, p.dlog_s, p.gp, Pp, st, sparse));
            have_grad = true;
        }
    }
    wait(st0, signal(sq));                                 // join
    if (two_q) wait(st0, signal(sq2));
    if (st != st0) wait(st0, signal(st));
    st = st0;
    const int64_t tot = Lu.P + 2 * K1 + 1 + (int64_t)T * K1;
    hipLaunchKernelGGL(k_finalize, dim3((int)std::min<int64_t>(1024, (tot + 255) / 256)), dim3(256), 0, st,
                       have_grad ? p.gq : nullptr, p.gp, Pp, Lu.P, T, p.lq, p.aq, K1, out, cut, shift, p.viol);
    GM_HIP(hipGetLastError());
    gm_batch_mark_use(spt, st); gm_batch_mark_use(qry, st);
    return GM_OK;
}

// ================================================================================ after the all-reduce
// grad[i] = head[i] / task count ;  found_inf = isnan(losses_q[K] / task count)  (meta.py:161-163).  `head` is the
// (all-reduced) [grad(P) | losses_q(K1) | corrects(K1) | count] block of gm_meta_step's output.
__global__ void k_meta_finish(const float* head, int64_t P, int K1, float* grad, float* found_inf) {
    const float cnt = head[P + 2 * K1];
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < P; id += (int64_t)gridDim.x * blockDim.x) grad[id] = head[id] / cnt;
    if (blockIdx.x == 0 && threadIdx.x == 0) { const float l = head[P + K1 - 1] / cnt; *found_inf = (l != l) ? 1.f : 0.f; }
}

extern "C" int gm_meta_finish(const float* head, int64_t P, int32_t K1, float* grad, float* found_inf, void* stream) {
    GM_REQUIRE(head && grad && found_inf && P >= 1 && K1 >= 1, GM_EINVAL, "meta_finish: bad arguments");
    hipLaunchKernelGGL(k_meta_finish, dim3((int)std::min<int64_t>(512, (P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head, P, K1, grad, found_inf);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// The same guard AND the Adam step (meta.py:97,161-169: optim.Adam(lr = meta_lr), default betas / eps, no weight decay) in ONE launch: between two
// meta-steps the stream otherwise carries k_meta_finish + three launches of torch's fused optimiser, and the host spends ~100 us inside
// optimizer.step() -- on the small configurations (1.5 - 4 ms per meta-step) that is where the GPU sat empty.  torch's fused Adam rule, in fp32:
//   m <- m + (1 - b1) (g - m);  v <- b2 v + (1 - b2) g g;  theta <- theta - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t = step + 1; a NaN reduced query loss leaves theta, m, v and the step count untouched (`if torch.isnan(loss_q): pass`).  steps[0 .. n_steps)
// are the per-parameter step counters of the torch optimiser's state (all equal): read by every block, written by the LAST block to finish
// (ticket), so that no block can see the incremented value.
__global__ __launch_bounds__(256) void k_meta_finish_adam(const float* head, int64_t P, int K1, float* theta, float* m, float* v, float* grad, float* steps, int n_steps,
                                                          float lr, float b1, float b2, float eps, float* found_inf, unsigned* ticket) {
    const float cnt = head[P + 2 * K1];
    const float l = head[P + K1 - 1] / cnt;
    const bool skip = l != l;
    const float t = steps[0] + 1.f;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)t)), bc2s = sqrtf((float)(1.0 - pow((double)b2, (double)t)));
    const float step_size = lr / bc1;
    for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < P; id += (int64_t)gridDim.x * blockDim.x) {
        const float g = head[id] / cnt;
        grad[id] = g;
        if (skip) continue;
        const float m0 = m[id], v0 = v[id];
        const float m1 = m0 + (1.f - b1) * (g - m0);
        const float v1 = b2 * v0 + (1.f - b2) * g * g;
        m[id] = m1; v[id] = v1;
        theta[id] -= step_size * (m1 / (sqrtf(v1) / bc2s + eps));
    }
    __syncthreads();                                       // every thread of this block has read steps[0]
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {      // every block has
            *ticket = 0u;
            *found_inf = skip ? 1.f : 0.f;
            if (!skip) for (int i = 0; i < n_steps; ++i) steps[i] = t;
        }
    }
}

extern "C" int gm_meta_finish_adam(const float* head, int64_t P, int32_t K1, float* theta, float* exp_avg, float* exp_avg_sq, float* grad, float* steps, int32_t n_steps,
                                   float lr, float beta1, float beta2, float eps, float* found_inf, uint32_t* ticket, void* stream) {
    GM_REQUIRE(head && theta && exp_avg && exp_avg_sq && grad && steps && found_inf && ticket && P >= 1 && K1 >= 1 && n_steps >= 1, GM_EINVAL, "meta_finish_adam: bad arguments");
    hipLaunchKernelGGL(k_meta_finish_adam, dim3((int)std::min<int64_t>(256, (P + 511) / 512)), dim3(256), 0, (hipStream_t)stream, head, P, (int)K1, theta, exp_avg, exp_avg_sq, grad,
                       steps, (int)n_steps, lr, beta1, beta2, eps, found_inf, ticket);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// ================================================================================ dense update, exported for numerics tests
extern "C" int gm_dense_update(const gm_batch_t* b, const float* x, int32_t K, const float* W, int64_t w_stride, int32_t N, float* out, int32_t mode,
                               void* stream) {
    GM_REQUIRE(b && x && W && out && K >= 1 && N >= 1, GM_EINVAL, "dense_update: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    gm_gemm_args g{};
    g.A = x; g.lda = K; g.B = W; g.b_stride = w_stride; g.C = out; g.ldc = N; g.K = K; g.N = N;
    g.tiles = b->d_tiles; g.n_tiles = b->n_tiles; g.rows = b->rows;
    uint16_t* planes = nullptr; unsigned* slots = nullptr;
    const bool split = mode == 1 || mode == 2 || (mode < 0 && gm_gemm_split_ok(b->n_tiles, K, N));
    if (split) {
        GM_REQUIRE((N == 256 || N == 128) && K % 16 == 0 && K >= 32, GM_EINVAL, "dense_update: the split kernels need N = 128 or 256 and K a multiple of 16 (>= 32)");
        const int sets = w_stride ? b->sets : 1;
        GM_TRY(gm_alloc(&planes, (size_t)sets * 3 * K * N, st));
        int rc = GM_OK;
        gm_bound wb = gm_no_bound();
        if (mode == 2) {
            // two fp16 pieces per operand: bounds taken here -- one for all of x (slot 0), one per weight matrix (slots 1..)
            rc = gm_alloc(&slots, (size_t)(sets + 1) * GM_BOUND_PAD, st);
            if (rc == GM_OK && hipMemsetAsync(slots, 0, sizeof(unsigned) * (sets + 1) * GM_BOUND_PAD, st) != hipSuccess) { gm_set_error("dense_update: memset failed"); rc = GM_EHIP; }
            if (rc == GM_OK) rc = gm_amax(x, 0, 0, (int64_t)b->rows * K, 1, slots, 0, st);
            if (rc == GM_OK) rc = gm_amax(W, w_stride, 0, (int64_t)K * N, sets, slots + GM_BOUND_PAD, GM_BOUND_PAD, st);
            wb.amax = slots + GM_BOUND_PAD; wb.stride = w_stride ? GM_BOUND_PAD : 0;
            g.np = 2; g.a_bound = gm_no_bound(); g.a_bound.amax = slots; g.b_bound = wb;
        }
        if (rc == GM_OK) rc = gm_split_weights(W, w_stride, 0, K, N, 0, sets, planes, st, mode == 2 ? 2 : 3, wb);
        if (rc != GM_OK) { gm_dev_free(planes, st); if (slots) gm_dev_free(slots, st); return rc; }
        g.Bsplit = planes; g.bsplit_stride = w_stride ? (int64_t)3 * K * N : 0;
    }
    const int rc = gm_launch_gemm_nn(g, st);
    if (planes) gm_dev_free(planes, st);
    if (slots) gm_dev_free(slots, st);
    gm_batch_mark_use(b, st);
    return rc;
}
