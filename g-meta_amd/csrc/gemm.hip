// Dense feature x W update of GraphConv (learner.py:36,47: torch.matmul(feat, weight)) and its two
// backward GEMMs, grouped by task because every task carries its own fast weights (meta.py:126,151).
// Exact fp32 on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain), so the
// 1e-4 tolerance on logits/meta-grads holds without a reduced-precision path.  MFMA-bound, not HBM-bound
// (43-64 flop/B at the arxiv config): reported as MFMA utilisation, separately from the aggregate.
#include <algorithm>
#include <stdlib.h>
#include "gm_internal.h"
#include <atomic>
#include <mutex>
#include "gemm_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 16
#ifndef GM_GEMM_MODE_DEFAULT
#define GM_GEMM_MODE_DEFAULT 1
#endif
#define AS_LD 20            // 16 + 4 pad (keeps 16-B alignment of every row)

struct GemmK {
    const float* A; int64_t lda; const float* B; int64_t b_stride; int transB; float* C; int64_t ldc; int K, N;
    const float* row_scale; const float* bias; int64_t bias_stride; const float* mask_h; int relu;
    const uint8_t* mask_b; uint8_t* relu_bits;
    const int32_t* tiles; int n_tiles; int n_col_tiles; int a_vec, b_vec, c_vec; int nt_store;
    float* zero_out;            // DMA kernels only: the epilogue also zero-fills this [rows, ldc] buffer (dQ of the backward pass that follows)
};

// Block tile 128 x (64*WC); 2 x WC waves, each wave a 64 x 64 sub-tile = 2 x 2 MFMA 32x32 blocks.
template <int WC, bool VEC, bool TB>
__global__ __launch_bounds__(128 * WC) void k_gemm_nn(GemmK g) {
    constexpr int NT = 128 * WC, BN = 64 * WC, BS_LD = BN + 4;
    constexpr int EP_LD = 68;                                      // epilogue staging: 32 rows x 64 cols (+4 pad) per wave
    constexpr int TILE_FLOATS = GM_GEMM_BM * AS_LD + BK * BS_LD, EPI_FLOATS = 2 * WC * 32 * EP_LD;
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS > EPI_FLOATS ? 2 * TILE_FLOATS : EPI_FLOATS];
    // XCD-aware: hardware block b -> XCD b%8; make logical ids contiguous per XCD so that the column
    // tiles of one row tile (which share the A rows) and neighbouring row tiles share an L2.
    const int nb = g.n_tiles * g.n_col_tiles, b = blockIdx.x;
    const int q = nb / GM_NXCD, r = nb % GM_NXCD, xcd = b % GM_NXCD, idx = b / GM_NXCD;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = lb / g.n_col_tiles, ct = lb % g.n_col_tiles;
    const int set = g.tiles[tile * 3], row0 = g.tiles[tile * 3 + 1], nrows = g.tiles[tile * 3 + 2];
    const int n0 = ct * BN;
    const float* Bp = g.B + (int64_t)set * g.b_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, kh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int A_PER = (GM_GEMM_BM * BK / 4 + NT - 1) / NT;   // float4 per thread for the A tile
    constexpr int B_PER = (BK * BN / 4) / NT;                     // == 2
    float4 ra[A_PER], rb[B_PER];
    unsigned okA = 0, okB = 0;       // validity bits, applied when the registers are written to LDS (after the MFMAs), so
                                     // the loads below are unconditional (clamped addresses), branch-free and stay in flight
    auto ld4 = [&](const float* base, int64_t off, int64_t lim) -> float4 {   // 4 floats at base[off..off+3], reads clamped to < lim
        if (VEC) return *reinterpret_cast<const float4*>(base + off);
        float4 v;
        v.x = base[min(off + 0, lim - 1)]; v.y = base[min(off + 1, lim - 1)]; v.z = base[min(off + 2, lim - 1)]; v.w = base[min(off + 3, lim - 1)];
        return v;
    };
    auto load_tiles = [&](int k0) {
        okA = 0; okB = 0;
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int id = min(tid + p * NT, GM_GEMM_BM * BK / 4 - 1);
            const int rr = id >> 2, c4 = (id & 3) * 4;
            const int kc = VEC ? min(k0 + c4, g.K - 4) : min(k0 + c4, g.K - 1);
            const int64_t off = (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + kc;
            ra[p] = ld4(g.A, off, (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + g.K);
            if (rr < nrows && k0 + c4 < g.K) okA |= 1u << p;
        }
#pragma unroll
        for (int p = 0; p < B_PER; ++p) {
            const int id = tid + p * NT;
            if (!TB) {
                const int kk = id / (BN / 4), n = n0 + (id % (BN / 4)) * 4;
                const int kr = min(k0 + kk, g.K - 1), nc = VEC ? min(n, g.N - 4) : min(n, g.N - 1);
                rb[p] = ld4(Bp, (int64_t)kr * g.N + nc, (int64_t)kr * g.N + g.N);
                if (k0 + kk < g.K && n < g.N) okB |= 1u << p;
            } else {   // B[k][n] = W[n][k], W stored [N][K]: read along k
                const int n = n0 + (id >> 2), c4 = (id & 3) * 4;
                const int nr_ = min(n, g.N - 1), kc = VEC ? min(k0 + c4, g.K - 4) : min(k0 + c4, g.K - 1);
                rb[p] = ld4(Bp, (int64_t)nr_ * g.K + kc, (int64_t)nr_ * g.K + g.K);
                if (n < g.N && k0 + c4 < g.K) okB |= 1u << p;
            }
        }
    };
    auto kmask = [&](float4 v, int kbase, int lim) -> float4 {        // zero the lanes of a k-vector that run past K (scalar path only)
        if (VEC) return v;
        if (kbase + 1 >= lim) v.y = 0.f;
        if (kbase + 2 >= lim) v.z = 0.f;
        if (kbase + 3 >= lim) v.w = 0.f;
        return v;
    };
    auto store_tiles = [&](int buf, int k0) {
        float* As = smem + buf * TILE_FLOATS;
        float* Bs = As + GM_GEMM_BM * AS_LD;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int id = tid + p * NT;
            const float4 v = (okA >> p) & 1u ? kmask(ra[p], k0 + (id & 3) * 4, g.K) : z;
            if (id < GM_GEMM_BM * BK / 4) *reinterpret_cast<float4*>(&As[(id >> 2) * AS_LD + (id & 3) * 4]) = v;
        }
#pragma unroll
        for (int p = 0; p < B_PER; ++p) {
            const int id = tid + p * NT;
            if (!TB) {
                const float4 v = (okB >> p) & 1u ? kmask(rb[p], n0 + (id % (BN / 4)) * 4, g.N) : z;
                *reinterpret_cast<float4*>(&Bs[(id / (BN / 4)) * BS_LD + (id % (BN / 4)) * 4]) = v;
            } else {
                const int n = id >> 2, c4 = (id & 3) * 4;
                const float4 v = (okB >> p) & 1u ? kmask(rb[p], k0 + c4, g.K) : z;
                Bs[(c4 + 0) * BS_LD + n] = v.x; Bs[(c4 + 1) * BS_LD + n] = v.y;
                Bs[(c4 + 2) * BS_LD + n] = v.z; Bs[(c4 + 3) * BS_LD + n] = v.w;
            }
        }
    };

    // double-buffered LDS: chunk k+1 is written to the other buffer while chunk k is consumed -> one barrier per chunk
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < g.K;
        if (more) load_tiles(k0 + BK);                 // global loads for the next chunk fly under the MFMAs
        const float* As = smem + buf * TILE_FLOATS;
        const float* Bs = As + GM_GEMM_BM * AS_LD;
        // A fragments: lane (li,kh) takes k = 8q + 4kh + r  (r = 0..3) of its row -> MFMA step 4q + r
        float4 af[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                af[i][qq] = *reinterpret_cast<const float4*>(&As[(wr * 64 + i * 32 + li) * AS_LD + qq * 8 + kh * 4]);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int kk = qq * 8 + kh * 4 + rr;
                const float b0 = Bs[kk * BS_LD + wc * 64 + li], b1 = Bs[kk * BS_LD + wc * 64 + 32 + li];
                const float a0 = rr == 0 ? af[0][qq].x : rr == 1 ? af[0][qq].y : rr == 2 ? af[0][qq].z : af[0][qq].w;
                const float a1 = rr == 0 ? af[1][qq].x : rr == 1 ? af[1][qq].y : rr == 2 ? af[1][qq].z : af[1][qq].w;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (more) store_tiles(buf ^ 1, k0 + BK);      // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5): each lane holds one
    // column, so a direct store would be 64 dword stores per lane (128-B runs, store-issue bound).  Stage each wave's
    // 32x64 half through LDS instead and store whole 256-B row segments as float4 (16 stores per lane).
    const float* biasp = g.bias ? g.bias + (int64_t)set * g.bias_stride : nullptr;
    float* E = smem + wave * (32 * EP_LD);
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int col = n0 + wc * 64 + ec;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (biasp) {
        if (col + 0 < g.N) b4.x = biasp[col + 0];
        if (col + 1 < g.N) b4.y = biasp[col + 1];
        if (col + 2 < g.N) b4.z = biasp[col + 2];
        if (col + 3 < g.N) b4.w = biasp[col + 3];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * kh) * EP_LD + j * 32 + li] = acc[i][j][e];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = wr * 64 + i * 32 + it * 4 + er;
            if (rl >= nrows || col >= g.N) continue;
            const int64_t row = row0 + rl;
            const float sc = g.row_scale ? g.row_scale[row] : 1.f;
            float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
            v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
            if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }   // NaN propagates like torch relu
            float* dst = g.C + row * g.ldc + col;
            if (g.c_vec) {
                if (g.mask_b) {
                    const unsigned m = g.mask_b[(row * g.ldc + col) >> 2];
                    v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
                } else if (g.mask_h) {
                    const float4 m = *reinterpret_cast<const float4*>(g.mask_h + row * g.ldc + col);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
                if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (col + k >= g.N) break;
                    float o = vv[k];
                    if (g.mask_h && !(g.mask_h[row * g.ldc + col + k] > 0.f)) o = 0.f;
                    dst[k] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct-to-LDS variant of the forward GEMM (C = epi(A W_t), W row-major [K,N], K % 16 == 0, N % (64*WC) == 0):
// tiles are DMA-ed HBM -> LDS with global_load_lds_dwordx4 (no staging registers), three LDS stages, and a COUNTED
// s_waitcnt vmcnt so that the loads of chunks k+1 and k+2 stay in flight across the single barrier of chunk k.
// LDS images are lane-linear (the DMA writes wave-uniform base + lane*16 B): A tile [128][16] floats unpadded (the
// b128 fragment reads are 4-way bank conflicted -- irrelevant next to 64-cycle f32 MFMAs), B tile [16][64*WC].
template <int WC>
__global__ __launch_bounds__(128 * WC) void k_gemm_glds(GemmK g) {
    constexpr int NW = 2 * WC, BN = 64 * WC;
    constexpr int A_FLOATS = GM_GEMM_BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
    constexpr int EP_LD = 68, EPI_FLOATS = NW * 32 * EP_LD;
    constexpr int MAIN_FLOATS = 3 * STAGE > EPI_FLOATS ? 3 * STAGE : EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS + GM_GEMM_BM];      // + the tile's 128 row scales
    const int nb = g.n_tiles * g.n_col_tiles, b = blockIdx.x;
    const int q = nb / GM_NXCD, r = nb % GM_NXCD, xcd = b % GM_NXCD, idx = b / GM_NXCD;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = lb / g.n_col_tiles, ct = lb % g.n_col_tiles;
    const int set = g.tiles[tile * 3], row0 = g.tiles[tile * 3 + 1], nrows = g.tiles[tile * 3 + 2];
    const int n0 = ct * BN;
    const float* Bp = g.B + (int64_t)set * g.b_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The epilogue's per-row scales (norm) are fetched now and parked in LDS: as dependent loads inside the epilogue they
    // cost 16 exposed L2 round trips per tile.
    float my_scale = 1.f;
    if (tid < GM_GEMM_BM && g.row_scale) my_scale = g.row_scale[row0 + min(tid, nrows - 1)];
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, kh = lane >> 5;
    // DMA pieces of 1 KiB (64 lanes x 16 B).  A tile = 8 pieces (16 rows x 16 floats each), B tile = BK*BN*4/1024 pieces.
    constexpr int A_PIECES = A_FLOATS / 256, B_PIECES = B_FLOATS / 256, PIECES = A_PIECES + B_PIECES;
    constexpr int PPW = (PIECES + NW - 1) / NW;                         // pieces per wave per chunk
    static_assert(PIECES % NW == 0, "tile pieces must divide evenly over the waves");
    // per-lane source pointers of this wave's pieces (advance by k0 / k0*N per chunk)
    const float* src[PPW]; int dstoff[PPW]; int64_t kstep[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int piece = wave * PPW + p;
        if (piece < A_PIECES) {                                        // rows piece*16 .. +15, lane -> (row, k4)
            const int rr = piece * 16 + (lane >> 2);
            src[p] = g.A + (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + (lane & 3) * 4;
            kstep[p] = BK; dstoff[p] = piece * 256;
        } else {                                                       // B: piece -> 256 consecutive floats of the [16][BN] tile
            const int e = (piece - A_PIECES) * 256 + lane * 4;
            const int kk = e / BN, n = e % BN;
            src[p] = Bp + (int64_t)kk * g.N + n0 + n;
            kstep[p] = (int64_t)BK * g.N; dstoff[p] = A_FLOATS + (piece - A_PIECES) * 256;
        }
    }
    // The DMA is issued through inline asm: with the builtin, hipcc cannot prove that the (dynamic) stage being filled
    // does not alias the stage being read and drains vmcnt(0) before the first ds_read, which serialises the pipeline.
    // An asm statement is outside its vmcnt bookkeeping; completion is counted by hand below (cdna_hip_programming.md 5.7).
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
    auto issue = [&](int chunk) {
        const unsigned st = lds_base + (unsigned)((chunk % 3) * STAGE * 4);
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const float* gsrc = src[p] + chunk * kstep[p];
            const unsigned dst = __builtin_amdgcn_readfirstlane(st + (unsigned)dstoff[p] * 4u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nchunks = g.K / BK;
    issue(0);
    if (nchunks > 1) issue(1);
    if (tid < GM_GEMM_BM) smem[MAIN_FLOATS + tid] = my_scale;      // visible to everyone after the first barrier of the main loop
    for (int c = 0; c < nchunks; ++c) {
        // this wave's pieces of chunk c have landed once at most the PPW pieces of chunk c+1 are still outstanding
        if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's pieces of chunk c landed; everyone is done reading chunk c-1
        if (c + 2 < nchunks) issue(c + 2);             // into the stage chunk c-1 occupied
        const float* As = smem + (c % 3) * STAGE;
        const float* Bs = As + A_FLOATS;
        float4 af[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                af[i][qq] = *reinterpret_cast<const float4*>(&As[(wr * 64 + i * 32 + li) * BK + qq * 8 + kh * 4]);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int kk = qq * 8 + kh * 4 + rr;
                const float b0 = Bs[kk * BN + wc * 64 + li], b1 = Bs[kk * BN + wc * 64 + 32 + li];
                const float a0 = rr == 0 ? af[0][qq].x : rr == 1 ? af[0][qq].y : rr == 2 ? af[0][qq].z : af[0][qq].w;
                const float a1 = rr == 0 ? af[1][qq].x : rr == 1 ? af[1][qq].y : rr == 2 ? af[1][qq].z : af[1][qq].w;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // all MFMAs' LDS reads done before the epilogue reuses the stages
    const float* biasp = g.bias ? g.bias + (int64_t)set * g.bias_stride : nullptr;
    float* E = smem + wave * (32 * EP_LD);
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int col = n0 + wc * 64 + ec;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (biasp) b4 = *reinterpret_cast<const float4*>(biasp + col);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * kh) * EP_LD + j * 32 + li] = acc[i][j][e];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = wr * 64 + i * 32 + it * 4 + er;
            if (rl >= nrows) continue;
            const int64_t row = row0 + rl;
            const float sc = smem[MAIN_FLOATS + rl];
            float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
            v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
            if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
            if (g.mask_h) {
                const float4 m = *reinterpret_cast<const float4*>(g.mask_h + row * g.ldc + col);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
            if (g.nt_store) {     // C is re-read by the NEXT kernel only (>> L2): keep L2 for the A rows and the weights
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
            } else *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
            if (g.zero_out) *reinterpret_cast<float4*>(g.zero_out + row * g.ldc + col) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// Small-launch variant of k_gemm_glds<1> (tile 128 x 64): EIGHT waves of 32 x 32 instead of two of 64 x 64.  A launch with
// a few hundred tiles (support batches of a 4-task shard, the Tissue / FirstMM shapes) puts about one workgroup on a CU, where
// the two-wave version walks a chain of K/2 * 4 dependent-issue MFMAs (64 cycles each) with one wave per SIMD and nothing to
// cover the per-chunk LDS latency: 34 us for a 14-us chain.  Same tiles, same DMA pipeline, a quarter of the MFMA chain per wave.
__global__ __launch_bounds__(512) void k_gemm_glds_small(GemmK g) {
    constexpr int NW = 8, BN = 64;
    constexpr int A_FLOATS = GM_GEMM_BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;         // 2048 + 1024 floats
    constexpr int EP_LD = 36, EPI_FLOATS = NW * 32 * EP_LD;                                              // == 3 * STAGE
    constexpr int MAIN_FLOATS = 3 * STAGE > EPI_FLOATS ? 3 * STAGE : EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS + 256 + GM_GEMM_BM];                 // + one dummy DMA piece + the row scales
    const int nb = g.n_tiles * g.n_col_tiles, b = blockIdx.x;
    const int q = nb / GM_NXCD, r = nb % GM_NXCD, xcd = b % GM_NXCD, idx = b / GM_NXCD;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = lb / g.n_col_tiles, ct = lb % g.n_col_tiles;
    const int set = g.tiles[tile * 3], row0 = g.tiles[tile * 3 + 1], nrows = g.tiles[tile * 3 + 2];
    const int n0 = ct * BN;
    const float* Bp = g.B + (int64_t)set * g.b_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float my_scale = 1.f;
    if (tid < GM_GEMM_BM && g.row_scale) my_scale = g.row_scale[row0 + min(tid, nrows - 1)];
    const int wr = wave >> 1, wc = wave & 1;                               // 4 x 2 grid of 32 x 32 sub-tiles
    const int li = lane & 31, kh = lane >> 5;
    // 12 one-KiB DMA pieces per chunk (8 of A, 4 of B) over 8 waves: two slots per wave, the last four slots re-read A piece 0
    // into a scratch KiB so that every wave issues exactly two loads per chunk (the vmcnt wait is a fixed count)
    constexpr int A_PIECES = 8, PIECES = 12, PPW = 2;
    const float* src[PPW]; int dstoff[PPW]; int64_t kstep[PPW]; bool dummy[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int piece = wave * PPW + p;
        dummy[p] = piece >= PIECES;
        if (piece < A_PIECES || dummy[p]) {
            const int pc = dummy[p] ? 0 : piece, rr = pc * 16 + (lane >> 2);
            src[p] = g.A + (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + (lane & 3) * 4;
            kstep[p] = BK; dstoff[p] = pc * 256;
        } else {
            const int e = (piece - A_PIECES) * 256 + lane * 4, kk = e / BN, n = e % BN;
            src[p] = Bp + (int64_t)kk * g.N + n0 + n;
            kstep[p] = (int64_t)BK * g.N; dstoff[p] = A_FLOATS + (piece - A_PIECES) * 256;
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
    auto issue = [&](int chunk) {
        const unsigned st = lds_base + (unsigned)((chunk % 3) * STAGE * 4);
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
            const float* gsrc = src[p] + chunk * kstep[p];
            const unsigned dst = __builtin_amdgcn_readfirstlane(dummy[p] ? lds_base + (unsigned)MAIN_FLOATS * 4u : st + (unsigned)dstoff[p] * 4u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nchunks = g.K / BK;
    issue(0);
    if (nchunks > 1) issue(1);
    if (tid < GM_GEMM_BM) smem[MAIN_FLOATS + 256 + tid] = my_scale;
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 2 < nchunks) issue(c + 2);
        const float* As = smem + (c % 3) * STAGE;
        const float* Bs = As + A_FLOATS;
        float4 af[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) af[qq] = *reinterpret_cast<const float4*>(&As[(wr * 32 + li) * BK + qq * 8 + kh * 4]);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int kk = qq * 8 + kh * 4 + rr;
                const float b0 = Bs[kk * BN + wc * 32 + li];
                const float a0 = rr == 0 ? af[qq].x : rr == 1 ? af[qq].y : rr == 2 ? af[qq].z : af[qq].w;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    const float* biasp = g.bias ? g.bias + (int64_t)set * g.bias_stride : nullptr;
    float* E = smem + wave * (32 * EP_LD);
    const int er = lane >> 3, ec = (lane & 7) * 4;                          // 8 rows x 32 columns per pass
    const int col = n0 + wc * 32 + ec;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (biasp) b4 = *reinterpret_cast<const float4*>(biasp + col);
#pragma unroll
    for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * kh) * EP_LD + li] = acc[e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = wr * 32 + it * 8 + er;
        if (rl >= nrows) continue;
        const int64_t row = row0 + rl;
        const float sc = smem[MAIN_FLOATS + 256 + rl];
        float4 v = *reinterpret_cast<const float4*>(&E[(it * 8 + er) * EP_LD + ec]);
        v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
        if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
        if (g.mask_h) {
            const float4 m = *reinterpret_cast<const float4*>(g.mask_h + row * g.ldc + col);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
        if (g.nt_store) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
        } else *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
        if (g.zero_out) *reinterpret_cast<float4*>(g.zero_out + row * g.ldc + col) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- split-bf16 path (gemm_split.h): mode switch, eligibility, weight planes
static std::atomic<int> g_gemm_mode{-1};        // gm_set_gemm_mode override (-1: the environment / library default)
int gm_gemm_mode() {
    const int o = g_gemm_mode.load(std::memory_order_relaxed);
    if (o >= 0) return o;
    const int e = gm_knob().gemm_mode;
    return e >= 0 ? e : GM_GEMM_MODE_DEFAULT;
}
extern "C" void gm_set_gemm_mode(int32_t mode) { g_gemm_mode.store(mode ? 1 : 0, std::memory_order_relaxed); }
extern "C" int32_t gm_get_gemm_mode(void) { return gm_gemm_mode(); }
static std::atomic<int> g_split_pieces{-1};     // gm_set_split_pieces override (-1: GM_SPLIT_PIECES / default 3)
extern "C" void gm_set_split_pieces(int32_t pieces) { g_split_pieces.store(pieces == 3 ? 3 : (pieces == 2 ? 2 : -1), std::memory_order_relaxed); }
extern "C" int32_t gm_get_split_pieces(void) { return gm_split_np(); }
// The persistent kernel walks 128 x 256 tiles, one workgroup per CU: worth it from about one tile per CU upwards.
const float* gm_zero_row(hipStream_t s) {
    static std::mutex mu;
    static float* rows[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!rows[dev]) {
        float* p = nullptr;
        if (hipMalloc(&p, 4096 * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(p, 0, 4096 * sizeof(float), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(p); return nullptr; }
        rows[dev] = p;
    }
    return rows[dev];
}
bool gm_gemm_split_ok(int n_tiles, int K, int N) {
    // default: from a quarter of the (current device's) CUs busy upwards (measured on the 141-tile support batch of a 4-task shard: still ahead of the fp32 small-tile kernel)
    const int min_tiles = gm_knob().gemm_split_min_tiles >= 0 ? gm_knob().gemm_split_min_tiles : gm_num_cus() / 4;
    return gm_gemm_mode() == 1 && (N == 256 || N == 128) && K % 16 == 0 && K >= 32 && n_tiles >= min_tiles;
}
int gm_split_weights(const float* params, int64_t pstride, int64_t w_off, int K, int N, int trans, int sets, uint16_t* out, hipStream_t s, int np, gm_bound bound) {
    GM_REQUIRE(np == 3 || (np == 2 && bound.amax), GM_EINVAL, "split_weights: two-piece planes need a bound");
    hipLaunchKernelGGL(k_split_w, dim3((K + 31) / 32, (N + 31) / 32, sets), dim3(256), 0, s, params, pstride, w_off, K, N, trans, out, np, bound);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
int gm_amax(const float* x, int64_t stride, int64_t off, int64_t n, int sets, unsigned* out, int64_t out_stride, hipStream_t s) {
    if (n <= 0 || sets <= 0) return GM_OK;
    const int bx = (int)std::min<int64_t>(64, (n + 2047) / 2048);
    hipLaunchKernelGGL(k_amax, dim3(bx, sets), dim3(256), 0, s, x, stride, off, n, out, out_stride);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
int gm_amax_segs(const float* x, const int64_t* off, const int64_t* n, int segs, unsigned* out, int64_t out_stride, hipStream_t s) {
    GM_REQUIRE(segs >= 1 && segs <= 8, GM_EINVAL, "amax_segs: 1..8 segments");
    AmaxSegs sg{}; int64_t nmax = 1;
    for (int i = 0; i < segs; ++i) { sg.off[i] = off[i]; sg.n[i] = n[i]; nmax = std::max(nmax, n[i]); }
    hipLaunchKernelGGL(k_amax_segs, dim3((int)std::min<int64_t>(32, (nmax + 2047) / 2048), segs), dim3(256), 0, s, x, sg, out, out_stride);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
int gm_split_np() {
    const int o = g_split_pieces.load(std::memory_order_relaxed);
    return o > 0 ? o : (gm_knob().split_pieces == 2 ? 2 : 3);
}

static int launch_gemm_nn(const gm_gemm_args& a, hipStream_t s);
int gm_launch_gemm_nn(const gm_gemm_args& a, hipStream_t s) {
    if (a.n_tiles <= 0) return GM_OK;
    const int cat = a.Bsplit ? (a.np == 2 ? GM_PROF_GEMM_SPLIT16 : GM_PROF_GEMM_SPLIT) : GM_PROF_GEMM;          // the pipe the launch runs on (bench.py prices each on its own peak)
    gm_prof_begin(cat, s, 2 * a.rows * a.K * a.N);
    if (a.Bsplit) gm_prof_note(GM_PROF_GEMM_SPLIT_BYTES, 4 * a.rows * (int64_t)a.K + 4 * (a.row_scale_keep ? a.n_keep : a.rows) * (int64_t)a.N);      // compulsory A + C bytes
    gm_prof_note(GM_PROF_GEMM_BYTES, 4 * a.rows * (int64_t)a.K + 4 * ((a.Bsplit && a.row_scale_keep) ? a.n_keep : a.rows) * (int64_t)a.N);
    const int rc = launch_gemm_nn(a, s);
    gm_prof_end(cat, s);
    return rc;
}
static int launch_gemm_nn(const gm_gemm_args& a, hipStream_t s) {
    if (a.Bsplit) {
        // fp32-accurate product on the bf16 matrix cores (exact 3-way operand split, 6 MFMA products, fp32 accumulation)
        const bool ok = (a.N == 256 || a.N == 128) && a.K % 16 == 0 && a.K >= 32 && !a.mask_h && !a.mask_b && (a.lda % 4 == 0) && (((uintptr_t)a.A & 15) == 0) &&
                        (a.ldc % 4 == 0) && (((uintptr_t)a.C & 15) == 0) && (!a.bias || a.bias_stride % 4 == 0);
        GM_REQUIRE(ok, GM_EINVAL, "gemm: launch not eligible for the split-bf16 kernel (N=%d K=%d)", a.N, a.K);
        SplitGemmK k{};
        k.A = a.A; k.lda = a.lda; k.Bt = a.Bsplit; k.bt_stride = a.bsplit_stride; k.C = a.C; k.ldc = a.ldc; k.K = a.K; k.N = a.N;
        k.row_scale = a.row_scale_keep ? a.row_scale_keep : a.row_scale; k.keep_signed = a.row_scale_keep ? 1 : 0; k.bias = a.bias; k.bias_stride = a.bias_stride; k.relu = a.relu; k.relu_bits = a.relu_bits;
        k.tiles = a.tiles; k.n_tiles = a.n_tiles; k.n_col_tiles = 1; k.nt_store = 1; k.zero_out = a.zero_out;      // (ordinary C stores lose at every size: 4-task shard +3 %, task_num 32 +1.3 %)
        const bool f16 = a.np == 2;
        GM_REQUIRE(!f16 || (a.a_bound.amax && a.b_bound.amax), GM_EINVAL, "gemm: the two-piece split kernel needs bounds for both operands");
        k.a_bound = f16 ? a.a_bound : gm_no_bound(); k.b_bound = f16 ? a.b_bound : gm_no_bound(); k.amax_out = a.amax_out;
        // persistent: one workgroup per CU (it fills the CU's register file, so nothing else co-resides).  GM_GEMM_SPLIT_GRID caps the
        // grid below the CU count, which leaves whole CUs to the kernels of the other stream (experiment knob).
        const int cus = gm_stream_cus(s);                                 // the stream's CU mask, if it has one
        int grid_cap = gm_knob().gemm_split_grid;
        if (grid_cap <= 0 && a.n_tiles > 2 * cus && a.n_tiles <= 6 * cus && cus >= 128) grid_cap = cus - 2 * GM_NXCD;
        // (round 6, measured on the 4-task arxiv shard -- the per-GPU share of an 8-GPU meta-batch, 1,101 query tiles: two CUs per XCD left to the
        // other queue's small kernels, 4.07-4.16 -> 4.00-4.02 ms per meta-step; from ~8 tiles per CU upwards the cap only costs: task_num 8 / 16 even,
        // task_num 32 24.8 -> 25.0 ms; CU-masked streams lose at every split at this size too: profiles/r06_t4_sweep.txt)
        if (grid_cap <= 0 || grid_cap > cus) grid_cap = cus;
        if (a.fuse2) {
            // fused aggregate + GEMM: A addresses the aggregate's input rows, rows of other degrees come finished from a.zside
            GM_REQUIRE(a.K / 16 >= PF_DA && a.zside && (a.ldz % 4 == 0) && (((uintptr_t)a.zside & 15) == 0), GM_EINVAL, "gemm: fused aggregate needs K >= %d and an aligned side buffer", 16 * PF_DA);
            GM_REQUIRE(a.K <= 4096, GM_EINVAL, "gemm: fused aggregate supports K <= 4096");
            k.f2 = reinterpret_cast<const int4*>(a.fuse2); k.zside = a.zside; k.ldz = a.ldz; k.zrow = gm_zero_row(s);
            GM_REQUIRE(k.zrow, GM_EHIP, "gemm: zero row allocation failed");
            // Workgroup lifetime: a strictly persistent grid (one workgroup per CU for the whole launch) holds every CU for ~0.65 ms on the
            // large query launches, and nothing co-resides with it (it fills the register file) -- the support chain's small kernels then wait
            // a whole GEMM for a CU (head/loss: 26 us of work, 680 us in the two-stream timeline).  `mult` workgroups per CU, each walking
            // 1/mult of the tiles, give the dispatcher a yield point every ~0.65/mult ms (default 4: 27.7 -> 27.4 ms at task_num 32, 5.12 -> 4.86 ms
            // for the 4-task shard, where the support chain is the critical path; 8 and more lose to the per-workgroup prologue).
            // With the two-piece kernels (shorter tiles) the optimum moved from 4 to 3, and to 2 for the launches of 16 and more tiles per CU:
            // 4-task shard 4.36 -> 4.25 ms, task_num 32 24.86 -> 24.50 (three runs each, same box).
            const int mult_f = gm_knob().gemm_fused_rounds > 0 ? gm_knob().gemm_fused_rounds : (a.n_tiles >= 16 * grid_cap ? 2 : 3);
            const dim3 grid(std::min(a.n_tiles, mult_f * grid_cap));
            if (a.N == 128 && f16) hipLaunchKernelGGL((k_gemm_split_p<true, 1, 2, 2>), grid, dim3(1024), 0, s, k);
            else if (a.N == 128) hipLaunchKernelGGL((k_gemm_split_p<true, 1, 2, 3>), grid, dim3(1024), 0, s, k);
            else if (f16) hipLaunchKernelGGL((k_gemm_split_p<true, 2, 4, 2>), grid, dim3(1024), 0, s, k);
            else hipLaunchKernelGGL((k_gemm_split_p<true, 2, 4, 3>), grid, dim3(1024), 0, s, k);
        } else {
            // a launch that would leave more than half of the CUs without a tile walks 64-row half tiles: half the MFMA chain per workgroup
            const int half_on = gm_knob().gemm_half_tiles;
            const dim3 grid_r(std::min(a.n_tiles, gm_knob().gemm_plain_rounds * grid_cap));
            if (a.N == 128 && f16) hipLaunchKernelGGL((k_gemm_split_p<false, 1, 2, 2>), grid_r, dim3(1024), 0, s, k);
            else if (a.N == 128) hipLaunchKernelGGL((k_gemm_split_p<false, 1, 2, 3>), grid_r, dim3(1024), 0, s, k);
            else if (half_on && 2 * a.n_tiles <= grid_cap) {
                if (f16) hipLaunchKernelGGL((k_gemm_split_p<false, 1, 4, 2>), dim3(2 * a.n_tiles), dim3(1024), 0, s, k);
                else hipLaunchKernelGGL((k_gemm_split_p<false, 1, 4, 3>), dim3(2 * a.n_tiles), dim3(1024), 0, s, k);
            } else if (f16) hipLaunchKernelGGL((k_gemm_split_p<false, 2, 4, 2>), grid_r, dim3(1024), 0, s, k);
            else hipLaunchKernelGGL((k_gemm_split_p<false, 2, 4, 3>), grid_r, dim3(1024), 0, s, k);
        }
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
    GemmK g{a.A, a.lda, a.B, a.b_stride, a.transB, a.C, a.ldc, a.K, a.N, a.row_scale, a.bias, a.bias_stride, a.mask_h, a.relu,
            a.mask_b, a.relu_bits, a.tiles, a.n_tiles, 0, 0, 0, 0, 0, nullptr};
    g.a_vec = (a.K % 4 == 0) && (a.lda % 4 == 0) && (((uintptr_t)a.A & 15) == 0);
    g.b_vec = (((uintptr_t)a.B & 15) == 0) && (a.b_stride % 4 == 0) && ((a.transB ? a.K : a.N) % 4 == 0);
    g.c_vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && (((uintptr_t)a.C & 15) == 0) && (!a.mask_h || (((uintptr_t)a.mask_h & 15) == 0));
    const bool vec = g.a_vec && g.b_vec && a.K >= 4 && a.N >= 4;
    GM_REQUIRE(!(a.mask_b || a.relu_bits) || g.c_vec, GM_EINVAL, "gemm: packed relu masks need 16-byte aligned C with N %% 4 == 0");
#define GM_LAUNCH_GEMM(WC_, THREADS_)                                                                                         \
    do {                                                                                                                      \
        const dim3 grid(g.n_tiles * g.n_col_tiles), blk(THREADS_);                                                            \
        if (vec && a.transB) hipLaunchKernelGGL((k_gemm_nn<WC_, true, true>), grid, blk, 0, s, g);                             \
        else if (vec) hipLaunchKernelGGL((k_gemm_nn<WC_, true, false>), grid, blk, 0, s, g);                                   \
        else if (a.transB) hipLaunchKernelGGL((k_gemm_nn<WC_, false, true>), grid, blk, 0, s, g);                              \
        else hipLaunchKernelGGL((k_gemm_nn<WC_, false, false>), grid, blk, 0, s, g);                                           \
    } while (0)
    // column-tile width: the widest (A read once) unless that leaves CUs idle -- small batches are latency-bound, so
    // trade A re-reads (L2 hits, same XCD) for parallelism
    const int bn_cap = gm_knob().gemm_bn;
    int bn = a.N > 128 ? 256 : a.N > 64 ? 128 : 64;
    if (bn > bn_cap) bn = bn_cap;
    while (bn > 64 && (int64_t)a.n_tiles * ((a.N + bn - 1) / bn) < 512) bn >>= 1;
    const int mid_tiles = gm_knob().gemm_mid_tiles;           // below this many row tiles a 256-wide tile grid is only a few rounds deep: halve the tile (shorter tail)
    if (bn == 256 && a.n_tiles < mid_tiles) bn = 128;
    g.n_col_tiles = (a.N + bn - 1) / bn;
    const int use_glds = gm_knob().gemm_glds;
    const bool bias_al = !a.bias || ((((uintptr_t)a.bias & 15) == 0) && (a.bias_stride % 4 == 0));
    g.nt_store = gm_knob().gemm_nt;
    const bool dma = use_glds && vec && !a.transB && !a.mask_b && g.c_vec && bias_al && a.K % BK == 0 && a.K >= 2 * BK && a.N % bn == 0;
    if (a.zero_out) {               // honoured on every path: in the DMA kernels' epilogue (needs the whole row range of ldc covered), else a memset
        if (dma && a.N == a.ldc) g.zero_out = a.zero_out;
        else GM_HIP(hipMemsetAsync(a.zero_out, 0, sizeof(float) * (size_t)a.rows * a.ldc, s));
    }
    if (dma) {
        const dim3 grid(g.n_tiles * g.n_col_tiles);
        if (bn == 256) hipLaunchKernelGGL((k_gemm_glds<4>), grid, dim3(512), 0, s, g);
        else if (bn == 128) hipLaunchKernelGGL((k_gemm_glds<2>), grid, dim3(256), 0, s, g);
        else {
            const int small = gm_knob().gemm_small;            // 1 (default): 8 x (32 x 32) waves per 128 x 64 tile when the launch leaves CUs mostly empty
            if (small && (int64_t)g.n_tiles * g.n_col_tiles <= 4 * gm_num_cus()) hipLaunchKernelGGL(k_gemm_glds_small, grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((k_gemm_glds<1>), grid, dim3(128), 0, s, g);
        }
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
    if (bn == 256) GM_LAUNCH_GEMM(4, 512);
    else if (bn == 128) GM_LAUNCH_GEMM(2, 256);
    else GM_LAUNCH_GEMM(1, 128);
#undef GM_LAUNCH_GEMM
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[K,N] = sum_rows a_scale[row] * A[row,:]^T G[row,:],  db[N] = sum_rows Gb[row,:].
// The reduction runs over the (huge, ragged) row dimension, so it is the MFMA k dimension here:
// both operands are read in their natural row-major layout (lane i of a half-wave reads 32
// consecutive floats of one row).  One block per row chunk accumulates the whole K x N tile grid
// in registers (16 waves x up to 4 MFMA 32x32 tiles); chunk partials are reduced in a second,
// deterministic pass (no float atomics).
#define WG_THREADS 1024
#define WG_WAVES 16
#define WG_MAXT 4

struct WgradK {
    const float* A; int64_t lda; int K; const int32_t* a_row;   // optional row indirection for A (feature gather)
    const float* G; int64_t ldg; int N; const float* Gb; int64_t ldgb;
    const float* a_scale; const int32_t* chunks; int n_chunks; float* partial; int RK; int TK, TN; int vec;
    gm_bound a_bound, g_bound;               // k_wgrad_split<., ., 2>: bounds of the A / G rows (two fp16 pieces per operand)
    // k_wgrad_split<., ., ., true> (GA): A = the aggregate Z of a pass whose forward ran FUSED, formed here the way the fused GEMM's feeders formed it:
    // f2[row] = {u0, u1, w0, w1} (gm_batch::d_fuse2 / d_fuse2_feat), row = w0 * gx[u0] + w1 * gx[u1]; entries flagged GM_FUSE_SELF read their finished row
    // of A (the partial aggregate launch wrote it), GM_FUSE_ZERO rows read zrow
    const int4* f2; const float* gx; int64_t ldgx; const float* zrow;
};

#define WG_PF 2     // float4 prefetch registers per thread: RK * (ldA + ldG) <= 2 * 1024 * 4 floats per stage

__global__ __launch_bounds__(WG_THREADS) void k_wgrad(WgradK w) {
    extern __shared__ __attribute__((aligned(16))) float sm_f[];
    float* sm = sm_f;
    // One LDS matrix S[RK][ld]: columns [0, ldA) hold norm-scaled A rows (zero padded to 32), [ldA, ld) hold G rows.
    const int ldA = w.TK * 32, ldG = w.TN * 32, ld = ldA + ldG, ld4 = ld >> 2, ldA4 = ldA >> 2;
    const int chunk = blockIdx.x, zt = blockIdx.y;     // zt: group of 64 output tiles
    const int row0 = w.chunks[chunk * 3 + 1], nrows = w.chunks[chunk * 3 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int TT = w.TK * w.TN;
    f32x16 acc[WG_MAXT];
#pragma unroll
    for (int t = 0; t < WG_MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum[2] = {0.f, 0.f};                         // db[tid + 1024*j] (only zt == 0; N <= 2048)
    float4 pf[WG_PF]; float pfs[WG_PF];                 // prefetched values and their (deferred) row scales
    const int per_stage4 = w.RK * ld4;                   // float4 elements per stage (<= WG_PF * 512)

    unsigned okS = 0;
    auto load_stage = [&](int r0) {                       // unconditional, clamped loads; validity applied in store_stage
        const int nr = min(w.RK, nrows - r0);
        okS = 0;
#pragma unroll
        for (int p = 0; p < WG_PF; ++p) {
            const int id = min(tid + p * WG_THREADS, per_stage4 - 1);
            const int rr = id / ld4, c4 = id - rr * ld4;
            const int64_t row = row0 + r0 + min(rr, nr - 1);
            const bool isA = c4 < ldA4;
            const int col = isA ? c4 * 4 : (c4 - ldA4) * 4;
            const int lim = isA ? w.K : w.N;
            const int64_t ar = (isA && w.a_row) ? w.a_row[row] : row;
            const float* base = isA ? w.A + ar * w.lda : w.G + row * w.ldg;
            float4 v;
            if (w.vec) v = *reinterpret_cast<const float4*>(base + min(col, lim - 4));
            else { v.x = base[min(col, lim - 1)]; v.y = base[min(col + 1, lim - 1)]; v.z = base[min(col + 2, lim - 1)]; v.w = base[min(col + 3, lim - 1)]; }
            pf[p] = v;
            pfs[p] = (isA && w.a_scale) ? w.a_scale[row] : 1.f;
            // validity bits: bit p = whole element valid; bits 8+4p.. = lanes y,z,w inside the column range
            unsigned ok = (tid + p * WG_THREADS < per_stage4 && rr < nr && col < lim) ? 1u : 0u;
            okS |= ok << p;
            okS |= ((col + 1 < lim ? 1u : 0u) | (col + 2 < lim ? 2u : 0u) | (col + 3 < lim ? 4u : 0u)) << (8 + 4 * p);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int p = 0; p < WG_PF; ++p) {
            const int id = tid + p * WG_THREADS;
            const float sc = (okS >> p) & 1u ? pfs[p] : 0.f;
            const unsigned lm = (okS >> (8 + 4 * p)) & 7u;
            const float4 v = make_float4(pf[p].x * sc, (lm & 1u) ? pf[p].y * sc : 0.f, (lm & 2u) ? pf[p].z * sc : 0.f, (lm & 4u) ? pf[p].w * sc : 0.f);
            if (id < per_stage4) *reinterpret_cast<float4*>(&sm[id * 4]) = v;
        }
    };

    // fragment base pointers of this wave's output tiles (inactive tiles read tile 0 harmlessly)
    int apT[WG_MAXT], gpT[WG_MAXT], actT = 0;          // LDS float offsets
#pragma unroll
    for (int t = 0; t < WG_MAXT; ++t) {
        const int tt = zt * (WG_WAVES * WG_MAXT) + t * WG_WAVES + wave;
        const int tq = tt < TT ? tt : 0;
        const int tk = tq / w.TN, tn = tq - tk * w.TN;
        apT[t] = tk * 32 + li + kh * ld; gpT[t] = ldA + tn * 32 + li + kh * ld;
        if (tt < TT) actT |= 1 << t;
    }
    load_stage(0);
    store_stage();
    __syncthreads();
    for (int r0 = 0; r0 < nrows; r0 += w.RK) {
        const bool more = r0 + w.RK < nrows;
        if (more) load_stage(r0 + w.RK);                 // next stage's HBM reads fly under this stage's MFMAs
        if (zt == 0) {
            if (!w.Gb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = tid + j * WG_THREADS;
                    if (n < w.N) for (int rr = 0; rr < w.RK; ++rr) bsum[j] += sm[rr * ld + ldA + n];
                }
            } else {
                const int nr = min(w.RK, nrows - r0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = tid + j * WG_THREADS;
                    if (n < w.N) for (int rr = 0; rr < nr; ++rr) bsum[j] += w.Gb[(int64_t)(row0 + r0 + rr) * w.ldgb + n];
                }
            }
        }
        // all fragments of one k-pair are read first (independent ds_reads), then the wave's MFMAs issue back to back
#pragma unroll 1
        for (int kk = 0; kk < w.RK; kk += 2) {
            float av[WG_MAXT], gv[WG_MAXT];
#pragma unroll
            for (int t = 0; t < WG_MAXT; ++t) { av[t] = sm[apT[t] + kk * ld]; gv[t] = sm[gpT[t] + kk * ld]; }
#pragma unroll
            for (int t = 0; t < WG_MAXT; ++t)
                if (actT & (1 << t)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], gv[t], acc[t], 0, 0, 0);
        }
        __syncthreads();
        if (more) { store_stage(); __syncthreads(); }
    }
    float* out = w.partial + (int64_t)chunk * (w.K + 1) * w.N;
#pragma unroll
    for (int t = 0; t < WG_MAXT; ++t) {
        const int tt = zt * (WG_WAVES * WG_MAXT) + t * WG_WAVES + wave;
        if (tt < TT) {
            const int tk = tt / w.TN, tn = tt - tk * w.TN;
            const int n = tn * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = tk * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (k < w.K && n < w.N) out[(int64_t)k * w.N + n] = acc[t][e];
            }
        }
    }
    if (zt == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int n = tid + j * WG_THREADS; if (n < w.N) out[(int64_t)w.K * w.N + n] = bsum[j]; }
    }
}

// Specialised weight-gradient kernel: K = 32*TK, N = 32*TN known at compile time (the hidden sizes 64/128/256 of
// the reference's configs), vectorised operands, no row indirection.  All per-thread staging coordinates are
// stage-invariant, nothing spills, and the only waits on the prefetch loads sit after the MFMA loop.
#ifndef GM_WG_SPLIT_NUM
#define GM_WG_SPLIT_NUM 4      // eighths of a stage computed before the LDS store of the next one
#endif
template <int TK, int TN, int ZS>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_fast(WgradK w) {
    constexpr int ldA = TK * 32, ldG = TN * 32, ld = ldA + ldG, ld4 = ld / 4;
    // stage = RK rows: its MFMA phase must outlast a loaded HBM round trip for the register prefetch of the next stage to
    // land in time -- 32 rows at ld <= 512 (two 64-KiB LDS buffers, one workgroup per CU anyway)
    constexpr int RK = (16384 / ld) >= 32 ? 32 : (16384 / ld) >= 16 ? 16 : 8;
    constexpr int PER4 = RK * ld4, PF = (PER4 + WG_THREADS - 1) / WG_THREADS;
    constexpr int TT = TK * TN, TPW = (TT + WG_WAVES - 1) / WG_WAVES;
    static_assert(PF <= 4 && TPW <= 4 && TPW % ZS == 0, "tile grid too large for the fast weight-gradient kernel");
    // gridDim.y = ZS splits this chunk's output tiles over ZS workgroups (each does 1/ZS of the MFMAs): used when a batch
    // has too few row chunks to fill the chip.  A wave's NACC active tiles are t = j*ZS + zs (compile-time count).
    constexpr int NACC = TPW / ZS;
    constexpr int BUF = RK * ld;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int zs = ZS > 1 ? (int)blockIdx.y : 0;
    const int chunk = blockIdx.x;
    const int row0 = w.chunks[chunk * 3 + 1], nrows = w.chunks[chunk * 3 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    int aoff[NACC], goff[NACC], tile_id[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int tt = (t * ZS + zs) * WG_WAVES + wave;
        tile_id[t] = tt < TT ? tt : -1;
        const int tq = tt < TT ? tt : 0;                       // inactive slots compute on tile 0 and are never written
        aoff[t] = kh * ld + (tq / TN) * 32 + li; goff[t] = kh * ld + ldA + (tq % TN) * 32 + li;
    }
    // stage-invariant staging coordinates of this thread's PF float4 slots
    int rrp[PF], colp[PF]; bool isA[PF], inr[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int id = tid + p * WG_THREADS;
        inr[p] = id < PER4;
        const int idc = inr[p] ? id : 0;
        rrp[p] = idc / ld4;
        const int c4 = idc % ld4;
        isA[p] = c4 < ldA / 4;
        colp[p] = isA[p] ? c4 * 4 : (c4 - ldA / 4) * 4;
    }
    float4 pf[PF]; float pfs[PF];
    float bsum = 0.f;
    int nr_loaded = 0;
    // load_stage only ISSUES loads (clamped rows, no select on a loaded value, no divergent branch), so nothing forces a
    // vmcnt wait before the MFMA loop; A/G selection, the norm scale and the row-validity zeroing happen at LDS-store time.
    auto load_stage = [&](int r0) {
        nr_loaded = min(RK, nrows - r0);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int64_t row = row0 + r0 + min(rrp[p], nr_loaded - 1);
            const float* src = isA[p] ? w.A + row * w.lda + colp[p] : w.G + row * w.ldg + colp[p];
            pf[p] = *reinterpret_cast<const float4*>(src);
            if (w.a_scale) pfs[p] = w.a_scale[row];            // wave-uniform branch (kernel argument)
        }
    };
    auto store_stage_to = [&](float* dst) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            float sc = (isA[p] && w.a_scale) ? pfs[p] : 1.f;
            if (rrp[p] >= nr_loaded) sc = 0.f;                  // rows past the chunk end contribute zeros
            if (inr[p]) *reinterpret_cast<float4*>(&dst[(tid + p * WG_THREADS) * 4]) = make_float4(pf[p].x * sc, pf[p].y * sc, pf[p].z * sc, pf[p].w * sc);
        }
    };
    // Pipeline per stage, LDS double-buffered: issue HBM loads of stage s+1 -> first half of the MFMAs of stage s ->
    // registers of s+1 -> LDS buffer (s+1)&1 -> second half of the MFMAs -> ONE barrier.
    load_stage(0);
    store_stage_to(sm);
    __syncthreads();
    int cur = 0;
    for (int r0 = 0; r0 < nrows; r0 += RK, cur ^= 1) {
        const bool more = r0 + RK < nrows;
        const float* S = sm + cur * BUF;
        if (more) load_stage(r0 + RK);
        if (zs == 0 && tid < ldG) {
#pragma unroll
            for (int rr = 0; rr < RK; ++rr) bsum += S[rr * ld + ldA + tid];
        }
        // the prefetched registers go to LDS after SPLIT of the stage's RK rows: the later, the longer the HBM round trip may take
        constexpr int SPLIT = RK >= 16 ? (RK * GM_WG_SPLIT_NUM / 8) / 2 * 2 : RK / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll 4
            for (int kk = half ? SPLIT : 0; kk < (half ? RK : SPLIT); kk += 2) {
#pragma unroll
                for (int t = 0; t < NACC; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[kk * ld + aoff[t]], S[kk * ld + goff[t]], acc[t], 0, 0, 0);
            }
            if (half == 0 && more) store_stage_to(sm + (cur ^ 1) * BUF);    // other buffer: last read before the previous barrier
        }
        __syncthreads();
    }
    float* out = w.partial + (int64_t)chunk * (ldA + 1) * ldG;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        if (tile_id[t] >= 0) {
            const int tk = tile_id[t] / TN, tn = tile_id[t] % TN;
#pragma unroll
            for (int e = 0; e < 16; ++e) out[(int64_t)(tk * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * ldG + tn * 32 + li] = acc[t][e];
        }
    }
    if (zs == 0 && tid < ldG) out[(int64_t)ldA * ldG + tid] = bsum;
}

template <int TK, int TN>
static int launch_wgrad_fast(const WgradK& w, hipStream_t s) {
    constexpr int ld = (TK + TN) * 32;
    constexpr int RK = (16384 / ld) >= 32 ? 32 : (16384 / ld) >= 16 ? 16 : 8;
    constexpr int TPW = (TK * TN + WG_WAVES - 1) / WG_WAVES;
    int zs = 1;
    while (zs < TPW && w.n_chunks * zs < 192) zs <<= 1;       // TPW is 1, 2 or 4; chunk counts sit just under 256 by construction
    const size_t lds = 2 * RK * ld * sizeof(float);   // double-buffered stages
    GM_TRY(gm_func_full_lds((const void*)k_wgrad_fast<TK, TN, 1>));
    GM_TRY(gm_func_full_lds((const void*)k_wgrad_fast<TK, TN, (TPW >= 2 ? 2 : 1)>));
    GM_TRY(gm_func_full_lds((const void*)k_wgrad_fast<TK, TN, (TPW >= 4 ? 4 : 1)>));
    if (zs == 4 && TPW >= 4) hipLaunchKernelGGL((k_wgrad_fast<TK, TN, (TPW >= 4 ? 4 : 1)>), dim3(w.n_chunks, 4), dim3(WG_THREADS), lds, s, w);
    else if (zs >= 2 && TPW >= 2) hipLaunchKernelGGL((k_wgrad_fast<TK, TN, (TPW >= 2 ? 2 : 1)>), dim3(w.n_chunks, 2), dim3(WG_THREADS), lds, s, w);
    else hipLaunchKernelGGL((k_wgrad_fast<TK, TN, 1>), dim3(w.n_chunks, 1), dim3(WG_THREADS), lds, s, w);
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores (same exact 3-way operand split as gemm_split.h; K, N in {128, 256}).
// The MFMA k dimension is the ROW index, so an operand lane needs 8 consecutive rows of one column: the staged planes are
// [row octet][column][8 rows] bf16 -- a fragment read is a contiguous ds_read_b128 run (conflict-free) and a thread that loads
// one column of 8 rows (8 dword loads, lanes on consecutive columns: 256-B coalesced rows) writes one 16-byte unit per plane.
// Stage = 16 rows (one MFMA k step): 2 octets x (K + N) columns = 1024 or 768 (octet, column) items, two per thread.
//
// Round 3: EIGHT waves (512 threads) in a 4 x 2 grid instead of sixteen in 4 x 4.  One 16-row stage is ~1.5 us of MFMA work and a global
// round trip under load is 2-3 us; the 16-wave kernel had 128 registers per wave -- 64 of them accumulators -- and room for ONE stage of
// loader registers, so every stage waited for its rows (the kernel ran at the memory latency: 851 us on the 1.14 M-row batch for 430 us
// of MFMA work).  With 256 registers per wave the rows of TWO stages are in flight (inline-asm loads, hand-counted vmcnt: the compiler
// collapses a deeper register prefetch at the loop back-edge) and a wave owns 2 x 4 (K = 256: NTK x NTN) tiles, i.e. 18 instead of 36
// fragment reads per 48 MFMAs.  Accumulation order per output element is unchanged (stages in row order, the six products in the same
// order), so the partials are bit-identical to the 16-wave kernel's.  Row scales: rows are wave-uniform, so lane j & 7 fetches row j's
// scale with ONE load per item and stage, read back with v_readlane.  db = column sums of G from the loader registers.
// Output: the same per-chunk partial [(K+1) x N] as k_wgrad_fast (reduced by k_wgrad_reduce).
#define WGS_THREADS 512
__device__ __forceinline__ gm_f32x16 wgs_mfma(gm_bf16x8 a, gm_bf16x8 b, gm_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ gm_f32x16 wgs_mfma(gm_f16x8 a, gm_f16x8 b, gm_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ const float gm_wgs_ones[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};      // the row "scales" of an unscaled operand
// NP = 3: three bf16 pieces per operand, six products;  NP = 2: two fp16 pieces under the per-set power-of-two scales of w.a_bound / w.g_bound
// (gemm_split.h), three products, the partial leaves multiplied by 1 / (s_a s_g).
// GA = true (round 6): the A operand is never read as a matrix -- the pass's forward ran the fused aggregate + GEMM, so Z_l exists only for the rows
// of three or more sources; every other row is formed HERE from its one or two source rows (per-row table gm_batch::d_fuse2), in the aggregate
// kernel's own fma order: the same floats as the unfused pass, bit for bit, and the differentiated passes no longer write and re-read Z_l.
// The table rides in the loader's counted queue: the sources of stage s + 1 are fetched FIRST in the load group of stage s (the group's wait leaves
// one load less outstanding, so they have landed before the next group is issued), the weights with the rows they scale; rows are wave-uniform, so
// lane j & 7 fetches row j's entry, forms its two addresses (src_addr), and v_readlane broadcasts them into the SGPR base pairs of the group's loads.
// (Forming them inside the stage's MFMA sequence instead of in front of the loads measured the same; that variant's first build came out with phi
// copies of in-flight registers at its loop head -- tools/check_inflight_regs.py / tests/test_kernel_resources.py now replay every build's ISA.)
template <int KT, int NT, int NP = 3, bool GA = false>
__global__ __launch_bounds__(WGS_THREADS) void k_wgrad_split(WgradK w) {
    constexpr int K = KT * 128, N = NT * 128, COLS = K + N;
    constexpr int NTK = KT, NTN = 2 * NT;                                           // 32 x 32 tiles per wave: (K / 32) / 4 x (N / 32) / 2
    constexpr int A_PLANE = 2 * K * 16, G_PLANE = 2 * N * 16;                       // bytes: [2 octets][cols][8 rows] bf16
    constexpr int STAGE = NP * A_PLANE + NP * G_PLANE;                              // 32 NP (K + N) bytes: 48 KiB at 256 + 256, three pieces
    constexpr int ITEMS = 2 * COLS;                                                 // (octet, column) items per stage: 512 .. 1024
    static_assert(2 * K <= WGS_THREADS && 2 * N <= WGS_THREADS && K % 64 == 0 && N % 64 == 0, "one wave-uniform item per operand and thread");
    extern __shared__ __attribute__((aligned(16))) float sm_f[];
    char* sm = reinterpret_cast<char*>(sm_f);
    const int chunk = blockIdx.x;
    const int row0 = w.chunks[chunk * 3 + 1], nrows = w.chunks[chunk * 3 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int wy = wave >> 1, wx = wave & 1;
    gm_f32x16 acc[NTK][NTN];
#pragma unroll
    for (int a = 0; a < NTK; ++a)
#pragma unroll
        for (int b = 0; b < NTN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    // ---- the two loader items of this thread: item 0 = (octet, column) of A, item 1 = (octet, column) of G; the octet is wave-uniform
    // (K, N are multiples of 64).  K or N = 128: only the first 256 threads have an item of that operand; the others shadow item
    // (0, 0) -- they load (uniform vmcnt) but never store.
    auto uni64 = [](const void* q) -> uint64_t {
        const uint64_t v = (uint64_t)(uintptr_t)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((uint64_t)hi << 32) | lo;                                           // (unsigned: the builtin returns int, which would sign-extend)
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    bool valid[2]; int oct8[2], dst[2], gcol;
    unsigned ld_b[2], col_b[2]; uint64_t sbase[2];
    constexpr bool isA[2] = {true, false};
    constexpr int plane[2] = {A_PLANE, G_PLANE};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        constexpr int dummy = 0; (void)dummy;
        const int W = it == 0 ? K : N;                                              // columns of this operand
        valid[it] = wave_u * 64 < 2 * W;                                            // (wave-uniform)
        const int oct = valid[it] ? (wave_u * 64) / W : 0, c = valid[it] ? tid - oct * W : 0;
        oct8[it] = oct * 8;
        dst[it] = (it == 0 ? 0 : NP * A_PLANE) + oct * W * 16 + c * 16;
        if (it == 1) gcol = oct * N + c;
        const int64_t ld = it == 0 ? w.lda : w.ldg;
        // addresses: a wave-uniform 64-bit base (the operand at the chunk's first row / the scale vector) in SGPRs + ONE 32-bit byte
        // offset per load (row * ld + column; a chunk spans far less than 4 GiB)
        const float* base = (it == 0 ? w.A : w.G) + (int64_t)row0 * ld;
        sbase[it] = uni64(base);
        ld_b[it] = (unsigned)ld * 4u; col_b[it] = (unsigned)c * 4u;
    }
    // A's row scales (G has none): lane j & 7 of the wave fetches row j's scale; an unscaled A reads a vector of ones (branch-free)
    const bool scaled = w.a_scale != nullptr;
    const uint64_t sscale = uni64(scaled ? (const void*)(w.a_scale + row0) : (const void*)gm_wgs_ones);
    float pf[2][2][8], ps[2];                                                       // [slot = stage parity][item][row], [slot]
    float pf1[GA ? 2 : 1][8];                                                       // GA: the rows' second sources
    typedef int i2v __attribute__((ext_vector_type(2)));
    i2v tu = {0, 0}, tw[2] = {{0, 0}, {0, 0}};                                      // GA: {u0, u1} of the NEXT group's rows; {w0, w1} of the slot's rows (lane j & 7: row j)
    const uint64_t stab = GA ? uni64(w.f2 + row0) : 0, sgx = GA ? uni64(w.gx) : 0, szr = GA ? uni64(w.zrow) : 0, sza = GA ? uni64(w.A) : 0;
    const unsigned ldgx_b = GA ? (unsigned)w.ldgx * 4u : 0u;
    float bsum = 0.f;
    float op_scale[2] = {1.f, 1.f};                                                 // NP == 2: s_a, s_g of this chunk's set
    if constexpr (NP == 2) { const int set = w.chunks[chunk * 3]; op_scale[0] = gs_bound_scale(w.a_bound, set); op_scale[1] = gs_bound_scale(w.g_bound, set); }
    // Stage R0 / 16 -> slot SL: 17 loads per thread, ALWAYS issued (rows are clamped to the chunk, so a stage past the end re-reads the
    // last row and is never stored): every wait is the same vmcnt(17) and the loop body has no control flow around the asm
    // GA: byte address of source row u of a table entry: zrow / the row's own finished aggregate in A / a row of gx.  Formed LANE-PARALLEL -- lane j & 7 holds
    // row j's entry, so one pass of ~16 VALU instructions gives all 16 addresses of a load group, which v_readlane then broadcasts into SGPR pairs (the
    // loads take a wave-uniform 64-bit base + the lane's column offset).  The first version formed them one after the other by scalar arithmetic: 16
    // dependent chains of ~16 SALU instructions in front of every group's loads, and the weight gradients ran 40-60 % longer than on a stored Z.
    auto src_addr = [&](int u) -> uint64_t {
        const bool zero = (u & GM_FUSE_ZERO) != 0, self = (u & GM_FUSE_SELF) != 0;
        const unsigned idx = zero ? 0u : (unsigned)(u & ~(GM_FUSE_SELF | GM_FUSE_ZERO));
        const uint64_t base = zero ? szr : (self ? sza : sgx);
        return base + (uint64_t)idx * (uint64_t)(self ? ld_b[0] : ldgx_b);
    };
    uint64_t nb0[8] = {}, nb1[8] = {};
#define WGS_BASES()                                                                                                        \
    do {                                                                                                                   \
        const uint64_t a0_ = src_addr(tu.x), a1_ = src_addr(tu.y);                                                         \
        const int a0l = (int)(unsigned)a0_, a0h = (int)(unsigned)(a0_ >> 32), a1l = (int)(unsigned)a1_, a1h = (int)(unsigned)(a1_ >> 32); \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                    \
            nb0[j] = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(a0h, j) << 32) | (unsigned)__builtin_amdgcn_readlane(a0l, j); \
            nb1[j] = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(a1h, j) << 32) | (unsigned)__builtin_amdgcn_readlane(a1l, j); \
        }                                                                                                                  \
    } while (0)
    // table row of this lane for the stage at R0: lane j & 7 fetches row j of the wave's octet (clamped like the row loads)
#define WGS_TAB_OFF(R0) ((unsigned)min((R0) + oct8[0] + (lane & 7), nrows - 1) * 16u)
#define WGS_ISSUE(R0, SL)                                                                                                  \
    do {                                                                                                                   \
        if constexpr (GA) {                                                                                                \
            WGS_BASES();                                          /* the group's 16 row addresses, from the table sources the last group fetched */ \
            { const unsigned offn = WGS_TAB_OFF((R0) + 16);              /* FIRST load of the group: the NEXT group's sources */ \
              asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(tu) : "v"(offn), "s"(stab) : "memory"); }               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {          /* (zrow holds K zero floats: the column offset stays inside it) */ \
                /* s_nop 4: the base pair comes straight from v_readlane (a VALU write of an SGPR), and a VMEM instruction that reads such an SGPR needs five \
                   wait states the compiler does not insert in front of inline asm (the first lane-parallel version faulted on exactly that) */ \
                asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"                       \
                             : "=&v"(pf[SL][0][j]), "=&v"(pf1[SL][j]) : "v"(col_b[0]), "s"(nb0[j]), "s"(nb1[j]) : "memory");  \
            }                                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
                const unsigned off = (unsigned)min((R0) + oct8[1] + j, nrows - 1) * ld_b[1] + col_b[1];                    \
                asm volatile("global_load_dword %0, %1, %2" : "=v"(pf[SL][1][j]) : "v"(off), "s"(sbase[1]) : "memory");    \
            }                                                                                                              \
            { const unsigned offw = WGS_TAB_OFF(R0) + 8u;                                                                  \
              asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(tw[SL]) : "v"(offw), "s"(stab) : "memory"); }           \
        } else {                                                                                                           \
        _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                                 \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
                const unsigned off = (unsigned)min((R0) + oct8[it] + j, nrows - 1) * ld_b[it] + col_b[it];                 \
                asm volatile("global_load_dword %0, %1, %2" : "=v"(pf[SL][it][j]) : "v"(off), "s"(sbase[it]) : "memory"); \
            }                                                                                                              \
        }                                                                                                                  \
        }                                                                                                                  \
        const unsigned offs = scaled ? (unsigned)min((R0) + oct8[0] + (lane & 7), nrows - 1) * 4u : (unsigned)(lane & 7) * 4u; \
        asm volatile("global_load_dword %0, %1, %2" : "=v"(ps[SL]) : "v"(offs), "s"(sscale) : "memory");                  \
    } while (0)
#define WGS_TIE(SL, IT) "+v"(pf[SL][IT][0]), "+v"(pf[SL][IT][1]), "+v"(pf[SL][IT][2]), "+v"(pf[SL][IT][3]), "+v"(pf[SL][IT][4]), "+v"(pf[SL][IT][5]), "+v"(pf[SL][IT][6]), "+v"(pf[SL][IT][7])
#define WGS_TIE1(SL) "+v"(pf1[SL][0]), "+v"(pf1[SL][1]), "+v"(pf1[SL][2]), "+v"(pf1[SL][3]), "+v"(pf1[SL][4]), "+v"(pf1[SL][5]), "+v"(pf1[SL][6]), "+v"(pf1[SL][7])
    // slot SL has landed once at most the loads of the stage issued after it are outstanding: 17 -- or, GA, 27 less the group's first (the next
    // group's table sources, which the next WGS_ISSUE reads)
#define WGS_WAIT(SL)                                                                                                       \
    do {                                                                                                                   \
        if constexpr (GA) asm volatile("s_waitcnt vmcnt(26)" : WGS_TIE(SL, 0), WGS_TIE(SL, 1), WGS_TIE1(SL), "+v"(ps[SL]), "+v"(tw[SL]), "+v"(tu) :: "memory"); \
        else asm volatile("s_waitcnt vmcnt(17)" : WGS_TIE(SL, 0), WGS_TIE(SL, 1), "+v"(ps[SL]) :: "memory");               \
    } while (0)
#define WGS_DRAIN(SL)                                                                                                      \
    do {                                                                                                                   \
        if constexpr (GA) asm volatile("s_waitcnt vmcnt(0)" : WGS_TIE(SL, 0), WGS_TIE(SL, 1), WGS_TIE1(SL), "+v"(ps[SL]), "+v"(tw[SL]), "+v"(tu) :: "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" : WGS_TIE(SL, 0), WGS_TIE(SL, 1), "+v"(ps[SL]) :: "memory");                \
    } while (0)
    // split + store of ONE item, one plane at a time with the residual kept in place (x <- x - hi16(x), exact)
#define WGS_STORE_IT(R0, SL, S_, IT)                                                                                       \
    do {                                                                                                                   \
        constexpr int it = (IT);                                                                                           \
        if (valid[it]) {                                                                                                   \
            float x[8];                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
                const int r = (R0) + oct8[it] + j;                                                                         \
                float v_ = pf[SL][it][j];                                                                                  \
                if constexpr (GA) if (it == 0)                           /* the aggregate kernel's chain: fma(x1, w1, fma(x0, w0, 0)) */ \
                    v_ = __fmaf_rn(pf1[SL][j], __int_as_float(__builtin_amdgcn_readlane(tw[SL].y, j)),                     \
                                   __fmaf_rn(v_, __int_as_float(__builtin_amdgcn_readlane(tw[SL].x, j)), 0.f));            \
                if (it == 0) v_ *= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ps[SL]), j));                   \
                x[j] = r < nrows ? v_ : 0.f;                                        /* rows past the chunk end contribute zeros */ \
                if (it == 1) bsum += x[j];                                                                                 \
                if constexpr (NP == 2) x[j] *= op_scale[it];                                                               \
            }                                                                                                              \
            if constexpr (NP == 2) {                                                                                       \
                typedef _Float16 h2_ __attribute__((ext_vector_type(2)));                                                  \
                uint4 vh, vm; unsigned* ph = &vh.x; unsigned* pm = &vm.x;                                                  \
                _Pragma("unroll") for (int j = 0; j < 8; j += 2) {                                                         \
                    const _Float16 h0 = (_Float16)x[j], h1 = (_Float16)x[j + 1];                                           \
                    const h2_ hh = {h0, h1}, mm = {(_Float16)(x[j] - (float)h0), (_Float16)(x[j + 1] - (float)h1)};        \
                    ph[j >> 1] = __builtin_bit_cast(unsigned, hh); pm[j >> 1] = __builtin_bit_cast(unsigned, mm);          \
                }                                                                                                          \
                *reinterpret_cast<uint4*>((S_) + dst[it]) = vh;                                                            \
                *reinterpret_cast<uint4*>((S_) + dst[it] + plane[it]) = vm;                                                \
            } else                                                                                                         \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) {                                                          \
                uint4 v;                                                            /* hi16 of eight values, packed pairwise */ \
                v.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);                    \
                v.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);                    \
                v.z = __builtin_amdgcn_perm(__float_as_uint(x[5]), __float_as_uint(x[4]), 0x07060302u);                    \
                v.w = __builtin_amdgcn_perm(__float_as_uint(x[7]), __float_as_uint(x[6]), 0x07060302u);                    \
                *reinterpret_cast<uint4*>((S_) + dst[it] + pl_ * plane[it]) = v;                                           \
                if (pl_ < 2) { _Pragma("unroll") for (int j = 0; j < 8; ++j) x[j] = x[j] - __uint_as_float(__float_as_uint(x[j]) & 0xffff0000u); } \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#define WGS_STORE(R0, SL, S_) do { WGS_STORE_IT(R0, SL, S_, 0); WGS_STORE_IT(R0, SL, S_, 1); } while (0)
    const int a_lane = kh * K * 16 + (wy * NTK * 32 + li) * 16, g_lane = NP * A_PLANE + kh * N * 16 + (wx * NTN * 32 + li) * 16;
    // One stage: the MFMAs over LDS buffer SCUR with the split + store of the NEXT stage (register slot SLN -> LDS buffer SNEXT, last read
    // before the previous barrier) interleaved between the tile columns, so that the VALU work of the split runs under the matrix pipe
    // instead of after it (two waves per SIMD reach the end of their MFMAs together: as a separate phase the split left the pipe idle
    // for ~30 % of a stage).  The A fragments of the wave's tile rows stay resident, the G fragments stream by; per tile the order of the
    // six products is the 16-wave kernel's (l*h, h*l, m*m, m*h, h*m, h*h).  A stage past the chunk end is all zeros: its MFMAs add +0.
#define WGS_STAGE(SCUR, R0N, SLN, SNEXT)                                                                                   \
    do {                                                                                                                   \
        frag_t af[NTK][NP];                                                                                                \
        _Pragma("unroll") for (int a = 0; a < NTK; ++a)                                                                    \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) af[a][p] = *reinterpret_cast<const frag_t*>((SCUR) + a_lane + p * A_PLANE + a * 512); \
        WGS_WAIT(SLN);                                                                                                     \
        _Pragma("unroll") for (int b = 0; b < NTN; ++b) {                                                                  \
            frag_t gf[NP];                                                                                                 \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) gf[p] = *reinterpret_cast<const frag_t*>((SCUR) + g_lane + p * G_PLANE + b * 512); \
            if constexpr (NP == 3) { WGS_PROD(2, 0) WGS_PROD(0, 2) WGS_PROD(1, 1) WGS_PROD(1, 0) WGS_PROD(0, 1) WGS_PROD(0, 0) } \
            else { WGS_PROD(1, 0) WGS_PROD(0, 1) WGS_PROD(0, 0) }                                                          \
            if (b == 0) WGS_STORE_IT(R0N, SLN, SNEXT, 0);                                                                  \
            if (b == NTN / 2) WGS_STORE_IT(R0N, SLN, SNEXT, 1);                                                            \
        }                                                                                                                  \
    } while (0)
    typedef typename std::conditional<NP == 3, gm_bf16x8, gm_f16x8>::type frag_t;
#define WGS_PROD(PA, PB) _Pragma("unroll") for (int a = 0; a < NTK; ++a) acc[a][b] = wgs_mfma(af[a][PA], gf[PB], acc[a][b]);
    // Stage s lives in LDS buffer s & 1 and came through register slot s & 1.  Half-iteration of stage s: issue stage s + 2 (its slot was
    // emptied into LDS one half-iteration ago), then the stage.  Two half-iterations per loop trip so that the slots are compile-time
    // registers; no control flow around the asm (a copy the compiler inserted on one arm of a branch read a register before its wait).
    const int nst = (nrows + 15) >> 4;
    char* const S0 = sm; char* const S1 = sm + STAGE;
    if constexpr (GA) {                                 // sources of stage 0 (later ones ride in the groups)
        const unsigned off0 = WGS_TAB_OFF(0);
        asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(tu) : "v"(off0), "s"(stab) : "memory");
    }
    WGS_ISSUE(0, 0);
    if constexpr (GA) asm volatile("s_waitcnt vmcnt(26)" : "+v"(tu) :: "memory");      // stage 1's sources (the group's first load) before its rows are issued
    WGS_ISSUE(16, 1);
    WGS_WAIT(0);
    WGS_STORE(0, 0, S0);
    GS_BARRIER();
#pragma unroll 1
    for (int st = 0; st < nst; st += 2) {
        WGS_ISSUE((st + 2) * 16, 0);
        WGS_STAGE(S0, (st + 1) * 16, 1, S1);
        GS_BARRIER();
        WGS_ISSUE((st + 3) * 16, 1);
        WGS_STAGE(S1, (st + 2) * 16, 0, S0);
        GS_BARRIER();
    }
    WGS_DRAIN(1);                                       // the last (never used) issue: nothing may land in a register after this point
#undef WGS_ISSUE
#undef WGS_TAB_OFF
#undef WGS_BASES
#undef WGS_TIE
#undef WGS_TIE1
#undef WGS_WAIT
#undef WGS_DRAIN
#undef WGS_STORE
#undef WGS_STORE_IT
#undef WGS_STAGE
#undef WGS_PROD
    float* out = w.partial + (int64_t)chunk * (K + 1) * N;
    const float inv_a = 1.f / op_scale[0], inv_g = 1.f / op_scale[1];               // powers of two (1 when NP == 3)
#pragma unroll
    for (int a = 0; a < NTK; ++a)
#pragma unroll
        for (int b = 0; b < NTN; ++b) {
            const int tk = wy * NTK + a, tn = wx * NTN + b;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[a][b][e];
                if constexpr (NP == 2) v = v * inv_a * inv_g;
                out[(int64_t)(tk * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * N + tn * 32 + li] = v;
            }
        }
    // db: the two octet threads of a G column add up through LDS (fixed order)
    float* red = reinterpret_cast<float*>(sm);
    if (valid[1]) red[gcol] = bsum;
    __syncthreads();
    if (tid < N) out[(int64_t)K * N + tid] = red[tid] + red[N + tid];
}

template <int KT, int NT>
static int launch_wgrad_split(const WgradK& w, hipStream_t s, int np) {
    constexpr int K = KT * 128, N = NT * 128;
    const size_t lds = 2 * 32 * (size_t)np * (size_t)(K + N);
    if (w.f2) {                                          // A formed from the per-row source table (three-piece kernels only: launch_wgrad checks)
        GM_TRY(gm_func_full_lds((const void*)k_wgrad_split<KT, NT, 3, true>));
        hipLaunchKernelGGL((k_wgrad_split<KT, NT, 3, true>), dim3(w.n_chunks), dim3(WGS_THREADS), lds, s, w);
    } else if (np == 2) {
        GM_TRY(gm_func_full_lds((const void*)k_wgrad_split<KT, NT, 2>));
        hipLaunchKernelGGL((k_wgrad_split<KT, NT, 2>), dim3(w.n_chunks), dim3(WGS_THREADS), lds, s, w);
    } else {
        GM_TRY(gm_func_full_lds((const void*)k_wgrad_split<KT, NT, 3>));
        hipLaunchKernelGGL((k_wgrad_split<KT, NT, 3>), dim3(w.n_chunks), dim3(WGS_THREADS), lds, s, w);
    }
    return GM_OK;
}

// out_t[j] = sum over the set's chunks of partial[c][j];  j < K*N -> dW, else db.
struct WgradSgd {
    const float* cur; int64_t cur_stride; float* next; int64_t next_stride; float lr; int64_t w_off, b_off; float* wt; uint16_t* pl_fwd; uint16_t* pl_dz;
    int pl_np; gm_bound pl_bound;       // planes as two fp16 pieces under pl_bound (pl_np == 2) or three bf16 pieces
};

__device__ __forceinline__ void wgrad_reduce_body(const float* partial, const int32_t* set_chunk_off, int KN, int N, float* dW, int64_t dw_stride,
                                                  float* db, int64_t db_stride, const WgradSgd& u, const int bx, const int gx) {
    const int set = blockIdx.y;
    const int c0 = set_chunk_off[set], c1 = set_chunk_off[set + 1];
    const int tot = KN + N;
    for (int j = bx * blockDim.x + threadIdx.x; j < tot; j += gx * blockDim.x) {
        // 8 independent loads in flight; the summation order is fixed (deterministic), just not sequential
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) s8[u_] += partial[(int64_t)(c + u_) * tot + j];
        }
        for (int u_ = 0; c < c1; ++c, ++u_) s8[u_] += partial[(int64_t)c * tot + j];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        if (j < KN) {
            dW[(int64_t)set * dw_stride + j] = s;
            if (u.next) {
                const float wn = u.cur[(int64_t)set * u.cur_stride + u.w_off + j] - u.lr * s;
                u.next[(int64_t)set * u.next_stride + u.w_off + j] = wn;
                if (u.wt) { const int k = j / N, n = j - k * N; u.wt[(int64_t)set * KN + (int64_t)n * (KN / N) + k] = wn; }
                if (u.pl_fwd || u.pl_dz) {          // exact 3-way bf16 split of the new weight, straight into the next step's GEMM operand planes
                    const int K = KN / N, k = j / N, n = j - k * N;
                    uint32_t bh, bm, bl = 0;
                    if (u.pl_np == 2) {             // two fp16 pieces under the set's scale (bit patterns in the high halves, as below)
                        const float xs = wn * gs_bound_scale_v(u.pl_bound, set);
                        if (u.pl_bound.viol && fabsf(xs) > 65504.f && fabsf(wn) < INFINITY) atomicOr(u.pl_bound.viol, GM_VIOL_WEIGHT);
                        const _Float16 h = (_Float16)xs, m = (_Float16)(xs - (float)h);
                        bh = (uint32_t)__builtin_bit_cast(uint16_t, h) << 16; bm = (uint32_t)__builtin_bit_cast(uint16_t, m) << 16;
                    } else {
                        const uint32_t bx = __float_as_uint(wn); bh = bx & 0xffff0000u;
                        const float r1 = wn - __uint_as_float(bh);
                        bm = __float_as_uint(r1) & 0xffff0000u;
                        bl = __float_as_uint(r1 - __uint_as_float(bm));
                    }
                    if (u.pl_fwd) {                 // B[k][n] = W[k][n]  ->  [k/8][n][8]
                        uint16_t* o = u.pl_fwd + (int64_t)set * 3 * KN + ((int64_t)(k >> 3) * N + n) * 8 + (k & 7);
                        o[0] = (uint16_t)(bh >> 16); o[KN] = (uint16_t)(bm >> 16); if (u.pl_np != 2) o[2 * (int64_t)KN] = (uint16_t)(bl >> 16);
                    }
                    if (u.pl_dz) {                  // B[k'][n'] = W[n'][k'] (k' = n, n' = k)  ->  [n/8][k][8]
                        uint16_t* o = u.pl_dz + (int64_t)set * 3 * KN + ((int64_t)(n >> 3) * K + k) * 8 + (n & 7);
                        o[0] = (uint16_t)(bh >> 16); o[KN] = (uint16_t)(bm >> 16); if (u.pl_np != 2) o[2 * (int64_t)KN] = (uint16_t)(bl >> 16);
                    }
                }
            }
        } else if (db) {
            db[(int64_t)set * db_stride + (j - KN)] = s;
            if (u.next) u.next[(int64_t)set * u.next_stride + u.b_off + (j - KN)] = u.cur[(int64_t)set * u.cur_stride + u.b_off + (j - KN)] - u.lr * s;
        }
    }
}

// The same reduction for the fast-weight chains of gm_meta_step, where the updated weights also leave as split-bf16 operand planes
// (K % 8 == 0, N % 32 == 0): a block owns an 8 (k) x 32 (n) patch of W, so that a k-octet of one column (forward planes [k/8][n][8])
// and an n-octet of one row (dZ planes [n/8][k][8]) are each ONE 16-byte store instead of eight scattered 2-byte ones -- the
// element-per-thread version spent more time on those stores than on the partial sums.  Blocks past the patches reduce db.
__device__ __forceinline__ void wgrad_reduce_pl_body(const float* partial, const int32_t* set_chunk_off, int K, int N, float* dW, int64_t dw_stride,
                                                     float* db, int64_t db_stride, const WgradSgd& u, const int bx) {
    __shared__ uint16_t pl[3][8][40];                       // [plane][k in patch][n in patch], rows padded to 80 B
    const int set = blockIdx.y, tid = threadIdx.x;
    const int c0 = set_chunk_off[set], c1 = set_chunk_off[set + 1];
    const int KN = K * N, tot = KN + N, npn = N / 32, n_patch = (K / 8) * npn;
    auto sum_of = [&](int j) -> float {
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) s8[u_] += partial[(int64_t)(c + u_) * tot + j];
        }
        for (int u_ = 0; c < c1; ++c, ++u_) s8[u_] += partial[(int64_t)c * tot + j];
        return ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));      // same order as k_wgrad_reduce
    };
    if (bx >= n_patch) {                                    // db (and the bias step)
        const int n = (bx - n_patch) * 256 + tid;
        if (n < N && db) {
            const float s = sum_of(KN + n);
            db[(int64_t)set * db_stride + n] = s;
            if (u.next) u.next[(int64_t)set * u.next_stride + u.b_off + n] = u.cur[(int64_t)set * u.cur_stride + u.b_off + n] - u.lr * s;
        }
        return;
    }
    const int kb = bx / npn, nb = bx - kb * npn, tk = tid >> 5, tn = tid & 31;
    const int k = kb * 8 + tk, n = nb * 32 + tn, j = k * N + n;
    const float s = sum_of(j);
    dW[(int64_t)set * dw_stride + j] = s;
    if (!u.next) return;                                    // (uniform)
    const float wn = u.cur[(int64_t)set * u.cur_stride + u.w_off + j] - u.lr * s;
    u.next[(int64_t)set * u.next_stride + u.w_off + j] = wn;
    if (u.wt) u.wt[(int64_t)set * KN + (int64_t)n * K + k] = wn;
    if (!(u.pl_fwd || u.pl_dz)) return;                     // (uniform)
    const int np = u.pl_np == 2 ? 2 : 3;
    if (np == 2) {                                          // two fp16 pieces under the set's scale
        const float xs = wn * gs_bound_scale_v(u.pl_bound, set);
        if (u.pl_bound.viol && fabsf(xs) > 65504.f && fabsf(wn) < INFINITY) atomicOr(u.pl_bound.viol, GM_VIOL_WEIGHT);      // the fast weight outgrew the step's bound
        const _Float16 h = (_Float16)xs, m = (_Float16)(xs - (float)h);
        pl[0][tk][tn] = __builtin_bit_cast(uint16_t, h); pl[1][tk][tn] = __builtin_bit_cast(uint16_t, m);
    } else {
        const uint32_t bits = __float_as_uint(wn), bh = bits & 0xffff0000u;
        const float r1 = wn - __uint_as_float(bh);
        const uint32_t bm = __float_as_uint(r1) & 0xffff0000u;
        const uint32_t bl = __float_as_uint(r1 - __uint_as_float(bm));
        pl[0][tk][tn] = (uint16_t)(bh >> 16); pl[1][tk][tn] = (uint16_t)(bm >> 16); pl[2][tk][tn] = (uint16_t)(bl >> 16);
    }
    __syncthreads();
    if (tid < 32 * np && u.pl_fwd) {                             // forward planes: unit (plane, n) = the patch's 8 k of column n
        const int p = tid >> 5, c = tid & 31;
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pl[p][q][c];
        uint4 w4 = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
        *reinterpret_cast<uint4*>(u.pl_fwd + (int64_t)set * 3 * KN + (int64_t)p * KN + ((int64_t)kb * N + nb * 32 + c) * 8) = w4;
    } else if (tid >= 128 && tid < 128 + 32 * np && u.pl_dz) {        // dZ planes: unit (plane, k, n-octet) = 8 consecutive n of row k
        const int t = tid - 128, p = t >> 5, r = t & 31, kk = r >> 2, no = r & 3;
        uint16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pl[p][kk][no * 8 + q];
        uint4 w4 = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
        *reinterpret_cast<uint4*>(u.pl_dz + (int64_t)set * 3 * KN + (int64_t)p * KN + (((int64_t)(nb * 4 + no)) * K + kb * 8 + kk) * 8) = w4;
    }
}

// One reduction launch: its arguments, the variant (use_pl: the updated weights also leave as split-bf16 planes) and its grid width
struct RedK {
    const float* partial; const int32_t* set_chunk_off; int K, N; float* dW; int64_t dw_stride; float* db; int64_t db_stride; WgradSgd u;
    int use_pl, gx;
};
__device__ __forceinline__ void wgrad_reduce_one(const RedK& r, const int bx) {
    if (bx >= r.gx) return;
    if (r.use_pl) wgrad_reduce_pl_body(r.partial, r.set_chunk_off, r.K, r.N, r.dW, r.dw_stride, r.db, r.db_stride, r.u, bx);
    else wgrad_reduce_body(r.partial, r.set_chunk_off, r.K * r.N, r.N, r.dW, r.dw_stride, r.db, r.db_stride, r.u, bx, r.gx);
}
__global__ __launch_bounds__(256) void k_wgrad_reduce(RedK r) { wgrad_reduce_one(r, blockIdx.x); }
// The reductions of SEVERAL layers in one launch (blockIdx.z = layer): the backward of gm_meta_step holds the reductions of the layers above
// the first back (gm_wgrad_args::hold) and runs them with the first layer's -- nobody needs the updated W_l before the next forward.
struct RedN { RedK r[GM_MAX_GCN]; };
__global__ __launch_bounds__(256) void k_wgrad_reduce_n(RedN a) { wgrad_reduce_one(a.r[blockIdx.z], blockIdx.x); }

static int launch_wgrad(const gm_wgrad_args& a, hipStream_t s);
static bool wgrad_fast_ok(const gm_wgrad_args& a) {
    return (a.K % 32 == 0) && (a.N % 32 == 0) && !a.a_row && !a.Gb && (a.lda % 4 == 0) && (a.ldg % 4 == 0) &&
           (((uintptr_t)a.A & 15) == 0) && (((uintptr_t)a.G & 15) == 0);
}
// exact 3-way bf16 split of both operands, fp32 accumulation (k_wgrad_split): the same arithmetic as the split GEMM
static bool wgrad_split_shape_ok(int n_chunks, int K, int N) {
    return n_chunks > 0 && gm_knob().wgrad_split && gm_gemm_mode() == 1 && n_chunks >= (gm_knob().wgrad_split_min_chunks >= 0 ? gm_knob().wgrad_split_min_chunks : gm_num_cus() / 4) &&
           (K == 128 || K == 256) && (N == 128 || N == 256);
}
static bool wgrad_takes_split(const gm_wgrad_args& a) { return wgrad_fast_ok(a) && wgrad_split_shape_ok(a.n_chunks, a.K, a.N); }
// Would a weight gradient over `n_chunks` row chunks with these widths run on the split kernel -- the one that can form its A operand from the per-row
// source table (gm_wgrad_args::fuse2)?  The forward of a differentiated pass asks before it leaves Z_l unwritten (model.hip).
bool gm_wgrad_gather_ok(int n_chunks, int K, int N) { return wgrad_split_shape_ok(n_chunks, K, N); }
int gm_launch_wgrad(const gm_wgrad_args& a, hipStream_t s) {
    const int cat = wgrad_takes_split(a) ? ((a.np == 2 && a.a_bound.amax && a.g_bound.amax) ? GM_PROF_WGRAD_SPLIT16 : GM_PROF_WGRAD_SPLIT) : GM_PROF_WGRAD;
    gm_prof_begin(cat, s, 2 * a.rows * a.K * a.N);
    gm_prof_note(GM_PROF_WGRAD_BYTES, 4 * a.rows * ((int64_t)a.K + a.N));
    const int rc = launch_wgrad(a, s);
    gm_prof_end(cat, s);
    return rc;
}
static int launch_wgrad(const gm_wgrad_args& a, hipStream_t s) {
    const WgradSgd sgd{a.sgd_cur, a.sgd_cur_stride, a.sgd_next, a.sgd_next_stride, a.sgd_lr, a.w_off, a.b_off, a.sgd_next ? a.wt_next : nullptr,
                       a.sgd_next ? a.pl_fwd : nullptr, a.sgd_next ? a.pl_dz : nullptr, a.pl_np, a.pl_bound};
    GM_REQUIRE(!(a.pl_np == 2 && (a.pl_fwd || a.pl_dz)) || a.pl_bound.amax, GM_EINVAL, "wgrad: two-piece planes need a bound");
    // the reduction of this call's partials: launched now, or held back and launched together with a later call's (gm_wgrad_hold)
    auto reduce = [&](bool want_pl) -> int {
        RedK r{a.partial, a.set_chunk_off, a.K, a.N, a.dW, a.dw_stride, a.db, a.db_stride, sgd, 0, 0};
        r.use_pl = want_pl && (sgd.pl_fwd || sgd.pl_dz) && a.K % 8 == 0 && a.N % 32 == 0;
        r.gx = r.use_pl ? (a.K / 8) * (a.N / 32) + (a.N + 255) / 256 : ((a.K + 1) * a.N + 255) / 256;
        gm_wgrad_hold* h = a.hold;
        if (h && a.hold_this && h->n < GM_MAX_GCN - 1) {            // keep it for the flush
            static_assert(sizeof(RedK) <= sizeof(h->slot[0]), "gm_wgrad_hold slot too small");
            memcpy(h->slot[h->n], &r, sizeof(RedK)); h->sets[h->n] = a.sets; ++h->n;
            return GM_OK;
        }
        if (h && h->n > 0) {                                        // flush: the held reductions + this one, one launch
            RedN all{}; int gx = r.gx, n = 0;
            for (; n < h->n; ++n) { memcpy(&all.r[n], h->slot[n], sizeof(RedK)); gx = std::max(gx, all.r[n].gx); GM_REQUIRE(h->sets[n] == a.sets, GM_EINVAL, "wgrad: held reductions of different batches"); }
            all.r[n++] = r; h->n = 0;
            hipLaunchKernelGGL(k_wgrad_reduce_n, dim3(gx, a.sets, n), dim3(256), 0, s, all);
        } else {
            hipLaunchKernelGGL(k_wgrad_reduce, dim3(r.gx, a.sets), dim3(256), 0, s, r);
        }
        GM_HIP(hipGetLastError());
        return GM_OK;
    };
    if (a.n_chunks <= 0) return reduce(false);      // no rows at all: the gradients are zero
    WgradK w{};
    w.A = a.A; w.lda = a.lda; w.K = a.K; w.a_row = a.a_row; w.G = a.G; w.ldg = a.ldg; w.N = a.N; w.Gb = a.Gb; w.ldgb = a.ldgb;
    w.a_scale = a.a_scale; w.chunks = a.chunks; w.n_chunks = a.n_chunks; w.partial = a.partial;
    w.TK = (a.K + 31) / 32; w.TN = (a.N + 31) / 32;
    bool launched = false;
    const bool fast_ok = wgrad_fast_ok(a);
    GM_REQUIRE(!a.fuse2 || (wgrad_takes_split(a) && a.gx && a.np != 2 && a.lda <= (int64_t)(1 << 20) && a.ldgx <= (int64_t)(1 << 20)), GM_EINVAL,
               "wgrad: a table-formed A operand needs the split kernel (K=%d N=%d chunks=%d)", a.K, a.N, a.n_chunks);
    if (wgrad_takes_split(a)) {
        const int np = (a.np == 2 && a.a_bound.amax && a.g_bound.amax) ? 2 : 3;
        if (a.fuse2) { w.f2 = (const int4*)a.fuse2; w.gx = a.gx; w.ldgx = a.ldgx; w.zrow = gm_zero_row(s); GM_REQUIRE(w.zrow, GM_ENOMEM, "wgrad: no zero row"); }
        w.a_bound = a.a_bound; w.g_bound = a.g_bound;
        if (a.K == 256 && a.N == 256) GM_TRY((launch_wgrad_split<2, 2>(w, s, np)));
        else if (a.K == 128 && a.N == 256) GM_TRY((launch_wgrad_split<1, 2>(w, s, np)));
        else if (a.K == 256 && a.N == 128) GM_TRY((launch_wgrad_split<2, 1>(w, s, np)));
        else GM_TRY((launch_wgrad_split<1, 1>(w, s, np)));
        launched = true;
    }
    if (fast_ok && !launched) {
#define GM_WG_CASE(TK_, TN_) if (!launched && w.TK == TK_ && w.TN == TN_) { GM_TRY((launch_wgrad_fast<TK_, TN_>(w, s))); launched = true; }
        GM_WG_CASE(8, 8) GM_WG_CASE(4, 8) GM_WG_CASE(8, 4) GM_WG_CASE(4, 4) GM_WG_CASE(2, 4) GM_WG_CASE(4, 2) GM_WG_CASE(2, 2)
        GM_WG_CASE(1, 2) GM_WG_CASE(2, 1) GM_WG_CASE(1, 4) GM_WG_CASE(1, 8) GM_WG_CASE(1, 1)
#undef GM_WG_CASE
    }
    if (launched) {
        GM_HIP(hipGetLastError());
        return reduce(true);
    }
    const int ld = (w.TK + w.TN) * 32;
    w.RK = 32;
    while (w.RK > 2 && w.RK * ld > WG_PF * WG_THREADS * 4) w.RK >>= 1;
    GM_REQUIRE(w.RK * ld <= WG_PF * WG_THREADS * 4, GM_ERANGE, "wgrad: K=%d N=%d too wide", a.K, a.N);
    w.vec = (a.K % 4 == 0) && (a.N % 4 == 0) && (a.lda % 4 == 0) && (a.ldg % 4 == 0) && (((uintptr_t)a.A & 15) == 0) && (((uintptr_t)a.G & 15) == 0);
    const size_t lds = (size_t)w.RK * ld * sizeof(float);
    GM_REQUIRE(lds <= 160 * 1024, GM_ERANGE, "wgrad: K=%d N=%d needs %zu B of LDS", a.K, a.N, lds);
    GM_TRY(gm_func_full_lds((const void*)k_wgrad));
    const int zgroups = (w.TK * w.TN + WG_WAVES * WG_MAXT - 1) / (WG_WAVES * WG_MAXT);
    hipLaunchKernelGGL(k_wgrad, dim3(a.n_chunks, zgroups), dim3(WG_THREADS), lds, s, w);
    GM_HIP(hipGetLastError());
    return reduce(false);
}
