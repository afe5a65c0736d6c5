// fp32 GEMM on the bf16 matrix cores by exact operand splitting (gfx950).
//
// The update GEMMs of GraphConv (learner.py:36,47) are fp32 in the reference.  The f32-input MFMA
// (v_mfma_f32_32x32x2_f32) runs at the vector rate, 1/16 of the bf16 rate, so the step was bound by it.  Here every
// fp32 operand x is split EXACTLY into three bf16 pieces by truncation,
//     x = x_h + x_m + x_l,   x_h = top 8 significand bits, x_m = next 8, x_l = last 8   (8+8+8 = 24 bits of an fp32),
// and a*b is evaluated as the six products whose weight is >= 2^-16,
//     a*b ~= a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m,
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped terms are
// bounded by (2^-24 + 2^-24 + 2^-32) |a||b| ~= 1.2e-7 |a||b| per product -- the size of ONE fp32 rounding -- so the
// result is an fp32-accurate product (measured against fp64: same error as the fmaf chain, see tools/gemm_split_bench.hip),
// at 6/16 of the f32-MFMA cost.  Splitting costs ~5.5 VALU ops per element and is done ONCE per block tile when the A
// rows are staged into LDS; the (small, per-task) weights are split once per launch by k_split_w into three bf16 planes
// stored [n][k] (k contiguous), so the B tiles are DMA-ed straight into LDS in MFMA fragment order.
//
// Second arithmetic (template parameter NP = 2, what gm_meta_step runs where magnitude bounds are recorded): two fp16
// pieces per operand under per-set power-of-two scales (gm_bound.h, gs_scale_of) and THREE products a_h b_h + a_h b_m + a_m b_h on
// v_mfma_f32_32x32x16_f16 -- 22 significand bits per operand, measured error vs fp64 below the three-piece kernel's; half the matrix work,
// two thirds of the LDS / weight-plane traffic.  Same kernel body: plane counts, the split in the feeders and the MFMA type are the only
// differences, plus the epilogue's running max |C| that feeds the next consumer's bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gm_bound.h"

typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gm_f16x8 __attribute__((ext_vector_type(8)));
typedef float gm_f32x16 __attribute__((ext_vector_type(16)));

#define GS_BM 128
// hipcc's __syncthreads() is fence + barrier: it drains vmcnt(0), i.e. every prefetch in flight.  The main loops use the raw
// barrier behind an explicit LDS wait (and counted vmcnt where a DMA must have landed).
#define GS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

// Tile descriptor {set, row0, nrows} through the SCALAR cache: a vector load here would sit in the wave's in-order vmcnt queue, and the
// wait for it would drain every prefetched operand load (and every store of the previous tile) with it.
struct GsTile { int set, row0, nrows; };
__device__ __forceinline__ GsTile gs_tile(const int32_t* tiles, int lt) {
    const uint64_t p = (uint64_t)(uintptr_t)(tiles + (int64_t)lt * 3);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const uint64_t sp = ((uint64_t)hi << 32) | lo;
    int a, b, c;
    asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %3, 0x4\n\ts_load_dword %2, %3, 0x8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(sp) : "memory");
    return GsTile{a, b, c};
}

// one dword through the scalar cache (same reason)
__device__ __forceinline__ unsigned gs_sload(const void* q) {
    const uint64_t p = (uint64_t)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const uint64_t sp = ((uint64_t)hi << 32) | lo;
    unsigned a;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(sp) : "memory");
    return a;
}

struct SplitGemmK {
    const float* A; int64_t lda;
    const uint16_t* Bt; int64_t bt_stride;     // split weights [set][3][K/8][N][8] bf16 (k_split_w), per-set stride in ELEMENTS (0 = shared)
    float* C; int64_t ldc; int K, N;
    const float* row_scale; const float* bias; int64_t bias_stride; const float* mask_h; int relu;
    const uint8_t* mask_b; uint8_t* relu_bits;
    const int32_t* tiles; int n_tiles; int n_col_tiles; int nt_store;
    unsigned long long* dbg;                   // experiment: per-iteration timestamps of one workgroup (FC_TRACE)
    int dephase;                               // k_gemm_split_h: s_sleep(127) count for the second-slot workgroups
    // k_gemm_split_p<true> (fused aggregate + GEMM): A / lda address the aggregate's INPUT rows, f2 is the per-row source table, rows
    // flagged GM_SPLIT_FUSE_SELF read their finished aggregate from zside (row stride ldz), rows without a source read zeros
    float* zero_out;                           // optional [rows, ldc]: the epilogue also zero-fills this buffer's tile (dQ of the backward pass that follows)
    const int4* f2; const float* zside; int64_t ldz; const float* zrow;     // zrow: >= K zero floats (rows flagged GM_SPLIT_FUSE_ZERO)
    // two-piece fp16 operands (NP == 2): per-set magnitude bounds of the A rows and of the weights (gm_bound.h), from which both sides
    // derive the same power-of-two scales (gs_scale_of);  amax_out[set * GM_BOUND_PAD] (NP == 2 kernels only): the epilogue records the largest |C| it stores
    // (atomicMax on the bit pattern) -- the bound of whoever consumes C
    gm_bound a_bound, b_bound; unsigned* amax_out;
    int keep_signed;                           // row_scale carries "nobody reads this row" in its sign bit: such rows are computed and not stored
};
#define GM_SPLIT_FUSE_SELF 0x40000000
#define GM_SPLIT_FUSE_ZERO 0x20000000

// x (4 floats) -> three packed bf16x4 (8 bytes each): h = trunc16(x), m = trunc16(x - h), l = x - h - m (exact in bf16)
__device__ __forceinline__ void gs_split4(const float4 v, uint2& h, uint2& m, uint2& l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint32_t xh[4], xm[4], xl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = __float_as_uint(x[i]);
        xh[i] = b & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(xh[i]);            // exact
        const uint32_t b1 = __float_as_uint(r1);
        xm[i] = b1 & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(xm[i]);              // exact, <= 8 significant bits
        xl[i] = __float_as_uint(r2);
    }
    // pack pairs: (hi16 of element 1) << 16 | hi16 of element 0
    h.x = __builtin_amdgcn_perm(xh[1], xh[0], 0x07060302u); h.y = __builtin_amdgcn_perm(xh[3], xh[2], 0x07060302u);
    m.x = __builtin_amdgcn_perm(xm[1], xm[0], 0x07060302u); m.y = __builtin_amdgcn_perm(xm[3], xm[2], 0x07060302u);
    l.x = __builtin_amdgcn_perm(xl[1], xl[0], 0x07060302u); l.y = __builtin_amdgcn_perm(xl[3], xl[2], 0x07060302u);
}

// Two-piece fp16 split (NP == 2): with s a power of two such that |x| s <= 2^15,  x s = h + m + r,  h = fp16(x s) (11 significant bits,
// round to nearest), m = fp16(x s - h) (the next 11 bits; the subtraction is exact), |r| <= max(2^-23 |x s|, 2^-25): fp16 keeps 2^-24
// absolute (subnormals), i.e. 2^-39 of the bound.  a b ~= (a_h b_h + a_h b_m + a_m b_h) / (s_a s_b): the dropped terms are bounded by
// ~2^-21.4 |a||b| -- three fp32 roundings' worth, against the K roundings of an fp32 dot product.
__device__ __forceinline__ float gs_scale_of(float bound) {
    // largest power of two s with bound * s <= 2^15, clamped to [2^-40, 2^40]; bound = 0 (or not finite) -> 1
    const unsigned e = (__float_as_uint(bound) >> 23) & 0xffu;               // bound < 2^(e - 126)
    if (e == 0u || e == 0xffu) return 1.f;
    int se = 15 - ((int)e - 126);                                            // bound * 2^se < 2^15
    se = se < -40 ? -40 : (se > 40 ? 40 : se);
    return __uint_as_float((unsigned)(se + 127) << 23);
}
// the scale of a set under a recorded bound, through the scalar cache (wave-uniform); no bound -> 1
__device__ __forceinline__ float gs_bound_scale(const gm_bound& b, int set) {
    if (!b.amax) return 1.f;
    const float gain = b.gain ? __uint_as_float(gs_sload(b.gain)) : 1.f;
    return gs_scale_of(__uint_as_float(gs_sload(b.amax + (int64_t)set * b.stride)) * gain * b.hgain);
}
// the same from an ordinary (vector) load, for kernels that do not hand-count their vmcnt
__device__ __forceinline__ float gs_bound_scale_v(const gm_bound& b, int set) {
    if (!b.amax) return 1.f;
    const float gain = b.gain ? b.gain[0] : 1.f;
    return gs_scale_of(__uint_as_float(b.amax[(int64_t)set * b.stride]) * gain * b.hgain);
}
// slot = max(slot, bits) for bit patterns of non-negative floats.  Device-scope atomics on one address are executed one at a time at the
// memory side (~0.2 us each on gfx950: a thousand of them were 0.25 ms of a reduction launch), so look first: the slot only grows, and a
// stale look costs one redundant atomic.
__device__ __forceinline__ void gs_note_max(unsigned* slot, unsigned bits) {
    if (bits == 0u) return;
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(slot, bits);
}
__device__ __forceinline__ void gs_split4_f16(const float4 v, const float s, uint2& h, uint2& m) {
    const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    _Float16 xh[4], xm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xh[i] = (_Float16)x[i]; xm[i] = (_Float16)(x[i] - (float)xh[i]); }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h0 = {xh[0], xh[1]}, h1 = {xh[2], xh[3]}, m0 = {xm[0], xm[1]}, m1 = {xm[2], xm[3]};
    h.x = __builtin_bit_cast(unsigned, h0); h.y = __builtin_bit_cast(unsigned, h1);
    m.x = __builtin_bit_cast(unsigned, m0); m.y = __builtin_bit_cast(unsigned, m1);
}

// Weights -> split planes.  W_t = params + t*pstride + w_off.  trans = 0: the GEMM's B is W itself, W stored [K][N]
// (forward, B[k][n] = W[k][n]); trans = 1: B = W^T with W stored [N][K] (dZ = dQ W^T), i.e. Bt[n][k] = W[n][k] as stored.
// Output Bt[t][p][k/8][n][8] (bf16 bits; K % 8 == 0), p = 0 h, 1 m, 2 l: the 8 k of one MFMA operand lane are contiguous and
// 64 consecutive n of one k-octet form a contiguous 1-KiB DMA piece.  grid (ceil(K/32), ceil(N/32), sets), block 256.
// np = 2: two fp16 planes of W * s, s = the scale under `bound` (the bound recorded for this set's weights); the per-set stride stays 3 planes.
__global__ __launch_bounds__(256) void k_split_w(const float* params, int64_t pstride, int64_t w_off, int K, int N, int trans, uint16_t* Bt, int np, gm_bound bound) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const float* W = params + (int64_t)t * pstride + w_off;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // tile[kk][nn] = B[k0+kk][n0+nn]
    for (int r = ty; r < 32; r += 8) {
        float v = 0.f;
        if (!trans) { const int k = k0 + r, n = n0 + tx; if (k < K && n < N) v = W[(int64_t)k * N + n]; tile[r][tx] = v; }
        else { const int n = n0 + r, k = k0 + tx; if (k < K && n < N) v = W[(int64_t)n * K + k]; tile[tx][r] = v; }
    }
    __syncthreads();
    uint16_t* O = Bt + (int64_t)t * 3 * N * K;
    const float sb = np == 2 ? gs_bound_scale_v(bound, t) : 1.f;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n >= N || k >= K) continue;
        const float x = tile[tx][r];
        if (np == 2) {
            const int64_t plane = (int64_t)N * K, at = ((int64_t)(k >> 3) * N + n) * 8 + (k & 7);
            const float xs = x * sb;
            if (bound.viol && fabsf(xs) > 65504.f && fabsf(x) < INFINITY) atomicOr(bound.viol, GM_VIOL_WEIGHT);      // (a finite weight whose pieces are not)
            const _Float16 h = (_Float16)xs, m = (_Float16)(xs - (float)h);
            O[at] = __builtin_bit_cast(uint16_t, h);
            O[plane + at] = __builtin_bit_cast(uint16_t, m);
            continue;
        }
        const uint32_t b = __float_as_uint(x), bh = b & 0xffff0000u;
        const float r1 = x - __uint_as_float(bh);
        const uint32_t bm = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(bm);
        const int64_t plane = (int64_t)N * K, at = ((int64_t)(k >> 3) * N + n) * 8 + (k & 7);      // [k/8][n][8]
        O[0 * plane + at] = (uint16_t)(bh >> 16);
        O[1 * plane + at] = (uint16_t)(bm >> 16);
        O[2 * plane + at] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
}


// out[t * out_stride] = max(out[..], max |x[t * stride + off + i]|, i < n) as fp32 bit patterns (zero the slots first).  grid (blocks, sets).
__global__ __launch_bounds__(256) void k_amax(const float* x, int64_t stride, int64_t off, int64_t n, unsigned* out, int64_t out_stride) {
    const float* p = x + (int64_t)blockIdx.y * stride + off;
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) gs_note_max(out + (int64_t)blockIdx.y * out_stride, __float_as_uint(m));
}

// the same for up to 8 segments of one vector in ONE launch (blockIdx.y = segment): out[y * out_stride] <- max |x[off[y] + i]|, i < n[y]
struct AmaxSegs { int64_t off[8]; int64_t n[8]; };
__global__ __launch_bounds__(256) void k_amax_segs(const float* x, AmaxSegs sg, unsigned* out, int64_t out_stride) {
    const float* p = x + sg.off[blockIdx.y]; const int64_t n = sg.n[blockIdx.y];
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) gs_note_max(out + (int64_t)blockIdx.y * out_stride, __float_as_uint(m));
}

// 4 x 4 transpose inside every quad of lanes, registers <-> lanes (DPP quad permutes, no LDS): in: register k of lane t = M[k][t];
// out: register k of lane t = M[t][k].  The 32 x 32 MFMA accumulator holds, in registers 4g .. 4g+3 of lane li, rows 8g + 4kh + (0..3) of
// COLUMN li; after the transpose lane 4q + t holds row 8g + 4kh + t, columns 4q .. 4q+3 -- a 16-byte store per lane, 128 contiguous bytes
// per row and store instruction.
template <int CTRL> __device__ __forceinline__ float gs_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void gs_quad_transpose(float& a0, float& a1, float& a2, float& a3, const bool odd1, const bool odd2) {
    // stage 1: bit 0 of the register index <-> bit 0 of the lane (quad_perm [1,0,3,2] = 0xB1)
    const float p1 = gs_dpp<0xB1>(a1), q0 = gs_dpp<0xB1>(a0), p3 = gs_dpp<0xB1>(a3), q2 = gs_dpp<0xB1>(a2);
    const float b0 = odd1 ? p1 : a0, b1 = odd1 ? a1 : q0, b2 = odd1 ? p3 : a2, b3 = odd1 ? a3 : q2;
    // stage 2: bit 1 (quad_perm [2,3,0,1] = 0x4E)
    const float r2 = gs_dpp<0x4E>(b2), s0 = gs_dpp<0x4E>(b0), r3 = gs_dpp<0x4E>(b3), s1 = gs_dpp<0x4E>(b1);
    a0 = odd2 ? r2 : b0; a2 = odd2 ? b2 : s0;
    a1 = odd2 ? r3 : b1; a3 = odd2 ? b3 : s1;
}

// ------------------------------------------------------------------------------------------------------------------
// PERSISTENT, wave-specialised split-bf16 GEMM for N == 256 or 128 (one 128 x N tile per step, 16-k chunks):
//   waves 0..7    COMPUTE  64x64 sub-tiles: LDS fragments -> 24 bf16 MFMAs per chunk; at the end of a tile the accumulators go
//                 through a wave-PRIVATE LDS staging area to row-contiguous 16-B global stores.  These waves never issue a
//                 global LOAD, so they never wait on vmcnt: the stores of tile i drain while tile i+1 is being computed.
//   waves 8,9     A FEEDERS  HBM -> registers PF_DA chunks ahead (inline-asm loads, hand-counted vmcnt) -> exact split -> LDS;
//                 they also stage the tile's row scales and bias (so that the compute waves need no global load).
//   waves 12..15  B FEEDERS  LDS-DMA of the three bf16 weight planes, TWO chunks ahead (6 one-KiB pieces per wave and chunk, three stages).
// Every wave runs the same flattened (tile, chunk) sequence with ONE raw barrier per chunk; feeders run one chunk ahead across
// tile boundaries.  One workgroup per CU, grid = min(tiles, CUs); workgroup b walks tiles b, b + G, ... of the XCD-contiguous
// order.  LDS: 2 stages x 36 KiB + 8 x 8.5 KiB staging + scales/bias = 143 KiB.
#ifndef PF_DA
#define PF_DA 4
#endif
//   NP = 3: three bf16 pieces per operand, six products (exact);  NP = 2: two fp16 pieces under per-set power-of-two scales, three products
template <bool GATHER, int MI, int WC = 4, int NP = 3>
__global__ __launch_bounds__(1024) void k_gemm_split_p(SplitGemmK g) {
    // The eight compute waves form a (8 / WC) x WC grid of (32 MI) x 64 blocks: N = 64 WC columns, tile height BM = 32 MI (8 / WC).
    //   <., 2, 4>  N = 256, 128-row tiles (each compute wave a 64 x 64 block)
    //   <., 1, 4>  N = 256, 64-row HALF tiles (32 x 64 per wave) for launches that would leave more than half of the CUs without a tile --
    //              a small batch is bound by one tile's MFMA chain, not by throughput
    //   <., 1, 2>  N = 128 (hidden_dim 128: the Tissue-PPI / FirstMM-DB configurations), 128-row tiles, 32 x 64 per wave
    constexpr int BK = 16, BN = 64 * WC, BM = 32 * MI * (8 / WC), SPLIT = GS_BM / BM;
    static_assert((WC == 4 || WC == 2) && (BM == 128 || BM == 64), "wave grid");
    constexpr int A_OCT = BM * 16, B_OCT = BN * 16, A_PLANE = 2 * A_OCT, B_PLANE = 2 * B_OCT;
    // LDS: two A stages (12 KiB each), NB = 3 B stages (24 KiB each: the weight planes are DMA-ed TWO chunks ahead -- in the meta-step the
    // part runs this kernel at ~2 GHz, where one chunk of MFMAs (1536 cycles) no longer covers the issue cost of six DMA pieces plus an L2
    // round trip, and every chunk's barrier waited for the B feeders), epilogue staging of 16 rows x 64 columns per compute wave
    constexpr int A_ST = NP * A_PLANE, B_ST = NP * B_PLANE, NB = 3;
    static_assert(NP == 2 || NP == 3, "pieces per operand");
    constexpr int OFF_B = 2 * A_ST;
    constexpr int E_WAVE = 0;                                                         // (no epilogue staging: the accumulators are transposed in registers)
    constexpr int OFF_E = OFF_B + NB * B_ST, OFF_SC = OFF_E + 8 * E_WAVE, OFF_BIAS = OFF_SC + 2 * GS_BM * 4;
    constexpr int A_PER = (BM * BK / 4) / 256;                                        // float4 per A-feeder lane and chunk: MI (four feeder waves)
    constexpr int B_PPW = (NP * 2 * (BN / 64)) / 4;                                   // DMA pieces per B-feeder wave and chunk: 6 (N = 256) / 3
    constexpr int OFF_MAX = OFF_BIAS + 2 * BN * 4;                                    // {max bits, arrivals} of the compute waves' amax_out flush
    __shared__ __attribute__((aligned(16))) char smem[OFF_MAX + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = g.K / BK;
    const int G = gridDim.x, b = blockIdx.x;
    // XCD-contiguous tile order over this launch: hardware id t -> logical tile
    const int nb = g.n_tiles * SPLIT, q8 = nb / 8, r8 = nb % 8;
    auto logical = [&](int t) -> int { const int x = t % 8, i = t / 8; return (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; };
    const int ntb = b < nb ? (nb - b + G - 1) / G : 0;                                // tiles of this workgroup
    auto tile_of = [&](int w) -> GsTile {                                             // hardware work id -> {set, row0, nrows} of its (half) tile
        const int lt = logical(w);
        if constexpr (SPLIT == 1) return gs_tile(g.tiles, lt);
        else { GsTile t = gs_tile(g.tiles, lt >> 1); const int h = (lt & 1) * 64; t.row0 += h; t.nrows = max(0, min(64, t.nrows - h)); return t; }
    };
    // NP == 2: the power-of-two operand scales of a set, from the recorded bounds (scalar loads)
    auto a_scale_of = [&](int set) -> float {
        if constexpr (NP == 3) return 1.f;
        else return gs_bound_scale(g.a_bound, set);
    };
    auto b_scale_of = [&](int set) -> float {
        if constexpr (NP == 3) return 1.f;
        else return gs_bound_scale(g.b_bound, set);
    };
    const int total = ntb * nchunks;                                                  // flattened chunk count
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    float* scales = reinterpret_cast<float*>(smem + OFF_SC);                          // [2][128] by tile parity
    float* biasl = reinterpret_cast<float*>(smem + OFF_BIAS);                         // [2][256]
    if (total == 0) return;
    if (NP == 2 && tid < 2) reinterpret_cast<unsigned*>(smem + OFF_MAX)[tid] = 0u;     // (visible to the compute waves after their first barrier)
    if (wave >= 12) {
        // ================= B feeder
        __builtin_amdgcn_s_setprio(2);            // feeders ahead of the MFMA-heavy compute waves in VALU / LDS arbitration
        const int fw = wave - 12;
        unsigned boff[B_PPW]; int bdst[B_PPW];
#pragma unroll
        for (int p = 0; p < B_PPW; ++p) {
            const int piece = fw * B_PPW + p, cb = piece % (BN / 64), po = piece / (BN / 64), oct = po % 2, plane = po / 2;
            boff[p] = (unsigned)(((int64_t)plane * g.N * g.K + ((int64_t)oct * g.N + cb * 64 + lane) * 8) * 2);
            bdst[p] = OFF_B + plane * B_PLANE + oct * B_OCT + cb * 1024;
        }
        const int64_t b_chunk_bytes = (int64_t)BK * g.N * 2;
        // chunks are issued in order: (tile, chunk) counters instead of a division per chunk, the tile's set looked up once per tile
        int ib_c = 0, ib_ti = 0;
        uint64_t ib_base = (uint64_t)(uintptr_t)(g.Bt + (int64_t)tile_of(b).set * g.bt_stride);
        int ib_st = 0;                                                                // B stage of the next chunk to issue (gc % NB)
        auto issue_b = [&](int gc) {                                                  // global chunk gc (= the next in order) -> B stage gc % NB
            if (ib_c == nchunks) { ib_c = 0; ++ib_ti; ib_base = (uint64_t)(uintptr_t)(g.Bt + (int64_t)tile_of(b + ib_ti * G).set * g.bt_stride); }
            const uint64_t base = ib_base + (uint64_t)(ib_c * b_chunk_bytes);
            ++ib_c;
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
            const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ib_st * B_ST + bdst[p]));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(boff[p]), "s"(sbase), "s"(dst) : "memory");
            }
            ib_st = ib_st + 1 == NB ? 0 : ib_st + 1;
            (void)gc;
        };
        issue_b(0);
        if (total > 1) { issue_b(1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(B_PPW) : "memory"); }      // chunk 0 has landed, chunk 1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GS_BARRIER();
        for (int gc = 0; gc < total; ++gc) {
            // chunk gc + 2 -> the stage chunk gc - 1 was read from (all waves are past that chunk's barrier); chunk gc + 1 must have landed
            // before the barrier that ends chunk gc
            if (gc + 2 < total) { issue_b(gc + 2); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(B_PPW) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GS_BARRIER();
        }
    } else if (wave >= 8) {
        // ================= A feeder
        __builtin_amdgcn_s_setprio(3);
        const int ft = tid - 512;                                                     // 0..255
        int rr[A_PER];
#pragma unroll
        for (int p = 0; p < A_PER; ++p) rr[p] = (ft + p * 256) >> 2;
        const int c4 = (ft & 3) * 4;
        int adst[A_PER];
#pragma unroll
        for (int p = 0; p < A_PER; ++p) adst[p] = (c4 >> 3) * A_OCT + rr[p] * 16 + (c4 & 7) * 2;
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef int i4v __attribute__((ext_vector_type(4)));
        constexpr int NJ = GATHER ? 2 : 1;                                            // loads per row and chunk (GATHER: the row's two sources)
        f4v ra[PF_DA][A_PER][NJ];
        // Chunks are loaded in order: (tile, chunk) counters, the tile looked up once per tile through the scalar cache.  From the second
        // tile on, the tile's row scales and bias ride in the same counted queue (two loads issued right before the tile's first A loads,
        // i.e. older than them; written to LDS when that chunk is stored): no vmcnt(0) drain of the prefetch at tile boundaries.
        // GATHER: the A tile is the aggregate itself -- row r of the tile is w0 * X[u0] + w1 * X[u1] with {u0, u1, w0, w1} from the batch's
        // per-row source table (gm_batch::d_fuse2: rows of one or two sources; any other row reads its own, already aggregated, row of
        // `zside` with weights 1, 0), summed in the aggregate kernel's order (fma chain from 0) before the split.  The table entries of
        // tile t+1 are fetched when tile t's first loads are issued (same counted queue), a whole tile ahead of their use.
        const bool fast_consts = nchunks >= PF_DA;                                    // the const registers are reused every nchunks chunks
        int la_c = 0, la_ti = 0;
        const float* la_src[A_PER][NJ];
        float w_next[A_PER][2], w_cur[A_PER][2];
        i4v fq[A_PER];
#pragma unroll
        for (int p = 0; p < A_PER; ++p) { w_next[p][0] = w_cur[p][0] = 1.f; w_next[p][1] = w_cur[p][1] = 0.f; fq[p] = i4v{0, 0, 0, 0}; }
        float rc_sc = 1.f, rc_b = 0.f;
        float as_next = 1.f, as_cur = 1.f, inv_next = 1.f, inv_cur = 1.f;             // NP == 2: A scale of the tile being loaded / stored; 1 / (s_a s_b) for the epilogue
        auto fq_prefetch = [&](int ti) {                                              // table entries of tile ti (if any) -> fq, counted loads
            if (ti >= ntb) return;
            const GsTile t = tile_of(b + ti * G);
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                const int4* q = g.f2 + t.row0 + min(rr[p], t.nrows - 1);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fq[p]) : "v"(q) : "memory");
            }
        };
        auto la_tile = [&](int ti, bool consts) {
            const GsTile t = tile_of(b + ti * G);
            if constexpr (NP == 2) { as_next = a_scale_of(t.set); inv_next = (1.f / as_next) * (1.f / b_scale_of(t.set)); }
            if constexpr (GATHER) {
                if constexpr (A_PER == 2) asm volatile("" : "+v"(fq[0]), "+v"(fq[A_PER - 1]) :: "memory");   // fetched a tile ago; every wait since then covered them
                else asm volatile("" : "+v"(fq[0]) :: "memory");
#pragma unroll
                for (int p = 0; p < A_PER; ++p) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int u = j ? fq[p].y : fq[p].x;
                        const float* base = (u & GM_SPLIT_FUSE_ZERO) ? g.zrow : (u & GM_SPLIT_FUSE_SELF) ? g.zside + (int64_t)(u & ~GM_SPLIT_FUSE_SELF) * g.ldz : g.A + (int64_t)u * g.lda;
                        la_src[p][j] = base + c4;
                        w_next[p][j] = __int_as_float(j ? fq[p].w : fq[p].z);
                    }
                }
                fq_prefetch(ti + 1);
            } else {
#pragma unroll
                for (int p = 0; p < A_PER; ++p) la_src[p][0] = g.A + (int64_t)(t.row0 + min(rr[p], t.nrows - 1)) * g.lda + c4;
            }
            if (consts) {
                if (g.row_scale) { const float* q = g.row_scale + t.row0 + min(ft & (BM - 1), t.nrows - 1); asm volatile("global_load_dword %0, %1, off" : "=v"(rc_sc) : "v"(q) : "memory"); }
                if (g.bias) { const float* q = g.bias + (int64_t)t.set * g.bias_stride + min(ft, BN - 1); asm volatile("global_load_dword %0, %1, off" : "=v"(rc_b) : "v"(q) : "memory"); }
            }
        };
        if constexpr (GATHER) {
            fq_prefetch(0);
            if constexpr (A_PER == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fq[0]), "+v"(fq[A_PER - 1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fq[0]) :: "memory");
        }
        la_tile(0, false);
        as_cur = as_next; inv_cur = inv_next;
        if constexpr (GATHER) {
#pragma unroll
            for (int p = 0; p < A_PER; ++p) { w_cur[p][0] = w_next[p][0]; w_cur[p][1] = w_next[p][1]; }
        }
        auto load_a = [&](int slot) {                                                 // the next chunk in order
            if (la_c == nchunks) { la_c = 0; ++la_ti; la_tile(la_ti, fast_consts); }
#pragma unroll
            for (int p = 0; p < A_PER; ++p)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float* src = la_src[p][j] + la_c * BK;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[slot][p][j]) : "v"(src) : "memory");
                }
            ++la_c;
        };
        int sa_c = 0, sa_ti = 0;                                                      // (tile, chunk) of the next store_a
        auto store_a = [&](int gc, int slot) {
            if (sa_c == nchunks) {
                sa_c = 0; ++sa_ti;
                as_cur = as_next; inv_cur = inv_next;
                if constexpr (GATHER) {                                                // the loads of this tile were issued with w_next's table entries
#pragma unroll
                    for (int p = 0; p < A_PER; ++p) { w_cur[p][0] = w_next[p][0]; w_cur[p][1] = w_next[p][1]; }
                }
            }
            if (sa_c == 0 && sa_ti > 0 && fast_consts) {                               // first chunk of a later tile: its consts arrived with (before) this chunk's loads
                asm volatile("" : "+v"(rc_sc), "+v"(rc_b) :: "memory");
                if (ft < BM) scales[(sa_ti & 1) * GS_BM + ft] = rc_sc * inv_cur;
                if (ft < BN) biasl[(sa_ti & 1) * BN + ft] = rc_b;
            }
            ++sa_c;
            char* As = smem + (gc & 1) * A_ST;
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                uint2 h, m, l;
                f4v v = ra[slot][p][0];
                if constexpr (GATHER) {
                    const f4v v1 = ra[slot][p][NJ - 1];
                    const float w0 = w_cur[p][0], w1 = w_cur[p][1];
                    v.x = __fmaf_rn(v1.x, w1, __fmaf_rn(v.x, w0, 0.f)); v.y = __fmaf_rn(v1.y, w1, __fmaf_rn(v.y, w0, 0.f));
                    v.z = __fmaf_rn(v1.z, w1, __fmaf_rn(v.z, w0, 0.f)); v.w = __fmaf_rn(v1.w, w1, __fmaf_rn(v.w, w0, 0.f));
                }
                if constexpr (NP == 3) {
                    gs_split4(make_float4(v.x, v.y, v.z, v.w), h, m, l);
                    *reinterpret_cast<uint2*>(As + adst[p]) = h;
                    *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
                    *reinterpret_cast<uint2*>(As + 2 * A_PLANE + adst[p]) = l;
                } else {
                    gs_split4_f16(make_float4(v.x, v.y, v.z, v.w), as_cur, h, m);
                    *reinterpret_cast<uint2*>(As + adst[p]) = h;
                    *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
                }
            }
        };
        // row scales + bias of tile ti -> LDS (parity ti & 1); ordinary loads, completed with the vmcnt(0) below
        auto stage_tile_consts = [&](int ti) {
            const GsTile t = tile_of(b + ti * G);
            const int set = t.set, row0 = t.row0, nrows = t.nrows;
            float sc = 1.f;
            if (g.row_scale) sc = g.row_scale[row0 + min(ft & (BM - 1), nrows - 1)];
            float b0 = 0.f;
            if (g.bias) b0 = (g.bias + (int64_t)set * g.bias_stride)[min(ft, BN - 1)];
            if constexpr (NP == 2) sc *= (1.f / a_scale_of(set)) * (1.f / b_scale_of(set));
            if (ft < BM) scales[(ti & 1) * GS_BM + ft] = sc;
            if (ft < BN) biasl[(ti & 1) * BN + ft] = b0;
        };
        static_assert((A_PER == 1 || A_PER == 2) && PF_DA >= 2 && PF_DA <= 8, "wait macro is written for 1 or 2 rows per lane and chunk");
#define PF_WAIT_SLOT(NEWER, SLOT)                                                                                                              \
        do {                                                                                                                                   \
            if constexpr (GATHER && A_PER == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[SLOT][0][0]), "+v"(ra[SLOT][A_PER - 1][0]), "+v"(ra[SLOT][0][NJ - 1]), "+v"(ra[SLOT][A_PER - 1][NJ - 1]) : "n"((NEWER) * 4) : "memory"); \
            else if constexpr (GATHER) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ra[SLOT][0][0]), "+v"(ra[SLOT][0][NJ - 1]) : "n"((NEWER) * 2) : "memory"); \
            else if constexpr (A_PER == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ra[SLOT][0][0]), "+v"(ra[SLOT][A_PER - 1][0]) : "n"((NEWER) * 2) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[SLOT][0][0]) : "n"((NEWER) * 1) : "memory");                                 \
        } while (0)
        stage_tile_consts(0);                                                          // (compiler-managed loads: done before the asm loads below are counted)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < PF_DA; ++d) if (d < total) load_a(d);
        PF_WAIT_SLOT(0, 0);
        store_a(0, 0);
        GS_BARRIER();
        for (int g0 = 0; g0 < total; g0 += PF_DA) {
#pragma unroll
            for (int u = 0; u < PF_DA; ++u) {
                const int gc = g0 + u;
                if (gc < total) {
                    if (gc + 1 < total) {
                        // chunk gc+1 sits in slot (u+1) % PF_DA; newer groups in flight: chunks gc+2 .. min(gc+PF_DA-1, total-1)
                        const int newer = min(PF_DA - 2, total - 2 - gc);
                        const int SL = (u + 1) % PF_DA;
                        if (newer >= PF_DA - 2 && PF_DA >= 2) PF_WAIT_SLOT(PF_DA - 2, SL);
                        else if (newer == 1 && PF_DA > 3) PF_WAIT_SLOT(1, SL);
                        else if (newer == 2 && PF_DA > 4) PF_WAIT_SLOT(2, SL);
                        else PF_WAIT_SLOT(0, SL);
                        store_a(gc + 1, SL);
                        // the first chunk of the NEXT tile is about to become visible: its scales / bias must be there as well
                        // (short-K launches only: the counted path above needs nchunks >= PF_DA)
                        if (!fast_consts && sa_c == 1 && sa_ti > 0) { stage_tile_consts(sa_ti); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    }
                    if (gc + PF_DA < total) load_a(u);                                 // slot u held chunk gc: already split into LDS
                    GS_BARRIER();
                }
            }
        }
#undef PF_WAIT_SLOT
    } else {
        // ================= compute
        const int wr = wave / WC, wc = wave % WC, li = lane & 31, kh = lane >> 5;
        const int a_lane = kh * A_OCT + (wr * 32 * MI + li) * 16, b_lane = OFF_B + kh * B_OCT + (wc * 64 + li) * 16;
        int b_st = 0;                                                                 // B stage of the current chunk (gc % NB)
        GS_BARRIER();
        int gc = 0;
        float vmax = 0.f; int vmax_set = -1;                      // amax_out: running bound of this wave's blocks, flushed when the set changes
        // The eight compute waves walk the same tiles, so they flush at the same points: they meet in LDS and the last one to arrive makes the
        // ONE global request of the workgroup.  Only flushes BETWEEN two tiles go through the LDS pair: a wave cannot reach the next of those
        // before the last arrival has reset the pair (a whole tile of chunk barriers lies in between).  The flush after the last tile is
        // separated from a preceding set-change flush by the barrier-free epilogue only, so there every wave makes its own request.
        auto vmax_note = [&](unsigned bits) {
            gs_note_max(g.amax_out + (int64_t)vmax_set * GM_BOUND_PAD, bits);
            if (g.b_bound.viol && bits > 0x58800000u && bits < 0x7f800000u) atomicOr(g.b_bound.viol, GM_VIOL_RANGE);      // > 2^50: beyond the scale clamp's reach
        };
        auto vmax_flush = [&]() {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if (lane == 0) {
                unsigned* mx = reinterpret_cast<unsigned*>(smem + OFF_MAX);
                atomicMax(mx, __float_as_uint(vmax));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");              // the maximum before the arrival
                const unsigned ticket = atomicAdd(mx + 1, 1u);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (ticket == 7u) {
                    const unsigned all = atomicExch(mx, 0u);
                    mx[1] = 0u;
                    vmax_note(all);
                }
            }
            vmax = 0.f;
        };
        auto vmax_flush_last = [&]() {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if (lane == 0) vmax_note(__float_as_uint(vmax));
        };
        for (int ti = 0; ti < ntb; ++ti) {
            gm_f32x16 acc[MI][2];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int c = 0; c < nchunks; ++c, ++gc) {
                const char* S = smem + (gc & 1) * A_ST;
                const char* SB = smem + b_st * B_ST;
                b_st = b_st + 1 == NB ? 0 : b_st + 1;
                if constexpr (NP == 3) {
                    gm_bf16x8 af[MI][3], bf[2][3];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const gm_bf16x8*>(S + a_lane + p * A_PLANE + i * 512);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const gm_bf16x8*>(SB + b_lane + p * B_PLANE + j * 512);
#define PF_PROD(PA, PB)                                                                                                  \
                    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
                    PF_PROD(2, 0) PF_PROD(0, 2) PF_PROD(1, 1) PF_PROD(1, 0) PF_PROD(0, 1) PF_PROD(0, 0)
#undef PF_PROD
                } else {
                    gm_f16x8 af[MI][2], bf[2][2];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p) af[i][p] = *reinterpret_cast<const gm_f16x8*>(S + a_lane + p * A_PLANE + i * 512);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p) bf[j][p] = *reinterpret_cast<const gm_f16x8*>(SB + b_lane + p * B_PLANE + j * 512);
#define PF_PROD(PA, PB)                                                                                                  \
                    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
                    PF_PROD(1, 0) PF_PROD(0, 1) PF_PROD(0, 0)                       // smallest terms first
#undef PF_PROD
                }
                GS_BARRIER();
            }
            // ---- epilogue of tile ti: stores only (no global load, no barrier, no LDS staging).  The accumulators are transposed inside the
            // quads of lanes (gs_quad_transpose) so that a lane stores 16 bytes of ONE row and eight lanes cover 128 contiguous bytes of it; the
            // LDS round trips of the staged epilogue (8.6 k of a tile's 47 k cycles with the matrix pipe idle) are gone.
            const GsTile tl = tile_of(b + ti * G);                // scalar (SMEM) loads: lgkmcnt, not vmcnt
            const int row0 = tl.row0, nrows = tl.nrows;
            const float* sc_t = scales + (ti & 1) * GS_BM;
            const int t4 = li & 3, q4 = li >> 2;
            const bool odd1 = (li & 1) != 0, odd2 = (li & 2) != 0;
            float4 b4[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b4[j] = *reinterpret_cast<const float4*>(biasl + (ti & 1) * BN + wc * 64 + j * 32 + q4 * 4);
            if constexpr (NP == 2) { if (g.amax_out && tl.set != vmax_set) { if (vmax_set >= 0) vmax_flush(); vmax_set = tl.set; } }
            // keep_signed: the sign bit of a row's scale says that nobody reads the row (forward-only pass, last layer: the head gathers the
            // centre rows only).  A wave whose 32 MI rows are all of that kind has no epilogue at all.
            // (differentiated passes still leave the relu' bits / the zero fill of every row: no shortcut there, only the C stores go)
            const bool keep_only = g.keep_signed && !g.relu_bits && !g.zero_out;
            if (keep_only && __ballot(!(__float_as_uint(sc_t[wr * 32 * MI + (lane & (32 * MI - 1))]) >> 31)) == 0ull) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {                  // registers 4 gq .. 4 gq + 3: rows 8 gq + 4 kh + (0..3) of the wave's 32-row block i
                    const int rl = wr * 32 * MI + i * 32 + gq * 8 + kh * 4 + t4;
                    const float sc_raw = sc_t[rl];
                    const bool keep = !g.keep_signed || !(__float_as_uint(sc_raw) >> 31);
                    if (keep_only && __ballot(keep) == 0ull) continue;
                    const float sc = fabsf(sc_raw);
                    const int64_t row = row0 + rl;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float a0 = acc[i][j][gq * 4 + 0], a1 = acc[i][j][gq * 4 + 1], a2 = acc[i][j][gq * 4 + 2], a3 = acc[i][j][gq * 4 + 3];
                        gs_quad_transpose(a0, a1, a2, a3, odd1, odd2);
                        float4 v;
                        v.x = a0 * sc + b4[j].x; v.y = a1 * sc + b4[j].y; v.z = a2 * sc + b4[j].z; v.w = a3 * sc + b4[j].w;
                        if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                        if (rl >= nrows || (keep_only && !keep)) continue;
                        const int col = wc * 64 + j * 32 + q4 * 4;
                        if constexpr (NP == 2) { if (g.amax_out) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }
                        if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
                        if (!keep) {
                        } else if (g.nt_store) {
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            f4v vv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
                        } else {
                            *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
                        }
                        if (g.zero_out) {                       // zeros made here: a loop-invariant zero quad gets hoisted and, at 128 VGPRs, spilled
                            typedef float f4z __attribute__((ext_vector_type(4)));
                            f4z z;
                            asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z.x), "=v"(z.y), "=v"(z.z), "=v"(z.w));
                            *reinterpret_cast<f4z*>(g.zero_out + row * g.ldc + col) = z;
                        }
                    }
                }
            }
        }
        if constexpr (NP == 2) { if (g.amax_out && vmax_set >= 0) vmax_flush_last(); }     // (bit patterns of non-negative floats order as integers)
    }
}


