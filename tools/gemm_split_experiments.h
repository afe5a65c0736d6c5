// Prototype / ablation kernels of the split-bf16 GEMM (tools/gemm_split_bench.hip only; NOT part of the library).
// They document what was tried and measured in DESIGN.md section 4 and profiles/r02_split_gemm_ablation*.txt:
//   k_gemm_split<WC,BK>, k_gemm_split_fc   non-persistent structures (round 2, first half)
//   k_gemm_split_r                         three LDS stages + rolling fragment reloads under the MFMAs
//   k_gemm_split_h                         two 512-thread half-tile workgroups per CU
#pragma once
#include "../g-meta_amd/csrc/gemm_split.h"
#ifndef GS_EXPERIMENTS
#define GS_EXPERIMENTS
#endif

// C[rows of tile] = epi( A[rows, K] @ B_set ), block tile 128 x (64*WC), 2*WC waves, each wave a 64x64 sub-tile
// (2x2 MFMA 32x32 blocks; per 16-wide k chunk 6 bf16 MFMAs per block, issued product-major so that consecutive MFMAs hit
// different accumulators).  LDS holds two stages of
//   A planes [3][2][128 rows][8 k] bf16  +  B planes [3][2][BN n][8 k] bf16,
// i.e. an MFMA fragment (lane (row|col = l&31, k octet = l>>5) reads 16 B) is two contiguous 512-B runs and the 16 lanes
// ds_read_b128 services per LDS cycle cover all 64 banks (a [row][16 k] image is 2-way bank conflicted: 41 % of the LDS
// cycles in the first version).
// The two wave rows split the feeding work, and the two waves that share a SIMD are one of each kind (wave w and w + WC):
//   wave row 0: DMA of the B planes (L2 -> LDS, one chunk ahead; 6 one-KiB pieces per wave and chunk).  Measured: issuing
//               those pieces costs the issuing wave ~100 cycles each, during which its SIMD partner runs MFMAs.
//   wave row 1: the A rows, HBM -> registers GS_D chunks ahead -> exact 3-way split -> LDS.  Its vmcnt queue holds nothing
//               but these loads, so the in-order completion rule does not cut the prefetch depth short (with the DMA in
//               the same queue, waiting for a one-chunk-old DMA would also wait for every older A load).
// Requires K % (16*GS_D) == 0, N % (64*WC) == 0, lda % 4 == 0, 16-byte aligned A / C / bias.
#ifndef GS_OCC
#define GS_OCC
#endif
#ifndef GS_D
#define GS_D 4
#endif
template <int WC, int BK>
__global__ __launch_bounds__(128 * WC) GS_OCC void k_gemm_split(SplitGemmK g) {
    static_assert(BK == 16, "one MFMA k step per chunk");
    constexpr int NW = 2 * WC, BN = 64 * WC, KS = 1;
    constexpr int A_OCT = GS_BM * 16, B_OCT = BN * 16;                              // bytes of one [rows][8 k] bf16 octet slab
    constexpr int A_SLAB = 2 * A_OCT, B_SLAB = 2 * B_OCT;                           // one MFMA k step (16 k)
    constexpr int A_PLANE = KS * A_SLAB, B_PLANE = KS * B_SLAB;
    constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;
    constexpr int EP_LD = 68, EPI_BYTES = NW * 32 * EP_LD * 4;
    constexpr int MAIN_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
    constexpr int A_PER = (GS_BM * BK / 4) / (64 * WC);                            // float4 of the A tile per thread of wave row 1
    constexpr int B_PPW = (3 * 2 * (BN / 64)) / WC;                                // 1-KiB DMA pieces (64 n x 8 k) per wave of wave row 0: 6
    __shared__ __attribute__((aligned(16))) char smem[MAIN_BYTES + GS_BM * 4];
    const int nb = g.n_tiles * g.n_col_tiles, b = blockIdx.x;
    const int q = nb / 8, r = nb % 8, xcd = b % 8, idx = b / 8;                   // XCD-contiguous logical ids
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = lb / g.n_col_tiles, ct = lb % g.n_col_tiles;
    const int set = g.tiles[tile * 3], row0 = g.tiles[tile * 3 + 1], nrows = g.tiles[tile * 3 + 2];
    const int n0 = ct * BN;
    const uint16_t* Bt = g.Bt + (int64_t)set * g.bt_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int li = lane & 31, kh = lane >> 5;
    float my_scale = 1.f;
    if (tid < GS_BM && g.row_scale) my_scale = g.row_scale[row0 + min(tid, nrows - 1)];
    float* scales = reinterpret_cast<float*>(smem + MAIN_BYTES);
    // ---- A (wave row 1): this thread's float4 slots of the [128][16] tile (row = id/4, k = (id%4)*4); rows past the
    // tile end are clamped (their products are never stored)
    const float* asrc[A_PER]; int adst[A_PER];
#pragma unroll
    for (int p = 0; p < A_PER; ++p) {
        const int id = (tid & (64 * WC - 1)) + p * (64 * WC), rr = id >> 2, c4 = (id & 3) * 4;
        asrc[p] = g.A + (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + c4;
        adst[p] = (c4 >> 3) * A_OCT + rr * 16 + (c4 & 7) * 2;
    }
    // ---- B (wave row 0): piece id = (plane * 2 + octet) * (BN/64) + colblock; wave wc takes pieces wc*B_PPW .. +B_PPW-1.
    // Address = scalar base (set, chunk) + a per-lane 32-bit byte offset: one VGPR per piece, the base advances on the SALU.
    unsigned boff[B_PPW]; int bdst[B_PPW];
#pragma unroll
    for (int p = 0; p < B_PPW; ++p) {
        const int piece = wc * B_PPW + p, cb = piece % (BN / 64), po = piece / (BN / 64), oct = po % 2, plane = po / 2;
        boff[p] = (unsigned)(((int64_t)plane * g.N * g.K + ((int64_t)oct * g.N + n0 + cb * 64 + lane) * 8) * 2);
        bdst[p] = 3 * A_PLANE + plane * B_PLANE + oct * B_OCT + cb * 1024;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int64_t b_chunk_bytes = (int64_t)BK * g.N * 2;                           // one 16-k chunk of a plane
    auto issue_b = [&](int chunk, int buf) {
#ifdef GS_EXP_NOB
        return;
#endif
        const uint64_t base = (uint64_t)(uintptr_t)Bt + (uint64_t)(chunk * b_chunk_bytes);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
#pragma unroll
        for (int p = 0; p < B_PPW; ++p) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE + bdst[p]));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(boff[p]), "s"(sbase), "s"(dst) : "memory");
        }
    };
    float4 ra[GS_D][A_PER];
    auto load_a = [&](int chunk, int slot) {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
#ifdef GS_EXP_NOA
            ra[slot][p] = make_float4(1.f + chunk, 2.f, 3.f, 4.f);
#else
            ra[slot][p] = *reinterpret_cast<const float4*>(asrc[p] + chunk * BK);      // (outer-loop base + compile-time offset after unrolling)
#endif
        }
    };
    auto store_a = [&](int buf, int slot) {
#ifdef GS_EXP_NOSPLIT
        if (ra[slot][0].x != 123.456f) return;
#endif
        char* As = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            uint2 h, m, l;
            gs_split4(ra[slot][p], h, m, l);
            *reinterpret_cast<uint2*>(As + adst[p]) = h;
            *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
            *reinterpret_cast<uint2*>(As + 2 * A_PLANE + adst[p]) = l;
        }
    };
    gm_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = g.K / BK;                      // a multiple of GS_D
    if (wr == 0) issue_b(0, 0);
    else {
#pragma unroll
        for (int d = 0; d < GS_D; ++d) load_a(d, d);
        store_a(0, 0);
    }
    if (tid < GS_BM) scales[tid] = my_scale;
    if (wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += GS_D) {
#pragma unroll
        for (int u = 0; u < GS_D; ++u) {
            const int c = c0 + u, buf = u & 1;                                     // GS_D is even: stage parity == u parity
            if (wr == 0) { if (c + 1 < nchunks) issue_b(c + 1, buf ^ 1); }         // the other stage was last read before the previous barrier
            else if (c + GS_D < nchunks) load_a(c + GS_D, u);                      // slot u held chunk c: split into LDS one iteration ago
            const char* As = smem + buf * STAGE;
            const char* Bs = As + 3 * A_PLANE;
#ifndef GS_EXP_NOMFMA
            gm_bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[i][p] = *reinterpret_cast<const gm_bf16x8*>(As + p * A_PLANE + kh * A_OCT + (wr * 64 + i * 32 + li) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[j][p] = *reinterpret_cast<const gm_bf16x8*>(Bs + p * B_PLANE + kh * B_OCT + (wc * 64 + j * 32 + li) * 16);
            // product-major: the four accumulators take turns (a dependent MFMA waits for its predecessor's result)
#define GS_PROD(PA, PB)                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
            GS_PROD(2, 0) GS_PROD(0, 2) GS_PROD(1, 1) GS_PROD(1, 0) GS_PROD(0, 1) GS_PROD(0, 0)      // smallest terms first
#undef GS_PROD
#else
            if (c == 1000) acc[0][0][0] = *reinterpret_cast<const float*>(As + lane * 4) + *reinterpret_cast<const float*>(Bs + lane * 4);
#endif
            if (wr == 0) {
#ifndef GS_EXP_BNOWAIT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            } else if (c + 1 < nchunks) store_a(buf ^ 1, (u + 1) % GS_D);
            GS_BARRIER();
        }
    }
    // ---- epilogue: each wave's 32x64 halves through LDS (the stages are free now), row-contiguous 16-B stores.
    // (Swapped operands + direct row-per-lane 16-B stores were tried: 32-byte runs per row cost +0.4 ms on the 1.1 M-row launch.)
    const float* biasp = g.bias ? g.bias + (int64_t)set * g.bias_stride : nullptr;
    float* E = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
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
            const float sc = scales[rl];
            float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
            v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
            if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }   // NaN propagates like torch relu
            if (g.mask_b) {
                const unsigned m = g.mask_b[(row * g.ldc + col) >> 2];
                v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
            } else if (g.mask_h) {
                const float4 m = *reinterpret_cast<const float4*>(g.mask_h + row * g.ldc + col);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
#ifdef GS_EXP_NOSTORE
            if (v.x != 123.456f) continue;
#endif
            if (g.nt_store) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
            } else {
                *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant (experiment): 12 waves per 128 x 256 tile, one workgroup per CU.
//   waves 0..7   COMPUTE: 64x64 sub-tiles, LDS fragment reads + MFMAs only; they never touch global memory in the main loop,
//                so nothing but the per-chunk barrier can stall their MFMA stream; the fragments of the chunk's second k step
//                are fetched while the first step's MFMAs run (BK = 32: two steps per barrier).
//   waves 8,9    A FEEDERS: HBM -> registers FC_DA chunks ahead -> exact split -> LDS planes (their vmcnt queue holds only
//                these loads: deep, in-order prefetch).
//   waves 10,11  B FEEDERS: DMA of the B planes, one chunk ahead (24 one-KiB pieces per wave and chunk).
// Two LDS stages of 72 KiB.  Requires K % 32 == 0, N == 256 column tiles, lda % 4 == 0.
#ifndef FC_DA
#define FC_DA 3
#endif
#ifndef FC_NB
#define FC_NB 2
#endif
__global__ __launch_bounds__(640 + 64 * FC_NB) void k_gemm_split_fc(SplitGemmK g) {
    constexpr int BK = 32, BN = 256, KS = 2, WC = 4;
    constexpr int A_OCT = GS_BM * 16, B_OCT = BN * 16, A_SLAB = 2 * A_OCT, B_SLAB = 2 * B_OCT;
    constexpr int A_PLANE = KS * A_SLAB, B_PLANE = KS * B_SLAB, STAGE = 3 * A_PLANE + 3 * B_PLANE;          // 24 + 48 KiB
    constexpr int EP_LD = 68;
    constexpr int A_PER = (GS_BM * BK / 4) / 128;                   // float4 per A-feeder lane and chunk: 8
    constexpr int B_PPW = (3 * (BK / 8) * (BN / 64)) / FC_NB;       // DMA pieces per B-feeder wave and chunk: 24 with two feeders
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + GS_BM * 4];
    const int nb = g.n_tiles * g.n_col_tiles, b = blockIdx.x;
    const int q = nb / 8, r = nb % 8, xcd = b % 8, idx = b / 8;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = lb / g.n_col_tiles, ct = lb % g.n_col_tiles;
    const int set = g.tiles[tile * 3], row0 = g.tiles[tile * 3 + 1], nrows = g.tiles[tile * 3 + 2];
    const int n0 = ct * BN;
    const uint16_t* Bt = g.Bt + (int64_t)set * g.bt_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = g.K / BK;
    float* scales = reinterpret_cast<float*>(smem + 2 * STAGE);
    if (tid < GS_BM) scales[tid] = g.row_scale ? g.row_scale[row0 + min(tid, nrows - 1)] : 1.f;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
#ifdef FC_TRACE
    const bool tr = (blockIdx.x == 2048) && lane == 0 && (wave == 0 || wave == 8 || wave == 10);
    const int trole = wave == 0 ? 0 : wave == 8 ? 1 : 2;
#define FC_T(C, WHICH) do { if (tr) g.dbg[((trole * 64 + (C)) * 4) + (WHICH)] = clock64(); } while (0)
#else
#define FC_T(C, WHICH) do {} while (0)
#endif

    if (wave >= 10) {
        // ================= B feeder
        const int fw = wave - 10;
        unsigned boff[B_PPW]; int bdst[B_PPW];
#pragma unroll
        for (int p = 0; p < B_PPW; ++p) {
            const int piece = fw * B_PPW + p, cb = piece % (BN / 64), po = piece / (BN / 64), oct = po % (BK / 8), plane = po / (BK / 8);
            boff[p] = (unsigned)(((int64_t)plane * g.N * g.K + ((int64_t)oct * g.N + n0 + cb * 64 + lane) * 8) * 2);
            bdst[p] = 3 * A_PLANE + plane * B_PLANE + oct * B_OCT + cb * 1024;
        }
        const int64_t b_chunk_bytes = (int64_t)BK * g.N * 2;
        auto issue_b = [&](int chunk, int buf) {
#ifdef FC_EXP_NOB
            return;
#endif
            const uint64_t base = (uint64_t)(uintptr_t)Bt + (uint64_t)(chunk * b_chunk_bytes);
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
            const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE + bdst[p]));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(boff[p]), "s"(sbase), "s"(dst) : "memory");
            }
        };
#ifdef FC_B_REGS
        // register path: global_load_dwordx4 -> ds_write_b128 (issue ~30-40 cycles per KiB against ~100 for an LDS-DMA piece)
        auto fetch_b = [&](int chunk, int buf) {
            float4 rb[B_PPW];
            const char* base = reinterpret_cast<const char*>(Bt) + chunk * b_chunk_bytes;
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) rb[p] = *reinterpret_cast<const float4*>(base + boff[p]);
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) *reinterpret_cast<float4*>(smem + buf * STAGE + bdst[p] + lane * 16) = rb[p];
        };
        fetch_b(0, 0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) fetch_b(c + 1, (c + 1) & 1);
            __syncthreads();
        }
#else
        issue_b(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            FC_T(c, 0);
            if (c + 1 < nchunks) issue_b(c + 1, (c + 1) & 1);
            FC_T(c, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FC_T(c, 2);
            GS_BARRIER();
            FC_T(c, 3);
        }
#endif
    } else if (wave >= 8) {
        // ================= A feeder
        const int ft = tid - 512;                                    // 0..127
        const float* asrc[A_PER]; int adst[A_PER];
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int id = ft + p * 128, rr = id / (BK / 4), c4 = (id % (BK / 4)) * 4;
            asrc[p] = g.A + (int64_t)(row0 + min(rr, nrows - 1)) * g.lda + c4;
            adst[p] = (c4 >> 3) * A_OCT + rr * 16 + (c4 & 7) * 2;
        }
        // The loads are issued through inline asm and their completion is counted by hand: with compiler-managed waits a
        // register prefetch deeper than one chunk degenerates (hipcc's vmcnt model goes conservative at the loop back-edge and
        // waits for the NEWEST loads before it touches the oldest slot -- measured: prefetch depth 2, 4 and 8 all ran alike).
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v ra[FC_DA][A_PER];
        auto load_a = [&](int chunk, int slot) {
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
#ifdef FC_EXP_NOA
                ra[slot][p] = f4v{1.f + chunk, 2.f, 3.f, 4.f};
#else
                #ifdef FC_A_NT
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(ra[slot][p]) : "v"(asrc[p] + chunk * BK) : "memory");      // streamed once: keep L2 for the B planes
#else
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[slot][p]) : "v"(asrc[p] + chunk * BK) : "memory");
#endif
#endif
            }
        };
        // wait until at most `newer` chunk groups (A_PER loads each) are outstanding, then the slot's registers are valid
#define FC_WAIT_SLOT(NEWER, SLOT)                                                                                              \
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ra[SLOT][0]), "+v"(ra[SLOT][1]), "+v"(ra[SLOT][2]), "+v"(ra[SLOT][3]),      \
                     "+v"(ra[SLOT][4]), "+v"(ra[SLOT][5]), "+v"(ra[SLOT][6]), "+v"(ra[SLOT][7]) : "n"((NEWER) * A_PER) : "memory")
        auto wait_slot = [&](int newer, int slot_static) {};
        (void)wait_slot;
        auto store_a = [&](int buf, int slot) {
            char* As = smem + buf * STAGE;
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                uint2 h, m, l;
                const f4v v = ra[slot][p];
                gs_split4(make_float4(v.x, v.y, v.z, v.w), h, m, l);
                *reinterpret_cast<uint2*>(As + adst[p]) = h;
                *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
                *reinterpret_cast<uint2*>(As + 2 * A_PLANE + adst[p]) = l;
            }
        };
        static_assert(A_PER == 8 && FC_DA >= 2 && FC_DA <= 8, "wait macro is written for 8 loads per chunk");
#pragma unroll
        for (int d = 0; d < FC_DA; ++d) if (d < nchunks) load_a(d, d);
#ifndef FC_EXP_NOA
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[0][2]), "+v"(ra[0][3]), "+v"(ra[0][4]), "+v"(ra[0][5]), "+v"(ra[0][6]), "+v"(ra[0][7]) :: "memory");
#endif
        store_a(0, 0);
        __syncthreads();
        for (int c0 = 0; c0 < nchunks; c0 += FC_DA) {
#pragma unroll
            for (int u = 0; u < FC_DA; ++u) {
                const int c = c0 + u;
                if (c < nchunks) {
                    FC_T(c, 0);
                    if (c + FC_DA < nchunks) load_a(c + FC_DA, u);                 // slot u held chunk c (already in LDS)
                    FC_T(c, 1);
                    if (c + 1 < nchunks) {
#ifndef FC_EXP_NOA
                        // chunk c+1 sits in slot (u+1)%FC_DA; newer groups outstanding: chunks c+2 .. min(c+FC_DA, nchunks-1)
                        const int newer = min(FC_DA - 1, nchunks - 2 - c);
                        const int SL = (u + 1) % FC_DA;
                        if (newer >= FC_DA - 1) FC_WAIT_SLOT(FC_DA - 1, SL);
                        else if (newer == 1 && FC_DA > 2) FC_WAIT_SLOT(1, SL);
                        else if (newer == 2 && FC_DA > 3) FC_WAIT_SLOT(2, SL);
                        else if (newer == 3 && FC_DA > 4) FC_WAIT_SLOT(3, SL);
                        else FC_WAIT_SLOT(0, SL);
#endif
                        store_a((c + 1) & 1, (u + 1) % FC_DA);
                    }
                    FC_T(c, 2);
                    GS_BARRIER();
                    FC_T(c, 3);
                }
            }
        }
#undef FC_WAIT_SLOT
    } else {
        // ================= compute
        const int wr = wave / WC, wc = wave % WC, li = lane & 31, kh = lane >> 5;
        gm_f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int a_lane = kh * A_OCT + (wr * 64 + li) * 16, b_lane = 3 * A_PLANE + kh * B_OCT + (wc * 64 + li) * 16;
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            FC_T(c, 0);
            const char* S = smem + (c & 1) * STAGE;
            gm_bf16x8 af[2][2][3], bf[2][2][3];                      // [k step][block][plane]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[ks][i][p] = *reinterpret_cast<const gm_bf16x8*>(S + a_lane + p * A_PLANE + ks * A_SLAB + i * 512);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[ks][j][p] = *reinterpret_cast<const gm_bf16x8*>(S + b_lane + p * B_PLANE + ks * B_SLAB + j * 512);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#define FC_PROD(PA, PB)                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PA], bf[ks][j][PB], acc[i][j], 0, 0, 0);
                FC_PROD(2, 0) FC_PROD(0, 2) FC_PROD(1, 1) FC_PROD(1, 0) FC_PROD(0, 1) FC_PROD(0, 0)
#undef FC_PROD
            }
            FC_T(c, 2);
            GS_BARRIER();
            FC_T(c, 3);
        }
        // epilogue (compute waves only; the feeders are done): 32x64 halves through LDS, row-contiguous 16-B stores
        const float* biasp = g.bias ? g.bias + (int64_t)set * g.bias_stride : nullptr;
        float* E = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
        const int er = lane >> 4, ec = (lane & 15) * 4;
        const int col = n0 + wc * 64 + ec;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (biasp) b4 = *reinterpret_cast<const float4*>(biasp + col);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * kh) * EP_LD + j * 32 + li] = acc[i][j][e];
            // each wave reads back only its OWN staging region: no workgroup barrier needed (wave-level LDS ordering suffices)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rl = wr * 64 + i * 32 + it * 4 + er;
                if (rl >= nrows) continue;
                const int64_t row = row0 + rl;
                const float sc = scales[rl];
                float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
                v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
                if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
#ifdef FC_EXP_NOSTORE
                if (v.x != 123.456f) continue;
#endif
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (tools/gemm_split_bench.hip mode 4): correct, but 0.99 ms against 0.97 ms for k_gemm_split_p on the 1.15 M x 256 x 256
// product -- spreading the LDS reads under the MFMAs does not help, the kernel is not LDS-burst-bound either (profiles/r02_split_gemm_ablation2.txt).
// k_gemm_split_r: the persistent kernel above with a THREE-stage LDS ring and ROLLING fragment reloads in the compute waves.
// In k_gemm_split_p every compute wave reads its 12 fragments right after the chunk barrier -- all eight waves at once, ~500 cycles
// of LDS traffic during which no MFMA runs -- and then all of them issue MFMAs (~1540 cycles) while the LDS idles (in-kernel
// timestamps: 2200-cycle chunks).  Here the feeders run TWO chunks ahead, so the fragments of chunk c+1 are already in LDS while
// chunk c is being multiplied: a compute wave reloads each fragment register with the next chunk's value right after the last MFMA
// of this chunk that reads it (product order (0,0) (0,1) (0,2) (1,0) (2,0) (1,1): a_h / b_l die after the third product group,
// b_h / a_l after the fifth, a_m / b_m after the last) -- no extra registers, the LDS reads spread under the MFMAs, and the compute
// waves' chunk barrier needs no LDS wait (what they read in chunk c is not overwritten before the barrier after chunk c+1).
// LDS: 3 stages x 36 KiB + 8 x 4.25 KiB staging (16-row passes) + scales/bias = 145 KiB.
__global__ __launch_bounds__(1024) void k_gemm_split_r(SplitGemmK g) {
    constexpr int BK = 16, BN = 256, WC = 4;
    constexpr int A_OCT = GS_BM * 16, B_OCT = BN * 16, A_PLANE = 2 * A_OCT, B_PLANE = 2 * B_OCT;
    constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;                                  // 12 + 24 KiB
    constexpr int NST = 3;
    constexpr int EP_LD = 68, E_WAVE = 16 * EP_LD * 4;                                // 4352 B per compute wave
    constexpr int OFF_E = NST * STAGE, OFF_SC = OFF_E + 8 * E_WAVE, OFF_BIAS = OFF_SC + 2 * GS_BM * 4;
    constexpr int A_PER = (GS_BM * BK / 4) / 256;                                     // float4 per A-feeder lane and chunk: 2
    constexpr int B_PPW = (3 * 2 * (BN / 64)) / 4;                                    // DMA pieces per B-feeder wave and chunk: 6
    __shared__ __attribute__((aligned(16))) char smem[OFF_BIAS + 2 * BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = g.K / BK;
    const int G = gridDim.x, b = blockIdx.x;
    const int nb = g.n_tiles, q8 = nb / 8, r8 = nb % 8;
    auto logical = [&](int t) -> int { const int x = t % 8, i = t / 8; return (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; };
    const int ntb = b < nb ? (nb - b + G - 1) / G : 0;                                // tiles of this workgroup
    const int total = ntb * nchunks;                                                  // flattened chunk count
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    float* scales = reinterpret_cast<float*>(smem + OFF_SC);                          // [2][128] by tile parity
    float* biasl = reinterpret_cast<float*>(smem + OFF_BIAS);                         // [2][256]
    if (total == 0) return;

    if (wave >= 12) {
        // ================= B feeder: chunk k -> stage k % 3, two chunks ahead of the compute waves
        __builtin_amdgcn_s_setprio(2);
        const int fw = wave - 12;
        unsigned boff[B_PPW]; int bdst[B_PPW];
#pragma unroll
        for (int p = 0; p < B_PPW; ++p) {
            const int piece = fw * B_PPW + p, cb = piece % (BN / 64), po = piece / (BN / 64), oct = po % 2, plane = po / 2;
            boff[p] = (unsigned)(((int64_t)plane * g.N * g.K + ((int64_t)oct * g.N + cb * 64 + lane) * 8) * 2);
            bdst[p] = 3 * A_PLANE + plane * B_PLANE + oct * B_OCT + cb * 1024;
        }
        const int64_t b_chunk_bytes = (int64_t)BK * g.N * 2;
        int ib_c = 0, ib_ti = 0, ib_st = 0;
        uint64_t ib_base = (uint64_t)(uintptr_t)(g.Bt + (int64_t)gs_tile(g.tiles, logical(b)).set * g.bt_stride);
        auto issue_b = [&]() {                                                        // the next chunk in order
            if (ib_c == nchunks) { ib_c = 0; ++ib_ti; ib_base = (uint64_t)(uintptr_t)(g.Bt + (int64_t)gs_tile(g.tiles, logical(b + ib_ti * G)).set * g.bt_stride); }
            const uint64_t base = ib_base + (uint64_t)(ib_c * b_chunk_bytes);
            ++ib_c;
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
            const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ib_st * STAGE + bdst[p]));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(boff[p]), "s"(sbase), "s"(dst) : "memory");
            }
            ib_st = ib_st == NST - 1 ? 0 : ib_st + 1;
        };
        issue_b();
        if (total > 1) issue_b();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GS_BARRIER();
        for (int gc = 0; gc < total; ++gc) {
            if (gc + 2 < total) issue_b();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GS_BARRIER();
        }
    } else if (wave >= 8) {
        // ================= A feeder: registers PF_DA chunks deep, chunk k -> stage k % 3, stored two chunks ahead of the compute waves
        __builtin_amdgcn_s_setprio(3);
        const int ft = tid - 512;                                                     // 0..255
        const int rr[A_PER] = {ft >> 2, (ft + 256) >> 2};
        const int c4 = (ft & 3) * 4;
        int adst[A_PER];
#pragma unroll
        for (int p = 0; p < A_PER; ++p) adst[p] = (c4 >> 3) * A_OCT + rr[p] * 16 + (c4 & 7) * 2;
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v ra[PF_DA][A_PER];
        const bool fast_consts = nchunks >= PF_DA;
        int la_c = 0, la_ti = 0;
        const float* la_src[A_PER];
        float rc_sc = 1.f, rc_b = 0.f;
        auto la_tile = [&](int ti, bool consts) {
            const GsTile t = gs_tile(g.tiles, logical(b + ti * G));
#pragma unroll
            for (int p = 0; p < A_PER; ++p) la_src[p] = g.A + (int64_t)(t.row0 + min(rr[p], t.nrows - 1)) * g.lda + c4;
            if (consts) {
                if (g.row_scale) { const float* q = g.row_scale + t.row0 + min(ft & 127, t.nrows - 1); asm volatile("global_load_dword %0, %1, off" : "=v"(rc_sc) : "v"(q) : "memory"); }
                if (g.bias) { const float* q = g.bias + (int64_t)t.set * g.bias_stride + ft; asm volatile("global_load_dword %0, %1, off" : "=v"(rc_b) : "v"(q) : "memory"); }
            }
        };
        la_tile(0, false);
        auto load_a = [&](int slot) {                                                 // the next chunk in order
            if (la_c == nchunks) { la_c = 0; ++la_ti; la_tile(la_ti, fast_consts); }
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                const float* src = la_src[p] + la_c * BK;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[slot][p]) : "v"(src) : "memory");
            }
            ++la_c;
        };
        auto stage_tile_consts = [&](int ti) {                                        // short-K launches and the first tile: plain loads, drained by the caller
            const GsTile t = gs_tile(g.tiles, logical(b + ti * G));
            float sc = 1.f;
            if (g.row_scale) sc = g.row_scale[t.row0 + min(ft & 127, t.nrows - 1)];
            float b0 = 0.f;
            if (g.bias) b0 = (g.bias + (int64_t)t.set * g.bias_stride)[ft];
            if (ft < GS_BM) scales[(ti & 1) * GS_BM + ft] = sc;
            biasl[(ti & 1) * BN + ft] = b0;
        };
        int sa_c = 0, sa_ti = 0, sa_st = 0;                                           // (tile, chunk, stage) of the next store
        auto store_a = [&](int slot) {
            if (sa_c == nchunks) { sa_c = 0; ++sa_ti; }
            if (sa_c == 0 && sa_ti > 0) {
                if (fast_consts) {                                                     // arrived with (before) this chunk's loads
                    asm volatile("" : "+v"(rc_sc), "+v"(rc_b) :: "memory");
                    if (ft < GS_BM) scales[(sa_ti & 1) * GS_BM + ft] = rc_sc;
                    biasl[(sa_ti & 1) * BN + ft] = rc_b;
                } else { stage_tile_consts(sa_ti); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            ++sa_c;
            char* As = smem + sa_st * STAGE;
            sa_st = sa_st == NST - 1 ? 0 : sa_st + 1;
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                uint2 h, m, l;
                const f4v v = ra[slot][p];
                gs_split4(make_float4(v.x, v.y, v.z, v.w), h, m, l);
                *reinterpret_cast<uint2*>(As + adst[p]) = h;
                *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
                *reinterpret_cast<uint2*>(As + 2 * A_PLANE + adst[p]) = l;
            }
        };
        static_assert(A_PER == 2 && PF_DA == 4, "wait macro / unrolling are written for 2 loads per chunk, 4 chunks deep");
#define PR_WAIT_SLOT(NEWER, SLOT) \
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ra[SLOT][0]), "+v"(ra[SLOT][1]) : "n"((NEWER) * A_PER) : "memory")
        // chunk k sits in slot k % PF_DA; when it is stored, chunks k+1 .. min(k+PF_DA-1, total-1) are the newer groups in flight
#define PR_DO_STORE(K_, SLOT)                                                                         \
        do {                                                                                          \
            const int newer_ = min(PF_DA - 1, total - 1 - (K_));                                     \
            if (newer_ >= 3) PR_WAIT_SLOT(3, SLOT); else if (newer_ == 2) PR_WAIT_SLOT(2, SLOT);      \
            else if (newer_ == 1) PR_WAIT_SLOT(1, SLOT); else PR_WAIT_SLOT(0, SLOT);                  \
            store_a(SLOT);                                                                            \
            if ((K_) + PF_DA < total) load_a(SLOT);                                                   \
        } while (0)
        stage_tile_consts(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < PF_DA; ++d) if (d < total) load_a(d);
        PR_DO_STORE(0, 0);
        if (total > 1) PR_DO_STORE(1, 1);
        GS_BARRIER();
        for (int g0 = 0; g0 < total; g0 += PF_DA) {
#pragma unroll
            for (int u = 0; u < PF_DA; ++u) {
                const int gc = g0 + u;
                if (gc < total) {
                    if (gc + 2 < total) PR_DO_STORE(gc + 2, (u + 2) % PF_DA);
                    GS_BARRIER();
                }
            }
        }
#undef PR_DO_STORE
#undef PR_WAIT_SLOT
    } else {
        // ================= compute
        const int wr = wave / WC, wc = wave % WC, li = lane & 31, kh = lane >> 5;
        const int a_lane = kh * A_OCT + (wr * 64 + li) * 16, b_lane = 3 * A_PLANE + kh * B_OCT + (wc * 64 + li) * 16;
        float* E = reinterpret_cast<float*>(smem + OFF_E + wave * E_WAVE);
        const int er = lane >> 4, ec = (lane & 15) * 4;
        GS_BARRIER();                                                                  // chunks 0 and 1 are staged
        gm_bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[i][p] = *reinterpret_cast<const gm_bf16x8*>(smem + a_lane + p * A_PLANE + i * 512);
                bf[i][p] = *reinterpret_cast<const gm_bf16x8*>(smem + b_lane + p * B_PLANE + i * 512);
            }
        int gc = 0, nst = 1;                                                           // nst: stage of chunk gc + 1
        for (int ti = 0; ti < ntb; ++ti) {
            gm_f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int c = 0; c < nchunks; ++c, ++gc) {
                const char* Sn = smem + nst * STAGE;                                   // next chunk's stage (stale after the last chunk: read, never used)
                nst = nst == NST - 1 ? 0 : nst + 1;
#define PR_PROD(PA, PB)                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
#define PR_RELOAD_A(P) _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i][P] = *reinterpret_cast<const gm_bf16x8*>(Sn + a_lane + (P) * A_PLANE + i * 512);
#define PR_RELOAD_B(P) _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[j][P] = *reinterpret_cast<const gm_bf16x8*>(Sn + b_lane + (P) * B_PLANE + j * 512);
                PR_PROD(0, 0) PR_PROD(0, 1) PR_PROD(0, 2)
                PR_RELOAD_A(0) PR_RELOAD_B(2)
                PR_PROD(1, 0) PR_PROD(2, 0)
                PR_RELOAD_B(0) PR_RELOAD_A(2)
                PR_PROD(1, 1)
                PR_RELOAD_A(1) PR_RELOAD_B(1)
#undef PR_PROD
#undef PR_RELOAD_A
#undef PR_RELOAD_B
                __builtin_amdgcn_s_barrier();                                          // no LDS wait: see the header comment
            }
            // ---- epilogue of tile ti: wave-private staging in 16-row passes, stores only (no global load, no barrier)
            const GsTile tl = gs_tile(g.tiles, logical(b + ti * G));
            const int row0 = tl.row0, nrows = tl.nrows;
            const float* sc_t = scales + (ti & 1) * GS_BM;
            const int col = wc * 64 + ec;
            const float4 b4 = *reinterpret_cast<const float4*>(biasl + (ti & 1) * BN + col);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 8 * hh; e < 8 * hh + 8; ++e) E[((e & 3) + 8 * ((e >> 2) & 1) + 4 * kh) * EP_LD + j * 32 + li] = acc[i][j][e];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rl = wr * 64 + i * 32 + hh * 16 + it * 4 + er;
                        if (rl >= nrows) continue;
                        const int64_t row = row0 + rl;
                        const float sc = sc_t[rl];
                        float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
                        v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
                        if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                        if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
                        if (g.nt_store) {
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            f4v vv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
                        } else {
                            *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (tools/gemm_split_bench.hip mode 3; measured 1.12-1.13 ms against 0.99 ms for k_gemm_split_p on the 1.15 M x 256 x 256
// product, whatever the de-phasing: the doubled A feeding -- loads, split VALU work, LDS writes -- costs more than the hidden store
// epilogue returns; grids that are not a multiple of 16 are not handled).  Kept for the ablation table, not built into the library.
// TWO workgroups per CU, each a 512-thread copy of the structure above on a 128 x 128 half tile (N == 256 split in two):
//   waves 0..3  COMPUTE (64 x 64 sub-tiles)   waves 4,5  A FEEDERS   waves 6,7  B FEEDERS
// The persistent kernel above loses ~18 % of a tile's lifetime to its store epilogue (the compute waves wait for the CU's write
// path and the feeders wait for the compute waves).  With two independent half-size workgroups on a CU, one's epilogue runs under
// the other's MFMAs; the second-slot workgroups start half a tile late so that the pair stays out of phase.  The two halves of a
// row tile are walked by neighbouring workgroups of the same XCD at about the same time (the second read of the A rows is an L2 /
// Infinity-Cache hit).  LDS: 2 stages x 24 KiB + 4 x 4.25 KiB staging (16-row passes) + scales/bias = 67 KiB per workgroup.
#ifndef PH_DA
#define PH_DA 4
#endif
__global__ __launch_bounds__(512, 2) void k_gemm_split_h(SplitGemmK g) {
    constexpr int BK = 16, BN = 128, WC = 2;
    constexpr int A_OCT = GS_BM * 16, B_OCT = BN * 16, A_PLANE = 2 * A_OCT, B_PLANE = 2 * B_OCT;
    constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;                                  // 12 + 12 KiB
    constexpr int EP_LD = 68, E_WAVE = 16 * EP_LD * 4;                                // 4352 B per compute wave
    constexpr int OFF_E = 2 * STAGE, OFF_SC = OFF_E + 4 * E_WAVE, OFF_BIAS = OFF_SC + 2 * GS_BM * 4;
    constexpr int A_PER = (GS_BM * BK / 4) / 128;                                     // float4 per A-feeder lane and chunk: 4 (two feeder waves)
    constexpr int B_PPW = (3 * 2 * (BN / 64)) / 2;                                    // DMA pieces per B-feeder wave and chunk: 6
    __shared__ __attribute__((aligned(16))) char smem[OFF_BIAS + 2 * BN * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = g.K / BK;
    // workgroup b: XCD x = b % 8, slot i = b / 8 on it; slots 2j and 2j+1 take the two halves of the same row tiles
    const int G2 = gridDim.x / 2, b = blockIdx.x;                                     // grid is even: G2 row-tile walkers per half
    const int xcd = b % 8, slot = b / 8, half = slot & 1;
    const int walker = (slot >> 1) * 8 + xcd;                                         // 0 .. G2-1
    const int n0 = half * BN;
    const int nb = g.n_tiles, q8 = nb / 8, r8 = nb % 8;
    auto logical = [&](int t) -> int { const int x = t % 8, i = t / 8; return (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i; };
    const int ntb = walker < nb ? (nb - walker + G2 - 1) / G2 : 0;                    // row tiles of this workgroup
    const int total = ntb * nchunks;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    float* scales = reinterpret_cast<float*>(smem + OFF_SC);                          // [2][128] by tile parity
    float* biasl = reinterpret_cast<float*>(smem + OFF_BIAS);                         // [2][128]
    if (total == 0) return;
    // de-phase the two workgroups of a CU (the dispatcher fills every CU of an XCD once before it doubles up: slots >= 32 are the
    // second residents -- a speed heuristic only): about half a tile of MFMA time
    if (g.dephase && slot >= 32 && ntb >= 3) {
        for (int i = 0; i < g.dephase; ++i) __builtin_amdgcn_s_sleep(127);
    }

    if (wave >= 6) {
        // ================= B feeder
        __builtin_amdgcn_s_setprio(2);
        const int fw = wave - 6;
        unsigned boff[B_PPW]; int bdst[B_PPW];
#pragma unroll
        for (int p = 0; p < B_PPW; ++p) {
            const int piece = fw * B_PPW + p, cb = piece % (BN / 64), po = piece / (BN / 64), oct = po % 2, plane = po / 2;
            boff[p] = (unsigned)(((int64_t)plane * g.N * g.K + ((int64_t)oct * g.N + n0 + cb * 64 + lane) * 8) * 2);
            bdst[p] = 3 * A_PLANE + plane * B_PLANE + oct * B_OCT + cb * 1024;
        }
        const int64_t b_chunk_bytes = (int64_t)BK * g.N * 2;
        auto issue_b = [&](int gc) {                                                  // global chunk gc -> stage gc & 1
            const int ti = gc / nchunks, c = gc - ti * nchunks;
            const int lt = logical(walker + ti * G2);
            const int set = g.tiles[lt * 3];
            const uint64_t base = (uint64_t)(uintptr_t)(g.Bt + (int64_t)set * g.bt_stride) + (uint64_t)(c * b_chunk_bytes);
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
            const uint64_t sbase = ((uint64_t)bhi << 32) | blo;
#pragma unroll
            for (int p = 0; p < B_PPW; ++p) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((gc & 1) * STAGE + bdst[p]));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(boff[p]), "s"(sbase), "s"(dst) : "memory");
            }
        };
        issue_b(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GS_BARRIER();
        for (int gc = 0; gc < total; ++gc) {
            if (gc + 1 < total) issue_b(gc + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GS_BARRIER();
        }
    } else if (wave >= 4) {
        // ================= A feeder
        __builtin_amdgcn_s_setprio(3);
        const int ft = tid - 256;                                                     // 0..127
        int rr[A_PER], adst[A_PER];
        const int c4 = (ft & 3) * 4;
#pragma unroll
        for (int p = 0; p < A_PER; ++p) { rr[p] = (ft + p * 128) >> 2; adst[p] = (c4 >> 3) * A_OCT + rr[p] * 16 + (c4 & 7) * 2; }
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v ra[PH_DA][A_PER];
        auto load_a = [&](int gc, int slot_) {
            const int ti = gc / nchunks, c = gc - ti * nchunks;
            const int lt = logical(walker + ti * G2);
            const int row0 = g.tiles[lt * 3 + 1], nrows = g.tiles[lt * 3 + 2];
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                const float* src = g.A + (int64_t)(row0 + min(rr[p], nrows - 1)) * g.lda + c4 + c * BK;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[slot_][p]) : "v"(src) : "memory");
            }
        };
        auto store_a = [&](int gc, int slot_) {
            char* As = smem + (gc & 1) * STAGE;
#pragma unroll
            for (int p = 0; p < A_PER; ++p) {
                uint2 h, m, l;
                const f4v v = ra[slot_][p];
                gs_split4(make_float4(v.x, v.y, v.z, v.w), h, m, l);
                *reinterpret_cast<uint2*>(As + adst[p]) = h;
                *reinterpret_cast<uint2*>(As + A_PLANE + adst[p]) = m;
                *reinterpret_cast<uint2*>(As + 2 * A_PLANE + adst[p]) = l;
            }
        };
        auto stage_tile_consts = [&](int ti) {
            const int lt = logical(walker + ti * G2);
            const int set = g.tiles[lt * 3], row0 = g.tiles[lt * 3 + 1], nrows = g.tiles[lt * 3 + 2];
            float sc = 1.f;
            if (g.row_scale) sc = g.row_scale[row0 + min(ft, nrows - 1)];
            float b0 = 0.f;
            if (g.bias) b0 = (g.bias + (int64_t)set * g.bias_stride)[n0 + ft];
            scales[(ti & 1) * GS_BM + ft] = sc;
            biasl[(ti & 1) * BN + ft] = b0;
        };
        static_assert(A_PER == 4 && PH_DA >= 2 && PH_DA <= 8, "wait macro is written for 4 loads per chunk");
#define PH_WAIT_SLOT(NEWER, SLOT) \
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[SLOT][0]), "+v"(ra[SLOT][1]), "+v"(ra[SLOT][2]), "+v"(ra[SLOT][3]) : "n"((NEWER) * A_PER) : "memory")
        stage_tile_consts(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < PH_DA; ++d) if (d < total) load_a(d, d);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[0][2]), "+v"(ra[0][3]) :: "memory");
        store_a(0, 0);
        GS_BARRIER();
        for (int g0 = 0; g0 < total; g0 += PH_DA) {
#pragma unroll
            for (int u = 0; u < PH_DA; ++u) {
                const int gc = g0 + u;
                if (gc < total) {
                    if (gc + 1 < total) {
                        const int newer = min(PH_DA - 2, total - 2 - gc);
                        const int SL = (u + 1) % PH_DA;
                        if (newer >= PH_DA - 2 && PH_DA >= 2) PH_WAIT_SLOT(PH_DA - 2, SL);
                        else if (newer == 1 && PH_DA > 3) PH_WAIT_SLOT(1, SL);
                        else if (newer == 2 && PH_DA > 4) PH_WAIT_SLOT(2, SL);
                        else PH_WAIT_SLOT(0, SL);
                        store_a(gc + 1, SL);
                        if ((gc + 1) % nchunks == 0) { stage_tile_consts((gc + 1) / nchunks); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    }
                    if (gc + PH_DA < total) load_a(gc + PH_DA, u);
                    GS_BARRIER();
                }
            }
        }
#undef PH_WAIT_SLOT
    } else {
        // ================= compute
        const int wr = wave / WC, wc = wave % WC, li = lane & 31, kh = lane >> 5;
        const int a_lane = kh * A_OCT + (wr * 64 + li) * 16, b_lane = 3 * A_PLANE + kh * B_OCT + (wc * 64 + li) * 16;
        float* E = reinterpret_cast<float*>(smem + OFF_E + wave * E_WAVE);
        const int er = lane >> 4, ec = (lane & 15) * 4;
        GS_BARRIER();
        int gc = 0;
        for (int ti = 0; ti < ntb; ++ti) {
            gm_f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int c = 0; c < nchunks; ++c, ++gc) {
                const char* S = smem + (gc & 1) * STAGE;
                gm_bf16x8 af[2][3], bf[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const gm_bf16x8*>(S + a_lane + p * A_PLANE + i * 512);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const gm_bf16x8*>(S + b_lane + p * B_PLANE + j * 512);
#define PH_PROD(PA, PB)                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
                PH_PROD(2, 0) PH_PROD(0, 2) PH_PROD(1, 1) PH_PROD(1, 0) PH_PROD(0, 1) PH_PROD(0, 0)
#undef PH_PROD
                GS_BARRIER();
            }
            // ---- epilogue of tile ti: wave-private staging in 16-row passes, stores only (no global load, no barrier)
            const int lt = logical(walker + ti * G2);
            const int row0 = g.tiles[lt * 3 + 1], nrows = g.tiles[lt * 3 + 2];
            const float* sc_t = scales + (ti & 1) * GS_BM;
            const int lcol = wc * 64 + ec, col = n0 + lcol;
            const float4 b4 = *reinterpret_cast<const float4*>(biasl + (ti & 1) * BN + lcol);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 8 * hh; e < 8 * hh + 8; ++e) E[((e & 3) + 8 * ((e >> 2) & 1) + 4 * kh) * EP_LD + j * 32 + li] = acc[i][j][e];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rl = wr * 64 + i * 32 + hh * 16 + it * 4 + er;
                        if (rl >= nrows) continue;
                        const int64_t row = row0 + rl;
                        const float sc = sc_t[rl];
                        float4 v = *reinterpret_cast<const float4*>(&E[(it * 4 + er) * EP_LD + ec]);
                        v.x = v.x * sc + b4.x; v.y = v.y * sc + b4.y; v.z = v.z * sc + b4.z; v.w = v.w * sc + b4.w;
                        if (g.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
                        if (g.relu_bits) g.relu_bits[(row * g.ldc + col) >> 2] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
                        if (g.nt_store) {
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            f4v vv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(g.C + row * g.ldc + col));
                        } else {
                            *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
    }
}
