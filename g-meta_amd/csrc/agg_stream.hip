// Stream aggregate: the batched segmented sum of agg.hip (DGL's update_all(copy_src, sum), learner.py:38-39,44-45, with GraphConv's
// normalisations, learner.py:29-32,49) as a REGISTER-LIGHT gather engine on LDS-DMA.
//
// Why a second kernel.  The window kernel (k_agg_win) hides memory latency with occupancy: ~36 VGPRs, 32 waves per CU, four 1-KiB row loads
// in flight per wave.  The persistent split GEMM of the other stream owns 16 waves x 120 VGPRs = 480 of the 512 registers of every SIMD lane:
// while it runs, no aggregate wave fits beside it, the two queues of a meta-step time-slice the chip and the HBM-bound aggregate never
// overlaps the matrix-bound update.  Here a wave keeps its gathers in flight in LDS instead of in registers:
//   * every gathered source row is ONE `global_load_lds_dwordx4` (1 KiB per instruction at width 256) into a wave-private ring of R
//     slots; the wave's in-order VM counter orders a read behind its gather (`s_waitcnt vmcnt(R - 1)` before slot i is read: the R - 1
//     younger gathers stay in flight) and an lgkmcnt(0) orders the gather that refills the slot behind that read (as_issue): no
//     destination registers, no barrier, R KiB in flight per wave;
//   * edge descriptors ride the same queue: the per-edge tables (sources for the issuing side, weights -- R edges behind -- for the consuming
//     side, laid out in row order) arrive 64 edges at a time by `global_load_lds_dword` into two small per-wave rings, a chunk ahead of
//     their use, and are read back four at a time with broadcast ds_read_b128 -- they never sit in asm-loaded registers the compiler could
//     copy before the data lands; row bounds come through the scalar cache (16 rows per round trip, parked in the lanes of one VGPR).  The
//     vector memory queue carries nothing but DMA and the output stores -- nothing the compiler would wait `vmcnt(0)` for;
//   * a wave owns a contiguous run of rows (cost-balanced segments built with the batch, XCD-contiguous like the window kernel's blocks),
//     accumulates a row in edge order with the same fma chain as k_agg_win (bitwise the same rows) and writes it once;
//   * hub rows (in-degree above the batch's hub threshold, up to ~1000 edges) hold no edges in the row-ordered part of the tables (their
//     row bound carries a flag); their edges follow behind it and are streamed in parts of ~128 edges by the launch's FIRST workgroups, one
//     wave per part, all parts of a row on one XCD: partial rows through write-through stores, an arrival ticket, and the last arriver sums
//     the parts in part order (deterministic) -- the scheme of agg.hip's hub blocks on the stream engine.
// <= 32 VGPRs and R + 1 KiB of LDS per wave: one 4-wave workgroup (52 KiB) fits next to a persistent GEMM workgroup (101 KiB, 480 VGPRs of 512), several
// fit an otherwise empty CU.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include "gm_internal.h"

typedef int as_i4 __attribute__((ext_vector_type(4)));

#define AS_WAVES 4                      // waves per workgroup: one per SIMD

// ---- scalar-cache loads.  SMEM returns out of order and the compiler does not count asm loads: every use sits behind an explicit lgkmcnt(0).
// Addresses are formed in full by scalar arithmetic (no register offset operand).
__device__ __forceinline__ int as_sload(const void* p) {            // one dword, waited for
    int r;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");       // early clobber: the result never aliases the address pair
    return r;
}
__device__ __forceinline__ void as_sload16(const void* p, as_i4& a, as_i4& b, as_i4& c, as_i4& d) {      // 16 dwords at p (4-byte aligned), issued only
    asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\ts_load_dwordx4 %3, %4, 0x30"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p) : "memory");
}
__device__ __forceinline__ void as_swait16(as_i4& a, as_i4& b, as_i4& c, as_i4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) :: "memory");
}
// lane LANE (a constant) of v <- val (wave-uniform); ignores EXEC
#define AS_WRITELANE(V, VAL, LANE) asm volatile("v_writelane_b32 %0, %1, " #LANE : "+v"(V) : "s"(VAL))

struct AggS {
    const int32_t* sptr;                // [rows + 1 (+ pad)] row bounds in the stream edge tables; bit 31 of entry r + 1: row r is a hub row (not written by its segment)
    const int32_t* su; const float* sw; // stream edge tables: source row of x / weight per edge -- the rows' edges in row order, then the hub rows' edges in hub order
    const float* x; unsigned row_bytes; // gathered matrix, bytes per row (ldx * 4; rows * row_bytes < 2 GiB)
    float* out; int width; int nt;      // [rows, width]
    const int32_t* rowlist;             // optional: output row of stream row i (tables over a compact row list; not used by the shipped dispatch)
    const int2* seg; int n_seg;         // row segments [x, y) of the non-hub workgroups' waves
    int hub_wgs;                        // blocks [0, hub_wgs): hub workgroups (a multiple of the XCD count)
    const int32_t* hsu; const float* hsw;                                    // the tables holding the hub rows' edges (the full launch's own; a list launch borrows the batch's)
    const int32_t* heavy; const int32_t* hcum; int n_heavy; int e_norm;      // hub rows, prefix of their edge counts; their edges start at hsu / hsw[e_norm + hcum[h]]
    const int32_t* hub; float* hub_scratch; int hub_part, hub_ld, n_parts;   // gm_agg_schedule's part table (NULL: one part per hub row), partial rows
    const int32_t* xord;                // with hub: [GM_NXCD + 1 offsets | n_parts part ids] -- the parts of a hub row all run on ONE XCD (see the hub section of the kernel)
    int prio;                           // s_setprio of the waves (GM_AGG_STREAM_PRIO): beside a GEMM workgroup whose feeder waves run at 2 / 3
    unsigned long long* dbg;            // timeline probe (tools/coreside_probe.py): [2 * blocks] start / end of every workgroup on the device's constant clock
};

// One gather: LDS[slot .. slot + LPR * 16) <- x[voff .. ), 16 bytes per lane of the lower LPR lanes (M0 = the slot's LDS byte address).  M0 is the
// compiler's: saved and restored.  So is EXEC: the narrow variants mask the upper lanes for the one DMA instruction and put back the mask they found
// (saved in an SGPR pair), so the statement is correct inside a predicated region too -- the kernel's control flow is wave-uniform today (act is applied at
// the stores, never around a gather), but nothing in this statement depends on that any more.  (The kernel is built xnack-: gfx950 default target, no
// replay of a scalar load whose destination overlaps its address.)
// The statement starts with lgkmcnt(0): the slot was READ (ds_read_b128) by the step that issues this gather, and nothing orders a DMA's LDS write behind
// a read that has been issued but not yet executed.  The first version had no such wait: with the other stream's GEMM saturating the CU's LDS the read
// could sit in its queue longer than a cache-resident source row takes to arrive, and the sum picked up 64-byte pieces of the row gathered for edge e + R
// -- in hub rows (whose sources are the hottest), ~1 launch in 500, only beside the GEMM (tools/agg_stream_race.py; profiles/r05_experiments_not_shipped.txt
// E).  Cost: the read's latency on every step's issue, ~3 % of a launch (1.14 M rows: 460 -> 476 us).  A ring of R + 1 slots (refill the slot read a step
// earlier: the wait is then free) was measured too: 56 KiB per workgroup are two workgroups per CU instead of three, 516 us.
template <int LPR>
__device__ __forceinline__ void as_issue(unsigned lds_slot, const float* xbase, unsigned voff) {
    unsigned keep;
    if constexpr (LPR == 64)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(xbase), "s"(lds_slot) : "memory");
    else {
        unsigned long long ex;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_and_b32 exec_lo, exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\ts_nop 1\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(xbase), "s"(lds_slot), "s"(LPR == 32 ? 0xffffffffu : 0xffffu) : "memory", "scc");
    }
}

// 64 descriptors (256 bytes): LDS[dst .. dst + 256) <- base[voff .. ), 4 bytes per lane, all 64 lanes
__device__ __forceinline__ void as_issue_desc(unsigned lds_dst, const void* base, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <int LPR, int R>
__global__ __launch_bounds__(AS_WAVES * 64) void k_agg_stream(AggS a) {
    static_assert(R % 4 == 0 && R >= 8 && R <= 60, "ring depth: whole descriptor blocks, within the 6-bit VM counter");
    constexpr int SLOT = LPR * 16;                                  // bytes per ring slot = one row
    constexpr int RB = R / 4;                                       // the consuming side runs RB descriptor blocks behind the issuing side
    constexpr int WAVE_LDS = R * SLOT + 1024;                       // the gather ring + two 128-entry descriptor rings (sources, weights)
    // DYNAMIC shared memory on purpose: with a static size hipcc derives an occupancy bound from it (3-4 workgroups per CU) and then pads the kernel's
    // register allocation up to what that occupancy leaves (amdhsa_next_free_vgpr 129 instead of 32) -- which is exactly what must not happen to a
    // kernel meant to fit into the 32 registers a persistent GEMM workgroup leaves per SIMD lane
    extern __shared__ __attribute__((aligned(16))) char ring_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool act = lane < LPR;                                    // narrow rows: the upper lanes idle (no early return: all control flow stays wave-uniform)
    char* ring = ring_all + wave * WAVE_LDS;
    const unsigned ring_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring);
    const int* su_l = reinterpret_cast<const int*>(ring + R * SLOT);          // sources of edges e: entry e & 127
    const int* sw_l = su_l + 128;                                             // weights, same indexing
    const unsigned lane4 = (unsigned)lane * 4u;
    const unsigned lane16 = act ? (unsigned)lane * 16u : 0u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- row walk of the consuming side (row segments only): rc = current row, p1 = its end bound (flagged); the bounds of 16 rows at a time
    // are parked in lanes (row & 31) of vp1 (a scalar-cache round trip every 16 rows; not prefetched: 16 more live SGPRs cost more than they hide)
    int vp1 = 0, rc = 0, p1 = 0, r_end = 0;
    auto win_load = [&](int r16) {
        as_i4 q0, q1, q2, q3;
        as_sload16(a.sptr + r16 + 1, q0, q1, q2, q3);
        as_swait16(q0, q1, q2, q3);
        if (r16 & 16) {
            AS_WRITELANE(vp1, q0[0], 16); AS_WRITELANE(vp1, q0[1], 17); AS_WRITELANE(vp1, q0[2], 18); AS_WRITELANE(vp1, q0[3], 19);
            AS_WRITELANE(vp1, q1[0], 20); AS_WRITELANE(vp1, q1[1], 21); AS_WRITELANE(vp1, q1[2], 22); AS_WRITELANE(vp1, q1[3], 23);
            AS_WRITELANE(vp1, q2[0], 24); AS_WRITELANE(vp1, q2[1], 25); AS_WRITELANE(vp1, q2[2], 26); AS_WRITELANE(vp1, q2[3], 27);
            AS_WRITELANE(vp1, q3[0], 28); AS_WRITELANE(vp1, q3[1], 29); AS_WRITELANE(vp1, q3[2], 30); AS_WRITELANE(vp1, q3[3], 31);
        } else {
            AS_WRITELANE(vp1, q0[0], 0); AS_WRITELANE(vp1, q0[1], 1); AS_WRITELANE(vp1, q0[2], 2); AS_WRITELANE(vp1, q0[3], 3);
            AS_WRITELANE(vp1, q1[0], 4); AS_WRITELANE(vp1, q1[1], 5); AS_WRITELANE(vp1, q1[2], 6); AS_WRITELANE(vp1, q1[3], 7);
            AS_WRITELANE(vp1, q2[0], 8); AS_WRITELANE(vp1, q2[1], 9); AS_WRITELANE(vp1, q2[2], 10); AS_WRITELANE(vp1, q2[3], 11);
            AS_WRITELANE(vp1, q3[0], 12); AS_WRITELANE(vp1, q3[1], 13); AS_WRITELANE(vp1, q3[2], 14); AS_WRITELANE(vp1, q3[3], 15);
        }
    };
    auto store_row = [&](int64_t orow) {
        if (!act) return;
        float4* dst = reinterpret_cast<float4*>(a.out + orow * a.width + lane * 4);
        if (a.nt) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v vv = {acc.x, acc.y, acc.z, acc.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(dst));
        } else *dst = acc;
    };
    auto row_done = [&]() {                                         // write row rc (unless a hub row), move to the next
        if (p1 >= 0) store_row(a.rowlist ? (int64_t)as_sload(a.rowlist + rc) : (int64_t)rc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ++rc;
        if ((rc & 15) == 0 && rc < r_end) win_load(rc);
        p1 = __builtin_amdgcn_readlane(vp1, rc & 31);
    };

    // ---- the gather pipeline over edges [e_lo, e_hi) of the stream tables; ROWS: rows end inside the run (a row segment), else everything is one sum (a hub part)
    auto run = [&](const int32_t* tsu, const float* tsw, const int e_lo, const int e_hi, auto rows_tag) {
        constexpr bool ROWS = decltype(rows_tag)::value;
        if (e_hi <= e_lo) return;
        int slot = 0;                                               // byte offset of the ring slot of step (kb, J): edge e and edge e - R share slot ((e - 4 kb0) % R)
        auto consume = [&](int ec, float w, bool steady) {
            if constexpr (ROWS) { while (ec == (p1 & 0x7fffffff)) row_done(); }      // rows that end before this edge (empty rows included)
            if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4 v = *reinterpret_cast<const float4*>(ring + slot + lane16);
            acc.x = __fmaf_rn(v.x, w, acc.x); acc.y = __fmaf_rn(v.y, w, acc.y); acc.z = __fmaf_rn(v.z, w, acc.z); acc.w = __fmaf_rn(v.w, w, acc.w);
        };
        auto issue = [&](int u) { as_issue<LPR>(ring_lds + (unsigned)slot, a.x, (unsigned)u * a.row_bytes + lane16); };      // (waits for the slot's read: as_issue)
        auto next_slot = [&]() { slot = slot + SLOT == R * SLOT ? 0 : slot + SLOT; };
        // descriptor chunk c (64 edges) -> half (c & 1) of the wave's source / weight rings
        auto load_su = [&](int c) { as_issue_desc(ring_lds + (unsigned)(R * SLOT + (c & 1) * 256), tsu, (unsigned)c * 256u + lane4); };
        auto load_sw = [&](int c) { as_issue_desc(ring_lds + (unsigned)(R * SLOT + 512 + (c & 1) * 256), tsw, (unsigned)c * 256u + lane4); };
        const int kb0 = e_lo >> 2, kb1 = (e_hi - 1) >> 2;           // blocks of 4 edges; the loop runs RB blocks past the last one to drain the ring
        const int c0 = e_lo >> 6;
        load_su(c0); load_su(c0 + 1); load_sw(c0); load_sw(c0 + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int kb = kb0; kb <= kb1 + RB; ++kb) {
            const int e0 = kb * 4;
            // the next chunks, a chunk ahead of their first reader: sources when the issuing side enters a chunk, weights when the consuming side (R edges
            // behind) does -- each lands in the half its side has just left; >= 64 - R later gathers are waited for before either is read
            if ((e0 & 63) == 0 && e0 > e_lo) load_su((e0 >> 6) + 1);
            if ((e0 & 63) == R && e0 - R > e_lo) load_sw(((e0 - R) >> 6) + 1);
            // four sources / four weights, broadcast reads; moved to SGPRs at once (wave-uniform values: eight VGPRs less across the block's steps --
            // the kernel has to stay within 32 VGPRs to fit beside a persistent GEMM workgroup)
            const int4 u4 = *reinterpret_cast<const int4*>(su_l + (e0 & 127));
            const int4 w4 = *reinterpret_cast<const int4*>(sw_l + ((e0 - R) & 127));
            const int u0 = __builtin_amdgcn_readfirstlane(u4.x), u1 = __builtin_amdgcn_readfirstlane(u4.y), u2 = __builtin_amdgcn_readfirstlane(u4.z), u3 = __builtin_amdgcn_readfirstlane(u4.w);
            const float w0 = __int_as_float(__builtin_amdgcn_readfirstlane(w4.x)), w1 = __int_as_float(__builtin_amdgcn_readfirstlane(w4.y));
            const float w2 = __int_as_float(__builtin_amdgcn_readfirstlane(w4.z)), w3 = __int_as_float(__builtin_amdgcn_readfirstlane(w4.w));
            if (e0 - R >= e_lo && e0 + 3 < e_hi) {                  // every edge of the issued and of the consumed block is inside the run
                consume(e0 - R, w0, true); issue(u0); next_slot();
                consume(e0 + 1 - R, w1, true); issue(u1); next_slot();
                consume(e0 + 2 - R, w2, true); issue(u2); next_slot();
                consume(e0 + 3 - R, w3, true); issue(u3); next_slot();
            } else {
#define AS_STEP(J, U, W) { const int e = e0 + J;                                                        \
                     if (e - R >= e_lo && e - R < e_hi) consume(e - R, W, e <= e_hi);                   \
                     if (e >= e_lo && e < e_hi) issue(U);                                                \
                     next_slot(); }
                AS_STEP(0, u0, w0) AS_STEP(1, u1, w1) AS_STEP(2, u2, w2) AS_STEP(3, u3, w3)
#undef AS_STEP
            }
        }
    };

    const int b = blockIdx.x;
    if (a.prio > 0) { if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
    if (a.dbg && threadIdx.x == 0) a.dbg[2 * b] = wall_clock64();
    struct Stamp { unsigned long long* p; __device__ ~Stamp() { if (p) *p = wall_clock64(); } } stamp_end{(a.dbg && threadIdx.x == 0) ? a.dbg + 2 * b + 1 : nullptr};
    if (b >= a.hub_wgs) {
        // ================= row segments (XCD-contiguous order: hardware block b runs on XCD b % 8; hub_wgs is a multiple of 8)
        const int bn = b - a.hub_wgs, nwg = gridDim.x - a.hub_wgs;
        const int sidx = ((bn % GM_NXCD) * (nwg / GM_NXCD) + bn / GM_NXCD) * AS_WAVES + wave;
        if (sidx >= a.n_seg) return;
        rc = as_sload(&a.seg[sidx].x); r_end = as_sload(&a.seg[sidx].y);
        if (rc >= r_end) return;
        const int e_lo = as_sload(a.sptr + rc) & 0x7fffffff, e_hi = as_sload(a.sptr + r_end) & 0x7fffffff;
        win_load(rc & ~15);
        p1 = __builtin_amdgcn_readlane(vp1, rc & 31);
        run(a.su, a.sw, e_lo, e_hi, std::true_type{});
        while (rc < r_end) row_done();                              // the last row with edges and the empty rows behind it
        return;
    }
    // ================= hub parts.  Unsplit hub rows: wave hw of the hub workgroups takes rows hw, hw + HW, ...  Split rows: ALL parts of a hub row run on waves
    // of one XCD (hardware block b runs on XCD b % 8; the host dealt the hub rows to the XCDs by part count: a.xord), like the window kernel's schedule: the
    // partial rows then travel through ONE L2 -- the ticket protocol below (write-through stores acknowledged by the writer's L2, a relaxed ticket, one
    // agent-scope acquire) has only ever been exercised that way, and the last arriver's reads hit that L2.
    const int HW = a.hub_wgs * AS_WAVES, hw = b * AS_WAVES + wave;
    int i0 = hw, i1 = a.n_parts, istep = HW;
    if (a.xord) {
        const int xcd = b % GM_NXCD;
        i0 = as_sload(a.xord + xcd) + (b / GM_NXCD) * AS_WAVES + wave; i1 = as_sload(a.xord + xcd + 1); istep = (a.hub_wgs / GM_NXCD) * AS_WAVES;
    }
    for (int i = i0; i < i1; i += istep) {
        const int g = a.xord ? as_sload(a.xord + GM_NXCD + 1 + i) : i;
        int h = g, p = 0, P = 1;
        if (a.hub) { h = as_sload(a.hub + a.n_heavy + 1 + g); const int o0 = as_sload(a.hub + h); p = g - o0; P = as_sload(a.hub + h + 1) - o0; }
        const int row = as_sload(a.heavy + h);
        int eb = a.e_norm + as_sload(a.hcum + h), ee = a.e_norm + as_sload(a.hcum + h + 1);
        if (P > 1) { eb += p * a.hub_part; if (p < P - 1) ee = eb + a.hub_part; }      // the last part takes the remainder (up to 1.5 parts)
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        run(a.hsu, a.hsw, eb, ee, std::false_type{});
        if (P == 1) { store_row((int64_t)row); continue; }
        // partial row -> scratch with write-through (sc1) stores, drained; one relaxed agent-scope ticket; the last arriver does ONE agent-scope acquire
        // and sums the P partial rows in part order (writers and reader share an L2: see above)
        typedef float f4v __attribute__((ext_vector_type(4)));
        if (act) {
            float* dst = a.hub_scratch + (int64_t)g * a.hub_ld + lane * 4;
            const f4v val = {acc.x, acc.y, acc.z, acc.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(val) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* ctr = const_cast<int*>(a.hub) + a.n_heavy + 1 + a.n_parts + h;
        int last = 0;
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == P - 1) {
                __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                last = 1;
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (!last) continue;
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            const float* sc = a.hub_scratch + (int64_t)(g - p) * a.hub_ld + lane * 4;
#pragma unroll 1
            for (int k = 0; k < P; ++k) { const float4 t = *reinterpret_cast<const float4*>(sc + (int64_t)k * a.hub_ld); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }      // (one at a time: 32 VGPRs)
        }
        store_row((int64_t)row);
    }
}

// ---------------------------------------------------------------------------------------------------------------- tables (built with the batch)
// Row bounds of the stream edge tables: the batch's CSR bounds minus the edges of the hub rows before each row (hub rows keep no edges in the
// row-ordered part; bit 31 of a hub row's END bound flags it).  hubs: ascending hub rows, cum[k] = edges of the hub rows before hub k (cum[n] = all).
__global__ void k_stream_bounds(const int32_t* indptr, int64_t rows, const int32_t* hubs, const int32_t* cum, int n_hubs, int32_t* sptr, int pad) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;         // entry r (r = rows: the end of the last row)
    if (r > rows + pad) return;
    if (r > rows) { sptr[r] = 0; return; }
    int lo = 0, hi = n_hubs;                                                  // hubs before row r
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hubs[mid] < r) lo = mid + 1; else hi = mid; }
    const bool prev_is_hub = r > 0 && lo > 0 && hubs[lo - 1] == r - 1;
    sptr[r] = (indptr[r] - cum[lo]) | (prev_is_hub ? (int)0x80000000 : 0);
}
// sources / weights of every edge at its stream position, one thread per EDGE: the stream order is the CSR order with the hub rows' edges taken out and
// appended (hub by hub) behind the rest, so an edge's position follows from the hub rows that start at or before it -- a binary search over the hub rows'
// edge ranges (hub_lo / hub_hi: a few thousand entries, cache-resident).  Coalesced reads and writes: 26-35 us for the 2.4 M edges of a query batch, where a
// thread per ROW copying its edges one after the other (a dependent chain per thread) took 190-380 us -- twice the extraction kernels' own time per batch.
__global__ void k_stream_hub_ranges(const int32_t* indptr, const int32_t* hubs, int n_hubs, int32_t* hub_lo, int32_t* hub_hi) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_hubs) return;
    const int r = hubs[k];
    hub_lo[k] = indptr[r]; hub_hi[k] = indptr[r + 1];
}
__global__ void k_stream_edges(int64_t edges, const int32_t* hub_lo, const int32_t* hub_hi, const int32_t* cum, int n_hubs, int e_norm, const int32_t* src,
                               const int32_t* src2, const float* wgt, int32_t* su, int32_t* su2, float* sw) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= edges) return;
    int lo = 0, hi = n_hubs;                                                  // hub rows whose first edge is at or before e
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (hub_lo[mid] <= e) lo = mid + 1; else hi = mid; }
    int64_t pos;
    if (lo > 0 && e < hub_hi[lo - 1]) pos = (int64_t)e_norm + cum[lo - 1] + (e - hub_lo[lo - 1]);      // an edge of hub lo - 1
    else pos = e - cum[lo];                                                   // lo hub rows lie entirely before e
    su[pos] = src[e];
    if (su2) su2[pos] = src2[e];
    sw[pos] = wgt ? wgt[e] : 1.f;
}
// Wave segments: segment k starts at the first row whose cost prefix (stream edges + rows before it) reaches k / n_seg of the total
__global__ void k_stream_segs(const int32_t* sptr, int64_t rows, int n_seg, int2* seg) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const int64_t total = (int64_t)(sptr[rows] & 0x7fffffff) + rows;
    auto first_row = [&](int kk) -> int {
        if (kk >= n_seg) return (int)rows;
        const int64_t target = total * kk / n_seg;
        int64_t lo = 0, hi = rows;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)(sptr[mid] & 0x7fffffff) + mid < target) lo = mid + 1; else hi = mid; }
        return (int)lo;
    };
    seg[k] = make_int2(first_row(k), first_row(k + 1));
}

// Workgroups of a stream launch over a table of `cost` (edges + rows): a multiple of the XCD count.  Waves get SHORT runs (~GM_AGG_STREAM_COST cost units each, about
// 30 rows and their edges): the workgroups resident on an XCD at one time then sweep a few dozen consecutive subgraphs, and the second gather of a source row -- by
// another row of its subgraph -- still finds it in that XCD's L2 (1.14 M-row launch at width 256: 534 us with 768 long-lived workgroups, 437 us with 6,144;
// window kernel 494-518).  Too short and a wave's start-up (descriptor chunk + first round trip) dominates (142 k-row launch: best at ~1,500 workgroups).
int gm_stream_wgs(int64_t cost) {
    const int per_cu = gm_knob().agg_stream_wgs;
    if (per_cu > 0) return std::max(2 * GM_NXCD, gm_num_cus() * per_cu / GM_NXCD * GM_NXCD);
    const int64_t per_wave = std::max(16, gm_knob().agg_stream_cost);
    const int64_t wgs = (cost / per_wave + AS_WAVES - 1) / AS_WAVES;
    return (int)std::min<int64_t>(65536, std::max<int64_t>(gm_num_cus() * 3 / GM_NXCD * GM_NXCD, (wgs + GM_NXCD - 1) / GM_NXCD * GM_NXCD));
}

// Does a batch get stream tables at all?  Where the stream kernel pays (measured): sparse induced subgraphs (the arxiv shape: ~2
// in-edges per row) in batches large enough to fill its pipelines.
bool gm_stream_batch_ok(const gm_batch* b, int o) {
    return b->rows > 0 && b->edges > 0 && b->d_enorm[o] && b->edges <= 8 * b->rows && b->rows >= gm_knob().agg_stream_min_rows;
}
// Stream tables of one orientation (o = 0: by destination; 1: by source).  hubs_host / deg_host: the orientation's ascending hub rows and
// their degrees (host copies from the finalisation's round trip); n_parts: hub parts of the orientation's part table (the hub count when the
// rows are not split).  Needs the batch's per-edge tables (d_enorm, d_efeat).
int gm_stream_tables(gm_batch* b, int o, const int32_t* hubs_host, const int32_t* deg_host, int n_hubs, int n_parts, const std::vector<int32_t>* part_tab, hipStream_t s,
                     gm_stager* sg) {
    if (!gm_stream_batch_ok(b, o)) return GM_OK;      // dense batches (Tissue shape, ~24 in-edges per row: 5.6 vs 3.2 ms per meta-step) and small ones (FirstMM shape) keep the window kernel
    const int32_t* indptr = o ? b->d_indptr_t : b->d_indptr;
    std::vector<int32_t> cum(n_hubs + 1, 0);
    for (int k = 0; k < n_hubs; ++k) cum[k + 1] = cum[k] + deg_host[k];
    int32_t* d_hubs = nullptr;
    GM_TRY(gm_balloc(b, &b->d_scum[o], cum.size(), s)); GM_TRY(sg->upload(b->d_scum[o], cum));
    GM_TRY(gm_balloc(b, &d_hubs, (size_t)std::max(n_hubs, 1), s));
    if (n_hubs > 0) GM_TRY(sg->upload((void*)d_hubs, (const void*)hubs_host, sizeof(int32_t) * (size_t)n_hubs));
    const int pad = 48;                                             // the 16-row bound windows read up to 31 entries past the end
    const int e_norm = (int)(b->edges - cum[n_hubs]);
    b->stream_enorm[o] = e_norm;
    GM_TRY(gm_balloc(b, &b->d_sptr[o], (size_t)b->rows + 1 + pad, s));
    hipLaunchKernelGGL(k_stream_bounds, dim3((unsigned)((b->rows + 1 + pad + 255) / 256)), dim3(256), 0, s, indptr, (int64_t)b->rows, d_hubs, b->d_scum[o], n_hubs, b->d_sptr[o], pad);
    const size_t n_ed = (size_t)b->edges + 192;                     // (+ the tail of the last 64-edge descriptor chunks)
    GM_TRY(gm_balloc(b, &b->d_su[o], n_ed, s)); GM_TRY(gm_balloc(b, &b->d_sw[o], n_ed, s));
    const bool feat = o == 0 && b->d_efeat;                         // layer 1 gathers rows of the store's feature table
    if (feat) GM_TRY(gm_balloc(b, &b->d_su_feat, n_ed, s));
    int32_t* d_hlo = nullptr; int32_t* d_hhi = nullptr;
    GM_TRY(gm_balloc(b, &d_hlo, (size_t)std::max(n_hubs, 1), s)); GM_TRY(gm_balloc(b, &d_hhi, (size_t)std::max(n_hubs, 1), s));
    if (n_hubs > 0) hipLaunchKernelGGL(k_stream_hub_ranges, dim3((n_hubs + 255) / 256), dim3(256), 0, s, indptr, d_hubs, n_hubs, d_hlo, d_hhi);
    hipLaunchKernelGGL(k_stream_edges, dim3((unsigned)((b->edges + 255) / 256)), dim3(256), 0, s, (int64_t)b->edges, d_hlo, d_hhi, b->d_scum[o], n_hubs, e_norm,
                       o ? b->d_indices_t : b->d_indices, feat ? b->d_efeat : nullptr, b->d_enorm[o], b->d_su[o], feat ? b->d_su_feat : nullptr, b->d_sw[o]);
    // workgroups: the hub parts' share of the launch by their share of its work (in whole XCD rounds), the rest for the row segments
    const int nwg = gm_stream_wgs(b->edges + b->rows);
    int hub_wgs = 0;
    if (n_parts > 0) {
        const double share = (double)cum[n_hubs] / (double)(b->edges + b->rows);
        hub_wgs = std::max(GM_NXCD, (int)(nwg * share / GM_NXCD + 0.5) * GM_NXCD);
        hub_wgs = std::min(hub_wgs, std::min(nwg / 2 / GM_NXCD * GM_NXCD, (n_parts + AS_WAVES * GM_NXCD - 1) / (AS_WAVES * GM_NXCD) * GM_NXCD));
    }
    b->stream_hubwg[o] = hub_wgs; b->stream_nwg[o] = nwg; b->stream_nparts[o] = n_parts;
    if (part_tab && n_parts > 0) {
        // split hub rows: deal the rows to the XCDs (each to the one with the fewest parts so far), list every XCD's parts -- interleaved over its rows, so that
        // the waves of an XCD start on different rows -- behind the GM_NXCD + 1 list offsets
        const std::vector<int32_t>& tab = *part_tab;                  // [part_off: n_hubs + 1][...]
        std::vector<std::vector<int>> rows_of(GM_NXCD);
        std::vector<int> load(GM_NXCD, 0);
        for (int h = 0; h < n_hubs; ++h) {
            const int x = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            rows_of[x].push_back(h); load[x] += tab[h + 1] - tab[h];
        }
        std::vector<int32_t> xord(GM_NXCD + 1, 0);
        for (int x = 0; x < GM_NXCD; ++x) {
            xord[x] = (int32_t)xord.size() - (GM_NXCD + 1);
            for (int k = 0, more = 1; more; ++k) {
                more = 0;
                for (int h : rows_of[x]) if (tab[h] + k < tab[h + 1]) { xord.push_back(tab[h] + k); more = 1; }
            }
        }
        xord[GM_NXCD] = (int32_t)xord.size() - (GM_NXCD + 1);
        GM_REQUIRE(xord[GM_NXCD] == n_parts, GM_EINVAL, "stream tables: %d hub parts listed, %d expected", xord[GM_NXCD], n_parts);
        GM_TRY(gm_balloc(b, &b->d_sxord[o], xord.size(), s)); GM_TRY(sg->upload(b->d_sxord[o], xord));
    }
    b->stream_nseg[o] = (nwg - hub_wgs) * AS_WAVES;
    GM_TRY(gm_balloc(b, &b->d_sseg[o], (size_t)b->stream_nseg[o], s));
    hipLaunchKernelGGL(k_stream_segs, dim3((b->stream_nseg[o] + 255) / 256), dim3(256), 0, s, b->d_sptr[o], (int64_t)b->rows, b->stream_nseg[o], b->d_sseg[o]);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

#ifdef GM_PROBES
// Probe build only (build.py --probes -> libgmeta_hip_probes.so; include/gmeta_hip_probes.h): workgroup timelines of the stream launches and the
// s_setprio experiment.  The product library exports neither the symbol nor the process-global buffer.
static unsigned long long* g_stream_dbg = nullptr; static int g_stream_dbg_n = 0;
// enable != 0: every later stream launch stamps its workgroups' start / end clocks (2 x uint64 each) into a device buffer of n entries; out != NULL: copy
// the buffer to the host (after the caller has synchronised)
extern "C" int gm_stream_debug(int32_t enable, unsigned long long* out, int32_t n) {
    if (enable && !g_stream_dbg) { GM_HIP(hipMalloc((void**)&g_stream_dbg, sizeof(unsigned long long) * (size_t)n)); g_stream_dbg_n = n; GM_HIP(hipMemset(g_stream_dbg, 0, sizeof(unsigned long long) * (size_t)n)); }
    if (out && g_stream_dbg) GM_HIP(hipMemcpy(out, g_stream_dbg, sizeof(unsigned long long) * (size_t)std::min(n, g_stream_dbg_n), hipMemcpyDeviceToHost));
    if (!enable && g_stream_dbg) { (void)hipFree(g_stream_dbg); g_stream_dbg = nullptr; g_stream_dbg_n = 0; }
    return GM_OK;
}
#endif

template <int LPR, int R>
static void launch_stream(const AggS& a, int nwg, hipStream_t s) {
    hipLaunchKernelGGL((k_agg_stream<LPR, R>), dim3(nwg), dim3(AS_WAVES * 64), AS_WAVES * (R * LPR * 16 + 1024), s, a);      // (<= 64 KiB: no attribute needed)
}

// The stream launch of a full aggregate over the batch the tables belong to; false: not eligible (the caller takes the window kernel)
bool gm_stream_ok(const gm_agg_args& g) {
    if (!g.stream || g.s_out || g.bias || g.mask_h || g.mask_b || g.relu || g.relu_bits) return false;
    if (g.stream_feat && !gm_knob().agg_stream_gather) return false;
    if (g.rowlist || g.skip_on) return false;                                  // partial launches stay on the window kernel (measured: profiles/r05_experiments_not_shipped.txt C.1)
    // The stream kernel takes every per-edge quantity from the batch's stream tables: the launch described by `g` must BE that aggregate -- same
    // orientation, rows, per-edge weights and (layer 1) per-edge feature rows -- or the window kernel, which reads g's own arrays, computes it
    const gm_batch* b = g.stream; const int o = g.stream_o;
    if (o < 0 || o > 1 || !b->d_sptr[o] || g.rows != b->rows || g.indptr != (o ? b->d_indptr_t : b->d_indptr) || g.e_w != b->d_enorm[o]) return false;
    if (g.stream_feat ? (o != 0 || !b->d_su_feat || g.x_idx != b->d_efeat) : (g.x_idx != nullptr || g.x_row != nullptr)) return false;
    return
           (g.width == 64 || g.width == 128 || g.width == 256) && g.ldx % 4 == 0 && (((uintptr_t)g.x | (uintptr_t)g.out) & 15) == 0 &&
           (uint64_t)g.stream_xrows * (uint64_t)g.ldx * 4u < ((uint64_t)1 << 31) && g.stream_xrows < (1 << 24);
}
int gm_launch_stream(const gm_agg_args& g, int nt, hipStream_t s) {
    const gm_batch* b = g.stream; const int o = g.stream_o;
    const bool split = b->d_hub[o] != nullptr;
    const int32_t* su = g.stream_feat ? b->d_su_feat : b->d_su[o];
    AggS a{b->d_sptr[o], su, b->d_sw[o], g.x, (unsigned)(g.ldx * 4), g.out, g.width, nt, nullptr, b->d_sseg[o], b->stream_nseg[o],
           b->stream_hubwg[o], su, b->d_sw[o], b->d_heavy[o], b->d_scum[o], b->n_heavy[o], b->stream_enorm[o], split ? g.hub : nullptr, split ? g.hub_scratch : nullptr,
           split ? b->hub_part[o] : 0, GM_AGG_HUB_LD, b->stream_nparts[o], split ? b->d_sxord[o] : nullptr, 0, nullptr};
    GM_REQUIRE(!split || a.xord, GM_EINVAL, "stream aggregate: split hub rows without their XCD lists");
#ifdef GM_PROBES
    { static const int pr = getenv("GM_AGG_STREAM_PRIO") ? atoi(getenv("GM_AGG_STREAM_PRIO")) : 0; a.prio = pr; }
    if (g_stream_dbg && 2 * b->stream_nwg[o] <= g_stream_dbg_n) a.dbg = g_stream_dbg;
#endif
    const int depth = gm_knob().agg_stream_depth, nwg = b->stream_nwg[o];
    // (ring depths: 8 / 12 KiB of gathers in flight per wave; beyond ~15 KiB per wave the workgroup's LDS would pass 64 KiB -- the reach of M0's 16-bit DMA base)
    if (g.width == 256) { if (depth == 8) launch_stream<64, 8>(a, nwg, s); else launch_stream<64, 12>(a, nwg, s); }
    else if (g.width == 128) { if (depth == 8) launch_stream<32, 16>(a, nwg, s); else launch_stream<32, 24>(a, nwg, s); }
    else { if (depth == 8) launch_stream<16, 32>(a, nwg, s); else launch_stream<16, 48>(a, nwg, s); }
    GM_HIP(hipGetLastError());
    return GM_OK;
}
