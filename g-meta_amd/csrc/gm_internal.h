// Internal declarations shared by the HIP translation units of libgmeta_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "gm_bound.h"
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../include/gmeta_hip.h"

#define GM_WAVE 64
#define GM_NXCD 8

void gm_set_error(const char* fmt, ...);

#define GM_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            gm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return GM_EHIP;                                                                       \
        }                                                                                         \
    } while (0)
#define GM_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            gm_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)
#define GM_TRY(call)            \
    do {                        \
        int r__ = (call);       \
        if (r__ != GM_OK) return r__; \
    } while (0)

struct gm_store {
    int32_t n_graphs = 0, feat_dim = 0;
    int32_t feat_ld = 0;             // row stride of d_feat: feat_dim padded with zero columns (32, else a multiple of 64) so that layer 1 runs on the
                                     // vectorised aggregate / DMA GEMM / fast weight-gradient kernels whatever the dataset's feature width
    int64_t total_nodes = 0, total_edges = 0, max_nodes = 0;
    std::vector<int64_t> node_off, edge_off;      // host prefix sums per graph
    // device
    int64_t* d_node_off = nullptr;   // [G+1]
    int64_t* d_in_ptr = nullptr;     // [total_nodes + 1] global edge offsets, rows = global node ids
    int32_t* d_in_idx = nullptr;     // [total_edges]     source node, LOCAL to its graph
    int64_t* d_out_ptr = nullptr;    // by-source CSR of the same edges (for the transposed induce)
    int32_t* d_out_idx = nullptr;    // destination node, LOCAL to its graph
    bool symmetric = false;          // out-CSR == in-CSR element for element (extract.hip: adjacency lists walked once)
    float* d_feat = nullptr;         // [total_nodes, feat_ld]
    unsigned* d_feat_amax = nullptr; // [1] bit pattern of max |feature| over the whole table
    // per graph (host): max |x| and mean |x| over the non-zero entries.  The two-piece fp16 kernels (opt-in, gm_bound.h) bound the layer-1 operand
    // of a task by the largest feature of the graphs the task draws from, and keep the three-piece kernels for a pass whose table is LOOSE: an
    // entry 2^r below its bound keeps min(22, 39 - r) bits, so a table whose largest entry sits more than 2^14 above its typical one (or is
    // not finite / above 2^50) would lose precision silently
    std::vector<float> h_feat_amax, h_feat_mean;
};

// Receptive-field tables (cone.hip): level l = rows whose layer-l activation reaches a centre.
struct gm_cone_level {
    int32_t n = 0, nnz = 0;            // rows of this level; edges from level l-1 into it
    int32_t* d_row = nullptr;          // [n]  batch row (ascending; level L: centre order)
    float* d_norm = nullptr;           // [n]  norm[row]
    int32_t* d_feat_row = nullptr;     // level 0 only: [n] feature row in the store
    int32_t* d_set_off = nullptr;      // [sets+1] compact row range of every set
    std::vector<int32_t> h_set_off;
    int32_t* d_tiles = nullptr; int32_t n_tiles = 0;                 // GEMM row tiles over the compact rows
    int32_t* d_chunks = nullptr; int32_t* d_set_chunk_off = nullptr; int32_t n_chunks = 0;
    int32_t* d_indptr = nullptr; int32_t* d_indices = nullptr;       // [n+1], [nnz]: in-edges, sources = compact ids of level l-1
    int32_t* d_indptr_t = nullptr; int32_t* d_indices_t = nullptr;   // [n_{l-1}+1], [nnz]: the same edges by source, destinations = compact ids of level l
    int32_t* d_heavy[2] = {nullptr, nullptr}; int32_t n_heavy[2] = {0, 0};
};
struct gm_cone {
    int L = 0; bool ok = false; int heavy_deg = 64;
    gm_cone_level lv[GM_MAX_GCN + 1];
    void* slab = nullptr;           // every device array of the levels is carved from this one region of the BATCH's slabs (sized from upper bounds: cone.hip); released with the batch
};
struct gm_batch;
int gm_batch_cone(const gm_batch* b, int L, hipStream_t s, const gm_cone** out);   // built on first use, cached in the batch
void gm_cone_free(gm_cone* c, hipStream_t s);

struct gm_batch {
    const gm_store* store = nullptr;
    int64_t rows = 0, edges = 0;
    int32_t subs = 0, sets = 0, centres = 1;
    // host mirrors (small)
    std::vector<int32_t> h_sub_off, h_set_sub_off, h_set_row_off, h_graph;
    // device
    int32_t* d_sub_off = nullptr;      // [subs+1]
    int32_t* d_set_sub_off = nullptr;  // [sets+1]
    int32_t* d_set_row_off = nullptr;  // [sets+1]
    int32_t* d_graph = nullptr;        // [subs]
    int32_t* d_parent = nullptr;       // [rows]
    int32_t* d_feat_row = nullptr;     // [rows]
    int32_t* d_indptr = nullptr;       // [rows+1]
    int32_t* d_indices = nullptr;      // [edges]
    int32_t* d_indptr_t = nullptr;     // [rows+1]
    int32_t* d_indices_t = nullptr;    // [edges]
    int32_t* d_centre = nullptr;       // [subs*centres] local index
    std::vector<int32_t> h_centre;     // host copy, brought by the finalisation's round trip (gm_batch_read serves it without another one)
    float* d_norm = nullptr;           // [rows]
    // derived launch tables (built by gm_batch_finalize)
    int32_t* d_sub_set = nullptr;      // [subs]  set of each subgraph
    int32_t* d_tiles = nullptr;        // [n_tiles*3]  GEMM row tiles: set, row0, nrows (<= GM_GEMM_BM)
    int32_t n_tiles = 0;
    int32_t* d_chunks = nullptr;       // [n_chunks*3] weight-gradient row chunks: set, row0, nrows
    int32_t n_chunks = 0;
    int32_t* d_set_chunk_off = nullptr;// [sets+1]
    int32_t* d_heavy[2] = {nullptr, nullptr};   // rows with more than gm_heavy_deg() edges, by-destination / by-source CSR
    int32_t n_heavy[2] = {0, 0};
    int32_t heavy_deg = 64;
    // block schedule of the window aggregate (one launch: hub-row blocks are interleaved with the window blocks of their
    // own subgraph on the XCD that owns them, see gm_agg_schedule): [8 * sched_len] entries per orientation
    int32_t* d_sched[2] = {nullptr, nullptr};
    int32_t sched_len[2] = {0, 0};
    int32_t sched_win = 0;
    // hub rows split into parts of hub_part edges, one block each (gm_agg_schedule): part table + arrival counters, and the partial
    // rows.  One aggregate launch per orientation at a time (a batch's launches are stream-ordered: they share the layer buffers too).
    int32_t* d_hub[2] = {nullptr, nullptr}; float* d_hub_scratch[2] = {nullptr, nullptr}; int32_t hub_part[2] = {0, 0};
    int32_t hub_words[2] = {0, 0}, hub_parts[2] = {0, 0};      // size of d_hub[o] (int32 words) and its part count (d_hub_scratch[o]: parts x GM_AGG_HUB_LD floats)
    // a SECOND set of arrival counters / partial rows (gm_batch_hub_alt): lets two streams run aggregate launches over the batch at the same
    // time (gm_meta_step's two query streams); the part tables are copies, only the counters and the partial rows are private
    mutable int32_t* d_hub2[2] = {nullptr, nullptr}; mutable float* d_hub_scratch2[2] = {nullptr, nullptr};
    // The arrival counters / partial rows belong to ONE launch at a time.  Launches of a batch are stream-ordered in gm_meta_step; a caller
    // that moves an orientation's launches to another stream (public gm_aggregate / gm_gcn_* API) is ordered behind the previous stream's
    // launch by gm_batch_hub_order (an event wait, only when the stream changes).  Host-side bookkeeping, guarded by hub_mu.
    mutable hipStream_t hub_stream[4] = {nullptr, nullptr, nullptr, nullptr}; mutable bool hub_used[4] = {false, false, false, false}; mutable hipEvent_t hub_ev[4] = {nullptr, nullptr, nullptr, nullptr};      // [set * 2 + orientation]
    // per-edge tables (gm_batch_finalize): the source's norm for both CSR orientations (enorm[o][e] = norm[indices_o[e]]) and the source's
    // feature row (efeat[e] = feat_row[indices[e]]): what the aggregate would otherwise fetch with a dependent 4-byte gather per edge
    float* d_enorm[2] = {nullptr, nullptr}; int32_t* d_efeat = nullptr;
    // row gains of the two aggregates (gm_bound.h): |out| <= gain * max |in| -- [0] forward: max_i sum_{u->i} norm[u]; [1] transposed with the
    // destination's norm on the output (the backward of an aggregate-first layer): max_i norm[i] * outdeg(i)
    float* d_gain = nullptr;
    // fused aggregate + GEMM (forward passes nobody differentiates): per row {u0, u1, bits(w0), bits(w1)} -- the row's one or two sources
    // (batch rows in d_fuse2, feature rows in d_fuse2_feat) and their norms; rows without a source carry {GM_FUSE_ZERO, same, 1, 0}, rows of any other degree {row | GM_FUSE_SELF, same, 1, 0}:
    // their aggregate is written by the ordinary kernel (gm_agg_args::skip_lo/hi) and picked up as is
    void* d_fuse2 = nullptr; void* d_fuse2_feat = nullptr;
    int64_t unfused_rows = 0, unfused_edges = 0;     // rows (and their in-edges) the ordinary aggregate still writes in a fused pass
    // The partial aggregate launch of a fused pass walks a COMPACT list of its window rows (in-degree GM_FUSE_MAXDEG+1 .. heavy_deg, ascending)
    // instead of every row of the batch: 8 % of the rows of the arxiv query batch carry work in that launch, and a wave whose 16-row window holds
    // one of them runs its gathers one dependent batch at a time; with the list every wave window is full of rows that have work.
    int32_t* d_mid = nullptr; int32_t n_mid = 0, mid_win = 0;
    // stream aggregate (agg_stream.hip), per orientation: row bounds in the stream edge tables (hub rows hold no edges in the row-ordered part; bit 31 of a
    // hub row's end bound flags it), the per-edge source / weight tables in row order with the hub rows' edges behind (d_su_feat: sources = feature rows
    // of the store, layer 1), the prefix of the hub rows' edge counts, the cost-balanced row segments of the launch's waves and its grid split
    int32_t* d_sptr[2] = {nullptr, nullptr}; int32_t* d_su[2] = {nullptr, nullptr}; int32_t* d_su_feat = nullptr; float* d_sw[2] = {nullptr, nullptr};
    int32_t* d_scum[2] = {nullptr, nullptr}; int2* d_sseg[2] = {nullptr, nullptr}; int32_t* d_sxord[2] = {nullptr, nullptr};
    int32_t stream_nseg[2] = {0, 0}, stream_nwg[2] = {0, 0}, stream_hubwg[2] = {0, 0}, stream_nparts[2] = {0, 0}, stream_enorm[2] = {0, 0};
    // The stream tables are built at the FIRST launch that can take the stream kernel (gm_agg_stream_args; round 6), from what the finalisation's round trip
    // brought to the host: with the fused passes of the dense schedule only a large support batch ever gets there, and never the by-source orientation --
    // built with every batch they were 0.23 ms of host time and eight kernels of each 32-task query batch for nothing.
    struct stream_pending { bool pending = false; bool has_tab = false; int n_parts = 0; std::vector<int32_t> hubs, deg, tab; };
    mutable stream_pending spend[2];
    int32_t* d_sched_mid = nullptr; int32_t sched_len_mid = 0;      // block schedule over the list (hub parts placed by the hub row's approximate list position)
    mutable int64_t unfused_src = -1;                // DISTINCT source rows of those in-edges (profiling only: counted on first use, gm_batch_unfused_sources)
    // compact row lists for the row-sparse backward (gm_hparams_t.sparse_bwd)
    int32_t n_c = 0;                   // centre rows: subs * centres
    int32_t* d_crow = nullptr;         // [n_c]  batch row of every centre
    float* d_cnorm = nullptr;          // [n_c]  norm[centre row]
    float* d_norm_c = nullptr;         // [rows] norm with the SIGN BIT SET on every row that is not a centre: row scale + "somebody reads this row" flag of the last layer's
                                       //        forward-only update (gm_gemm_args::row_scale_keep)
    int32_t n_e1 = 0;                  // in-edges of centres, concatenated in centre order
    int32_t* d_e1_row = nullptr;       // [n_e1] source row of the edge
    int32_t* d_e1_par = nullptr;       // [n_e1] compact index of the centre it enters
    float* d_e1_norm = nullptr;        // [n_e1] norm[source row]
    int32_t* d_c_tiles = nullptr; int32_t n_c_tiles = 0;          // GEMM tiles over centre rows (per set)
    int32_t* d_c_chunks = nullptr; int32_t* d_c_set_chunk_off = nullptr; int32_t n_c_chunks = 0;
    int32_t* d_e1_chunks = nullptr; int32_t* d_e1_set_chunk_off = nullptr; int32_t n_e1_chunks = 0;
    mutable gm_cone* cone[GM_MAX_GCN + 1] = {};     // receptive-field tables per number of GCN layers (gm_hparams_t.cone)
    hipStream_t stream = nullptr;      // stream the arrays were produced on (and are freed on, stream-ordered)
    // The ~45 device arrays above are carved out of a few slabs (gm_balloc): a stream-ordered allocation or free costs the host 20-35 us, and a
    // meta-batch builds and drops two batches per step (3 ms of frees in Subgraphs.get_batch before this)
    struct slab { char* base; size_t cap, used; };
    std::vector<slab> slabs;
    std::mutex slab_mu;
    mutable hipEvent_t used_ev = nullptr;   // last consumer on ANOTHER stream: the frees wait for it (gm_batch_mark_use)
};
struct gm_stager;
int gm_batch_finalize(gm_batch* b, hipStream_t s, gm_stager& sg);
int gm_batch_gains(const gm_batch* b, hipStream_t s);
// n elements of T from the batch's slabs (256-byte aligned; lives until the batch is destroyed)
int gm_balloc_bytes(gm_batch* b, void** p, size_t bytes, hipStream_t s);
template <class T> static inline int gm_balloc(gm_batch* b, T** p, size_t n, hipStream_t s) { return gm_balloc_bytes(b, (void**)p, (n ? n : 1) * sizeof(T), s); }     // d_gain at first use (two-piece kernels only)
// A consumer that ran kernels over the batch on `st` calls this afterwards: gm_batch_destroy then orders its frees behind
// that work instead of relying on the host having synchronised (deferred read-back, prefetch threads).
void gm_batch_mark_use(const gm_batch* b, hipStream_t st);

// ---- tuning / debug knobs: every GM_* environment variable is read ONCE, under std::call_once, into this
// struct (the prefetch thread and the training thread both enter the library); per-device quantities are derived at the call site.
struct gm_knobs {
    int agg_min_waves, agg_min_win, agg_sched, agg_hub_part, agg_unr, agg_nt, agg_variant, agg_edge_tables, heavy_deg;
    int extract_global_bitmap, feat_pad, timing;
    int extract_pref16;            // GM_EXTRACT_PREF16: 16-bit per-word prefix counts in the extraction kernels' LDS (1, default): four resident workgroups per CU instead of three at the arxiv parent size
    int gemm_mode;                 // 0 exact fp32, 1 split-bf16, -1 not set (library default)
    int gemm_split_min_tiles;      // -1: a quarter of the current device's CUs
    int gemm_split_grid;           // 0: the current device's CU count
    int centre_store;              // GM_CENTRE_STORE: the last layer's update stores only the centre rows of its activation -- in every pass (2, default), in the forward-only passes (1) -- or every row (0)
    int gemm_fused_rounds, gemm_plain_rounds, gemm_half_tiles, gemm_bn, gemm_mid_tiles, gemm_glds, gemm_nt, gemm_small, wgrad_split, dz_glds;
    int fuse_agg, head_stage, side_stream_priority;
    int fuse_diff;                 // GM_FUSE_DIFF: the differentiated passes of the dense schedule take the fused aggregate + GEMM too, their weight gradients form Z's rows from the source table: 2 (default) everywhere except a support batch whose full launches take the stream aggregate, 1 everywhere, 0 never
    int agg_mid_win;               // GM_AGG_MID_WIN: rows per wave window over that list (0 = by its length)
    int agg_mid_list;              // GM_AGG_MID_LIST: the partial aggregate launch of a fused pass walks a compact list of its window rows (1, default) or every row (0)
    int query_streams;             // GM_QUERY_STREAMS: 2 = the query evaluations of gm_meta_step alternate between two streams; anything else (default 0) = one stream
    int head_threads;              // GM_HEAD_THREADS: workgroup size of k_head_loss: 256 or 512; anything else (default 0) = 1024
    int split16_min_rows;          // GM_SPLIT16_MIN_ROWS: support + query rows from which gm_meta_step takes the two-piece kernels (smaller steps are launch-bound: no gain)
    int split_pieces;              // GM_SPLIT_PIECES: pieces per operand of the split kernels inside gm_meta_step: 3 = exact bf16 triple, all 24 operand bits (default); 2 = opt-in fp16 pair under recorded bounds
    int wgrad_round_bias;          // weight-gradient chunking: percent of row-slot efficiency another round of chunks must gain over fewer, longer chunks
    int wgrad_split_min_chunks;    // GM_WGRAD_SPLIT_MIN_CHUNKS: smallest launch (row chunks) that takes the split weight-gradient kernel; -1: a quarter of the CUs
    int cu_mask_support;           // CUs per XCD reserved for the support chain's stream (0: no CU masks)
    int agg_stream;                // GM_AGG_STREAM: eligible full aggregate launches take the LDS-DMA stream kernel (agg_stream.hip)
    int agg_stream_wgs;            // GM_AGG_STREAM_WGS: its workgroups per CU (0 = by the batch's size: agg_stream_cost)
    int agg_stream_cost;           // GM_AGG_STREAM_COST: edges + rows per wave of a stream launch
    int agg_stream_gather;         // GM_AGG_STREAM_GATHER: the layer-1 launches (sources = rows of the store's feature table) take the stream kernel too
    int agg_stream_depth;          // GM_AGG_STREAM_DEPTH: KiB of gathers in flight per wave (8 / 12)
    int agg_stream_min_rows;       // GM_AGG_STREAM_MIN_ROWS: smallest batch (rows) that builds stream tables; dense batches (more than 8 edges per row) never do
};
const gm_knobs& gm_knob();

// ---- host phase timing for the setup paths (env GM_TIMING=1 prints to stderr)
#include <chrono>
struct gm_phase_timer {
    const char* what; bool on; std::chrono::steady_clock::time_point t0, t;
    explicit gm_phase_timer(const char* w) : what(w) {
        on = gm_knob().timing != 0; t0 = t = std::chrono::steady_clock::now();
    }
    void lap(const char* phase) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[gm timing] %s/%s %.1f us\n", what, phase, std::chrono::duration<double, std::micro>(n - t).count());
        t = n;
    }
    ~gm_phase_timer() {
        if (on) fprintf(stderr, "[gm timing] %s total %.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---- device allocation helpers (stream-ordered pool so that per-batch builds do not sync the device)
int gm_dev_alloc(void** p, size_t bytes, hipStream_t s);
void gm_dev_free(void* p, hipStream_t s);
// big, batch-lived blocks: a process-level cache in front of the stream-ordered pool (common.hip: why); *cap >= need
int gm_slab_acquire(char** base, size_t* cap, size_t need, hipStream_t s);
void gm_slab_release(char* base, size_t cap, hipStream_t s);
template <class T>
static inline int gm_alloc(T** p, size_t n, hipStream_t s) { return gm_dev_alloc((void**)p, (n ? n : 1) * sizeof(T), s); }

// ---- pinned staging for batch builds.  A hipMemcpyAsync from / to PAGEABLE host memory waits for the stream on the host (and the callers used
// to add a hipStreamSynchronize so that their std::vectors could go out of scope): every small table of a batch build was a host round trip,
// and under a saturated GPU (extraction prefetched while a meta-step runs) each round trip waits for the build's kernel to get a CU -- ~20 of them
// made one extraction as long as the meta-step it was meant to hide behind.  A gm_stager hands out pinned host memory from a per-thread pool;
// copies through it are truly asynchronous and the memory returns to the pool once the stream has passed the stager's end.
struct gm_stager {
    hipStream_t s;
    std::vector<int> used;           // pool chunks this build holds
    char* cur = nullptr; size_t left = 0;
    explicit gm_stager(hipStream_t st) : s(st) {}
    ~gm_stager();
    gm_stager(const gm_stager&) = delete;
    gm_stager& operator=(const gm_stager&) = delete;
    void* take(size_t bytes);                                        // pinned, 64-byte aligned; NULL when the host allocation fails
    int upload(void* dptr, const void* src, size_t bytes);           // src (any host memory) -> staging -> device, asynchronous
    template <class T> int upload(T* dptr, const std::vector<T>& v) { return v.empty() ? GM_OK : upload((void*)dptr, (const void*)v.data(), v.size() * sizeof(T)); }
    template <class T> T* download(const T* dsrc, size_t n) {        // device -> pinned (valid after the caller synchronises `s`); NULL on failure
        T* h = (T*)take((n ? n : 1) * sizeof(T));
        if (h && n && hipMemcpyAsync(h, dsrc, n * sizeof(T), hipMemcpyDeviceToHost, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return h;
    }
};

// ---- layout helpers
struct gm_layout {
    int n_gcn;
    int dims[GM_MAX_GCN + 1];
    int n_out, link, hc;              // hc = width of the head input
    int64_t w_off[GM_MAX_GCN], b_off[GM_MAX_GCN], wl_off, bl_off, P;
};
int gm_make_layout(const gm_model_t* m, gm_layout* L);
static inline int gm_pad_feat(int F) { return F <= 32 ? 32 : (F + 63) / 64 * 64; }

// ---- kernels launched across translation units
// Generic CSR aggregate: out[r,:] = epi( s_out[r] * sum_{c in row r} s_in[c] * x[src(c),:] ).
struct gm_agg_args {
    const int32_t* indptr;
    const int32_t* indices;
    const float* x;            // [*, ldx]
    const int32_t* x_row;      // optional indirection applied to the column index (feature gather)
    const float* e_w;          // optional per-edge source scale (= s_in[indices[e]]; gm_batch::d_enorm): replaces the s_in gather
    const int32_t* x_idx;      // optional per-edge source row of x (= x_row[indices[e]]; gm_batch::d_efeat): replaces the x_row gather
    int skip_on, skip_lo, skip_hi;   // window kernel only, with skip_on: leave rows with skip_lo <= degree <= skip_hi unwritten (the fused aggregate + GEMM forms them)
    int64_t ldx;
    const float* s_in;         // optional per-source scale
    const float* s_out;        // optional per-destination scale
    const float* mask_h;       // optional [rows, width]: zero the output where mask_h <= 0 (relu')
    const uint8_t* mask_b;     // same mask, packed: byte (row*width + col)/4 holds the relu' bits of 4 consecutive columns
    uint8_t* relu_bits;        // optional output: packed relu' bits of `out` (written with relu; width % 4 == 0)
    const float* bias;         // optional per-set bias (bias + set*bias_stride) -- matmul-first layers
    int64_t bias_stride;
    const int32_t* set_row_off;// required when bias != NULL and bias_stride != 0: row range of every set
    int n_sets;
    int relu;
    float* out;                // [rows, width]
    int64_t rows;
    int width;
    const int32_t* heavy;      // optional list of rows with more than GM_HEAVY_DEG edges (processed by a whole workgroup)
    int n_heavy;
    int heavy_deg;
    const int32_t* sched;      // optional block schedule (gm_agg_schedule): hub rows ride in the window launch
    int sched_len, sched_win;
    const int32_t* hub; float* hub_scratch; int hub_part;     // with sched: hub rows split over several blocks (gm_agg_sched)
    const int32_t* rowlist; int64_t n_list; int list_win;      // window kernel only: the wave windows walk rowlist[0 .. n_list) instead of every row (sched / sched_len then index list blocks)
    // optional: the batch / orientation whose stream tables this launch may use (gm_agg_stream_args): eligible launches take the LDS-DMA stream kernel (agg_stream.hip)
    const gm_batch* stream; int stream_o; int stream_feat; int64_t stream_xrows;
};
// fills the stream fields of `a` for orientation o of batch b (gather: the sources are rows of the store's feature table); a no-op without tables
int gm_agg_stream_args(gm_agg_args& a, const gm_batch* b, int o, bool gather, hipStream_t s);      // (builds the orientation's tables at first use, on s)
bool gm_stream_ok(const gm_agg_args& g);
bool gm_stream_batch_ok(const gm_batch* b, int o);      // the batch is of the kind that gets stream tables (built at first use)
int gm_launch_stream(const gm_agg_args& g, int nt, hipStream_t s);
int gm_stream_tables(gm_batch* b, int o, const int32_t* hubs_host, const int32_t* deg_host, int n_hubs, int n_parts, const std::vector<int32_t>* part_tab, hipStream_t s,
                     gm_stager* sg);
#define GM_FUSE_SELF 0x40000000   // gm_batch::d_fuse2 entry: the source is the row's own, already aggregated, row
#define GM_FUSE_ZERO 0x20000000   // ... the row has no source: an all-zero row
const float* gm_zero_row(hipStream_t s);   // 4096 zero floats on the current device (allocated once)
#define GM_FUSE_MAXDEG 2
#define GM_AGG_HUB_LD 512      // floats per partial hub row (the widest window-kernel width)
struct gm_agg_sched { int32_t* d_sched = nullptr; int32_t len = 0; int32_t* d_hub = nullptr; float* d_hub_scratch = nullptr; int32_t hub_part = 0; int32_t hub_words = 0, parts = 0;
                      std::vector<int32_t> tab; };      // tab: the host copy of the part table (gm_agg_schedule_flat builds further schedules over the same parts)
int gm_batch_hub_order(const gm_batch* b, int o, hipStream_t s, int set = 0);
int gm_batch_hub_alt(const gm_batch* b, hipStream_t s);      // allocates the second counter / partial-row set (once)
template <class A> inline int gm_agg_hub(A& a, const gm_batch* b, int o, hipStream_t s, int set = 0) {
    const bool alt = set == 1 && b->d_hub2[o];
    a.hub = alt ? b->d_hub2[o] : b->d_hub[o]; a.hub_scratch = alt ? b->d_hub_scratch2[o] : b->d_hub_scratch[o]; a.hub_part = b->hub_part[o];
    return a.hub ? gm_batch_hub_order(b, o, s, alt ? 1 : 0) : GM_OK;
}
// Rows per wave window for a launch over `rows` rows (64 at most, halved until the launch has enough waves).
int gm_agg_window(int64_t rows, int64_t edges);
// Block schedule for the window aggregate over `rows` rows with the given (ascending, host) hub-row list: 8 per-XCD lists of
// equal length out->len; entry >= 0: window block id, <= -2: hub part -(entry) - 2 (hub row heavy[..] itself when the rows are not
// split: out->d_hub == NULL), -1: nothing.  A hub row's blocks follow the window block that contains the row, on the XCD whose L2 is
// streaming that subgraph.  heavy_deg_host: the rows' edge counts (NULL: never split).
struct gm_stager;
// Only the flat block schedule, for hub rows at (virtual, ascending) positions pos[] among `rows` window rows, over the part table `tab` of an earlier
// gm_agg_schedule call (empty: one block per hub row)
int gm_agg_schedule_flat(gm_batch* b, int64_t rows, int win, const int32_t* pos, int n_heavy, const std::vector<int32_t>& tab, int32_t** d_sched, int32_t* len, hipStream_t s, gm_stager* sg);
int gm_agg_schedule(gm_batch* b, int64_t rows, int win, const int32_t* heavy_host, const int32_t* heavy_deg_host, int n_heavy, gm_agg_sched* out, hipStream_t s, gm_stager* sg);
int gm_heavy_deg();   // rows with more edges than this are aggregated by a whole workgroup (env GM_HEAVY_DEG; default by density, see gm_heavy_deg_for)
int gm_heavy_deg_for(int64_t rows, int64_t edges);
int gm_launch_aggregate(const gm_agg_args& a, hipStream_t s);

// Grouped GEMM  C[rows of set t] = epi( A[rows] @ op(B_t) ),  A [rows,K] (lda), C [rows,N] (ldc).
struct gm_gemm_args {
    const float* A; int64_t lda;
    const float* B; int64_t b_stride;   // B_t = B + t*b_stride ; [K,N] row-major, or [N,K] when transB
    int transB;
    float* C; int64_t ldc;
    int K, N;
    const float* row_scale;             // optional: C *= row_scale[row] (before bias)
    const float* row_scale_keep;        // optional, split kernels only (others ignore it): |value| = row_scale, sign bit set = nobody reads this row of C: it is computed and
                                        //   not stored (the h[to_fetch] select of Classifier.forward, learner.py, fused into the last layer's epilogue); n_keep = rows stored (accounting)
    int64_t n_keep;
    const float* bias; int64_t bias_stride;   // optional per-set bias [N]
    int relu;
    const float* mask_h;                // optional [rows, ldc]: zero C where mask_h <= 0 (relu')
    const uint8_t* mask_b;              // same mask packed, byte (row*ldc + col)/4 = bits of 4 consecutive columns (needs vector stores)
    uint8_t* relu_bits;                 // optional output: packed relu' bits of C (with relu; N % 4 == 0, ldc == N)
    const uint16_t* Bsplit; int64_t bsplit_stride;   // optional: B_t as three bf16 planes (gm_split_weights) -> the split-bf16 kernel (gemm_split.h)
    // with Bsplit: fused aggregate + GEMM.  fuse2 = gm_batch::d_fuse2 / d_fuse2_feat; A / lda then address the aggregate's INPUT rows
    // (previous layer's output or the feature table) and rows the table flags GM_FUSE_SELF read their finished aggregate from zside
    const void* fuse2; const float* zside; int64_t ldz;
    // split kernel with two fp16 pieces per operand (np = 2; 0 / 3 = three bf16 pieces): Bsplit then holds fp16 planes made under b_bound, and
    // a_bound bounds the A rows (gm_bound.h).  amax_out (np = 2 only): per-set slots that receive the largest |C| stored (atomicMax on zeroed slots)
    int np; gm_bound a_bound, b_bound; unsigned* amax_out;
    float* zero_out;                    // also zero-fill this [rows, ldc] buffer (dQ of the backward pass that follows): in the epilogue of the split / DMA kernels, a memset on the other paths
    const int32_t* tiles;               // device [n_tiles*3]: set, row0, nrows  (nrows <= BM)
    int n_tiles;
    int64_t rows;                       // total rows covered by the tiles (profiling: flops = 2*rows*K*N)
};
#define GM_GEMM_BM 128
int gm_launch_gemm_nn(const gm_gemm_args& a, hipStream_t s);
// fp32 GEMM on the bf16 matrix cores by exact 3-way operand splitting (gemm_split.h): eligibility of a launch and the weight planes.
// gm_gemm_mode(): 0 = exact-fp32 MFMA kernels only, 1 = split-bf16 kernel where eligible (env GM_GEMM_MODE=f32|split, gm_set_gemm_mode).
int gm_gemm_mode();
bool gm_gemm_split_ok(int n_tiles, int K, int N);
// np = 3: three bf16 planes; np = 2: two fp16 planes under `bound` (per-set stride 3 planes either way)
int gm_split_weights(const float* params, int64_t pstride, int64_t w_off, int K, int N, int trans, int sets, uint16_t* out, hipStream_t s, int np = 3,
                     gm_bound bound = gm_no_bound());
// out[t * out_stride] = max(out[...], max_i |x[t * stride + off + i]|), i < n, as fp32 bit patterns (the slots must have been zeroed)
int gm_amax(const float* x, int64_t stride, int64_t off, int64_t n, int sets, unsigned* out, int64_t out_stride, hipStream_t s);
int gm_amax_segs(const float* x, const int64_t* off, const int64_t* n, int segs, unsigned* out, int64_t out_stride, hipStream_t s);   // up to 8 segments of x, one launch
int gm_split_np();                      // pieces per operand of the split kernels: 3 (bf16 triple, default) or 2 (opt-in fp16 pair, only where bounds are recorded); env GM_SPLIT_PIECES

// Grouped transposed-A GEMM for weight gradients:
//   dW_t[K,N] = sum_{rows of set t} a_scale[row] * A[row,:]^T  G[row,:]   and   db_t[N] = sum G[row,:]
// computed as per-chunk partials followed by a deterministic reduction that also applies `beta`:
//   out_t = (beta_is_sgd ? w_t - lr * sum : sum)
// Reductions held back by gm_launch_wgrad (hold_this) until a later call on the same batch flushes them together with its own
// (k_wgrad_reduce_n): opaque argument blocks, filled and consumed by gemm.hip.
struct gm_wgrad_hold { int n = 0; int sets[GM_MAX_GCN] = {}; alignas(16) unsigned char slot[GM_MAX_GCN][256] = {}; };
struct gm_wgrad_args {
    const float* A; int64_t lda; int K;
    const int32_t* a_row;               // optional row indirection for A (layer-1 feature gather)
    const float* G; int64_t ldg; int N;
    const float* Gb; int64_t ldgb;      // optional: matrix whose column sums give db (default: G)
    const float* a_scale;               // optional
    const int32_t* chunks;              // device [n_chunks*3]: set, row0, nrows
    int n_chunks;
    const int32_t* set_chunk_off;       // device [sets+1]: chunk range per set
    int sets;
    float* partial;                     // workspace [n_chunks, (K+1)*N]  (row K = bias partial)
    float* dW; int64_t dw_stride;       // dW_t = dW + t*dw_stride   [K,N]
    float* db; int64_t db_stride;       // db_t                      [N]
    // optional fused inner-loop SGD (meta.py:126,151): next_t[off + j] = cur_t[off + j] - lr * grad, written together with the gradient
    const float* sgd_cur; int64_t sgd_cur_stride; float* sgd_next; int64_t sgd_next_stride; float sgd_lr;
    int64_t w_off, b_off;               // offsets of this layer's W and b inside a parameter vector
    // two fp16 pieces per operand (np = 2) in the split weight-gradient kernel: bounds of A and G.  With planes: pl_np = 2 writes fp16 planes
    // under pl_bound
    int np; gm_bound a_bound, g_bound; int pl_np; gm_bound pl_bound;
    uint16_t* pl_fwd; uint16_t* pl_dz;  // optional (with sgd_next): the updated W also written as split-bf16 planes for the NEXT step's GEMMs
                                        // ([set][3][K/8][N][8] for X @ W, [set][3][N/8][K][8] for dQ @ W^T; gemm_split.h layouts): no k_split_w launches
    float* wt_next;                     // optional (with sgd_next): the updated W also written transposed, [set][N][K] -- what the NEXT
                                        // step's dZ GEMM (dQ @ W^T on the row-major DMA kernel) reads, instead of a transpose launch
    int64_t rows;                       // total rows covered by the chunks (profiling: flops = 2*rows*K*N)
    // optional (split kernel only: gm_wgrad_gather_ok): A = Z of a pass whose forward ran the FUSED aggregate + GEMM -- rows of one or two sources are formed
    // from gx (row stride ldgx) through the per-row table fuse2 (gm_batch::d_fuse2 / d_fuse2_feat) in the aggregate's fma order, rows the table flags
    // GM_FUSE_SELF are read from A (the partial aggregate launch wrote them), GM_FUSE_ZERO rows are zero
    const void* fuse2; const float* gx; int64_t ldgx;
    gm_wgrad_hold* hold; int hold_this; // optional: hold this call's reduction back (hold_this = 1; its `partial` must then stay untouched) /
                                        // flush the held ones with this call's reduction (hold_this = 0)
};
#define GM_WGRAD_ROWS 1024
int gm_launch_wgrad(const gm_wgrad_args& a, hipStream_t s);
bool gm_wgrad_gather_ok(int n_chunks, int K, int N);

// Rows per weight-gradient chunk for sets of the given sizes.  One workgroup (one CU: 128 accumulator VGPRs x 16 waves)
// takes one chunk and writes a (K+1)xN partial, so the chunk count should sit just under a multiple of the CU count (256) --
// 280 chunks cost two rounds for the work of 1.1 -- and be small: every chunk adds a partial to write and re-read.
// Chunks never straddle two sets (per-task weights).  Returns a multiple of 32.
int gm_num_cus();                         // compute units of the current device (cached per device)
// CUs a stream may use: the device's, unless the stream was created with a CU mask (gm_meta_step's CU-partitioned streams register theirs).
// Persistent kernels size their grid with this.
int gm_stream_cus(hipStream_t s);
void gm_stream_set_cus(hipStream_t s, int cus);
int gm_func_full_lds(const void* fn);     // allow 160 KiB of dynamic LDS for a kernel, once per (device, kernel)
static inline int gm_wgrad_chunk_rows(const std::vector<int32_t>& set_off, int n_cu = gm_num_cus()) {
    const int sets = (int)set_off.size() - 1;
    int64_t total = 0; int mx = 1;
    for (int t = 0; t < sets; ++t) { const int n = set_off[t + 1] - set_off[t]; total += n; mx = n > mx ? n : mx; }
    if (total <= 0) return 128;
    auto chunks_at = [&](int cr) { int64_t c = 0; for (int t = 0; t < sets; ++t) c += (set_off[t + 1] - set_off[t] + cr - 1) / cr; return c; };
    int best = 128; double best_eff = -1.0;
    for (int rounds : {1, 2, 3, 4, 6, 8}) {
        const int64_t cap = (int64_t)n_cu * rounds;
        int lo = 1, hi = (mx + 31) / 32;                 // in units of 32 rows; chunks_at(32*hi) == #non-empty sets
        if (chunks_at(32 * hi) > cap) continue;          // more sets than workgroup slots at this round count
        while (lo < hi) { const int mid = (lo + hi) / 2; if (chunks_at(32 * mid) <= cap) hi = mid; else lo = mid + 1; }
        const int cr = std::max(128, 32 * lo);
        const int64_t c = chunks_at(cr);
        const double eff = (double)total / ((double)((c + n_cu - 1) / n_cu) * n_cu * cr);   // useful rows / rows the rounds have room for
        if (eff > best_eff + 0.01 * gm_knob().wgrad_round_bias) { best_eff = eff; best = cr; }
    }
    return best;
}

// launch profiling by category (bench.py): work = algorithmic bytes (aggregate) or flops (GEMM, weight gradient)
#define GM_PROF_AGG 0
#define GM_PROF_GEMM 1
#define GM_PROF_WGRAD 2
#define GM_PROF_AGG_STRICT 3   // work-only shadow of GM_PROF_AGG: compulsory HBM bytes (a layer-1 gather reads at most the feature table)
#define GM_PROF_GEMM_SPLIT 4   // grouped GEMM launches that ran on the split-bf16 kernel (6 bf16 MFMA flops per fp32 flop; work = fp32 flops)
#define GM_PROF_WGRAD_SPLIT 5  // weight gradients on the split-bf16 kernel
#define GM_PROF_GEMM_SPLIT16 6 // grouped GEMMs on the two-piece fp16 split kernel (3 fp16 MFMA flops per fp32 flop)
#define GM_PROF_WGRAD_SPLIT16 7
#define GM_PROF_STEP_CATS 8     // categories of gm_meta_step (reset at the start of every step)
#define GM_PROF_EX_NODES 8      // extraction, phase A: k_nodes (BFS, sampling, node lists, induced degrees); work = subgraphs
#define GM_PROF_EX_FILL 9       // extraction, phase B: k_fill (batched CSR in both orientations, parents, norms, centres); work = subgraphs
#define GM_PROF_EX_FINAL 10     // batch finalisation (launch tables, hub schedule, gains, per-edge / per-row tables): GPU span incl. the host round trips inside it
#define GM_PROF_GEMM_SPLIT_BYTES 11   // work-only shadow of the split GEMM launches (categories 4 and 6): compulsory HBM bytes 4 rows (K + N) -- the A operand read once, C written once
#define GM_PROF_AGG_BOUND 12          // work-only shadow of category 0 with the partial launches priced as in rounds 2-3 (sources = min(edges, rows), an upper bound)
#define GM_PROF_GEMM_BYTES 13         // work-only shadow of EVERY grouped GEMM launch (categories 1, 4, 6): compulsory HBM bytes, A rows read once + the C rows it stores
#define GM_PROF_WGRAD_BYTES 14        // ... of every weight-gradient launch (categories 2, 5, 7): the A and G rows read once
#define GM_PROF_HEAD 15               // head + prototypical loss (+ head backward) launches: time; work = subgraphs
#define GM_PROF_CATS 16
void gm_prof_begin(int cat, hipStream_t s, int64_t work);
void gm_prof_end(int cat, hipStream_t s);
void gm_prof_reset(int n_cats = GM_PROF_CATS);
void gm_prof_reset_cat(int cat);
bool gm_prof_enabled();
// distinct source rows of the in-edges of the rows with more than GM_FUSE_MAXDEG sources (what a partial aggregate launch reads): counted on the
// device at first use (a bitmap over the batch rows; synchronises the stream) -- only the launch accounting of bench.py asks for it
int64_t gm_batch_unfused_sources(const gm_batch* b, hipStream_t s);
void gm_prof_note(int cat, int64_t work);      // adds work to a category without timing events
static inline void gm_prof_agg_begin(hipStream_t s, int64_t bytes) { gm_prof_begin(GM_PROF_AGG, s, bytes); }
static inline void gm_prof_agg_end(hipStream_t s) { gm_prof_end(GM_PROF_AGG, s); }
