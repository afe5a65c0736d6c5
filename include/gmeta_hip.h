/* gmeta_hip.h -- C ABI of libgmeta_hip.so: the MI355X (gfx950) implementation of G-Meta's
 * inner-loop hot path.  Plain pointers and sizes only; no torch types.
 *
 * The reference (mims-harvard/G-Meta) is pure Python and has no FFI layer: its boundary is the
 * Python call surface Subgraphs.__getitem__/collate -> Meta.forward/finetunning ->
 * Classifier.forward, with all native work delegated to the third-party DGL 0.4.3 + torch 1.5.
 * Each entry point below names the reference code it replaces (paths relative to
 * /root/reference/G-Meta/).  The Python host mirror in g-meta_amd/ binds these through ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: every function returns GM_OK (0) or a negative GM_E* code; gm_last_error()
 * returns a thread-local message for the last failure.  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  Pointers documented "device" are HBM addresses on the
 * current HIP device; "host" are ordinary host pointers.  Handles (gm_store_t, gm_batch_t) own
 * their HBM and are released by the matching *_destroy; workspaces are caller-provided.
 * A handle is not thread-safe; distinct handles may be used from distinct threads.
 */
#ifndef GMETA_HIP_H
#define GMETA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_EINVAL (-1)  /* bad argument (also: update_step < 2, unequal class counts)      */
#define GM_ENOMEM (-2)  /* HBM / workspace too small                                      */
#define GM_EHIP (-3)    /* a HIP runtime call failed                                      */
#define GM_ERANGE (-4)  /* size outside what the kernels support (e.g. a batch above 2^31 rows) */
#define GM_MAX_GCN 4

typedef struct gm_store gm_store_t; /* parent graphs (in/out CSR) + node features, resident in HBM */
typedef struct gm_batch gm_batch_t; /* a batched set of induced subgraphs (one or many task sets)   */

typedef struct gm_seed { int32_t graph, i, j; } gm_seed_t; /* j = -1 for node classification */

/* Model description == the `config` list built at train.py:67-75:
 * [('GraphConv',[dims[0],dims[1]]), ..., ('Linear',[dims[n_gcn], n_out])] (+('LinkPred',[True])).
 * Parameter vector layout (== Classifier.vars order, learner.py:81-97), flat fp32:
 *   W_1[dims0 x dims1] row-major [in,out], b_1[dims1], ..., W_lin[n_out x (dims[n_gcn]*(1+link_pred))], b_lin[n_out] */
typedef struct gm_model {
    int32_t n_gcn;
    int32_t dims[GM_MAX_GCN + 1];
    int32_t n_out;
    int32_t link_pred;
} gm_model_t;

/* Hyper-parameters read by Meta.__init__ (meta.py:85-92). */
typedef struct gm_hparams {
    float update_lr;       /* args.update_lr                                            */
    int32_t update_step;   /* K: args.update_step (train) or args.update_step_test      */
    int32_t k_spt;         /* n_support of proto_loss_spt (meta.py:123)                 */
    int32_t need_meta_grad;/* 1 = Meta.forward (meta.py:101-173), 0 = finetunning (175-234) */
    int32_t hoist_z1;      /* 0 = reference-equivalent schedule (every forward re-aggregates layer 1);
                              1 = aggregate layer-1 input once per call (loop-invariant)   */
    int32_t serialize;     /* 0 = support chain and query evaluations on two HIP streams (default);
                              1 = everything on `stream` (for per-kernel timing / profiling)  */
    int32_t sparse_bwd;    /* 0 = dense backward over every subgraph row (reference-equivalent schedule, default);
                              1 = exact row-sparse backward: only the head touches the last layer, so dQ_L is non-zero
                              at centre rows only and dQ_{L-1} only along in-edges of centres -- same sums without
                              the structural zeros (models with <= 2 aggregate-first GCN layers; else falls back) */
    int32_t cone;          /* 0 = every layer is evaluated on every subgraph row, as DGL does (reference-equivalent
                              schedule, default); 1 = receptive-field schedule: only the centre rows of the last GCN layer
                              reach the head (learner.py:159-170), so layer l is evaluated only on the rows that are
                              (L-l) in-hops upstream of a centre, forward and backward -- the same sums for every row
                              that matters, nothing for the rows that cannot influence logits or gradients.
                              Supersedes sparse_bwd; falls back to the dense schedule if a pair has i == j */
} gm_hparams_t;

const char* gm_last_error(void);
int gm_version(void);

/* ---- GraphStore: replaces the list of DGLGraph objects + `feat` list (train.py:41-44,63-65).
 * indptr[g] (host int64[n_nodes[g]+1]) / indices[g] (host int32) = IN-edge CSR of graph g:
 * row v lists the sources u of every edge u->v (parallel edges and self loops kept, DGL
 * multigraph semantics of G.in_edges(v), sdp.py:301).  feat[g] = host fp32 [n_nodes[g], feat_dim]. */
int gm_store_create(int32_t n_graphs, const int64_t* n_nodes, const int64_t* const* indptr,
                    const int32_t* const* indices, const float* const* feat, int32_t feat_dim,
                    gm_store_t** out);
void gm_store_destroy(gm_store_t* s);

/* ---- Extraction: replaces Subgraphs.generate_subgraph / generate_subgraph_link_pred
 * (sdp.py:295-346: h-hop in-neighbour expansion, node sampling, G.subgraph) and dgl.batch
 * (sdp.py:399-406) for n_sets sets at once.  seeds/set_offsets are host arrays; set s owns seeds
 * [set_offsets[s], set_offsets[s+1]).  h in {1,2,3} (ignored when link_pred: i side 2 hops,
 * j side 1 hop -- the reference's sdp.py:332 behaviour).  Nodes inside a subgraph are in
 * ASCENDING parent id.  If a neighbourhood has more than sample_nodes nodes, sample_nodes of them
 * are kept by a keyed permutation of (rng_seed, graph, i, j) and the centre(s) re-added
 * (sdp.py:312-314,337-339). */
int gm_extract(const gm_store_t* store, const gm_seed_t* seeds, int32_t n_seeds,
               const int32_t* set_offsets, int32_t n_sets, int32_t h, int32_t sample_nodes,
               uint64_t rng_seed, int32_t link_pred, void* stream, gm_batch_t** out);
/* The support AND the query batch of a meta-batch in one build (Subgraphs.__getitem__ extracts both per task, sdp.py:363-386; dgl.batch twice,
 * sdp.py:399-406): the same two batches two gm_extract calls return -- bit for bit -- from ONE launch of the node-set kernel and ONE of the fill
 * kernel over all subgraphs (a 32-task arxiv meta-batch: 288 + 2,304 of them) and one host round trip for both finalisations. */
int gm_extract_pair(const gm_store_t* store, const gm_seed_t* seeds_a, int32_t n_seeds_a, const int32_t* set_offsets_a, int32_t n_sets_a,
                    const gm_seed_t* seeds_b, int32_t n_seeds_b, const int32_t* set_offsets_b, int32_t n_sets_b,
                    int32_t h, int32_t sample_nodes, uint64_t rng_seed, int32_t link_pred, void* stream,
                    gm_batch_t** out_a, gm_batch_t** out_b);
/* Same, but the node set of every subgraph is given (host, ascending, concatenated; subgraph k
 * owns nodes_flat[nodes_off[k]..nodes_off[k+1])): G.subgraph(nodes) + dgl.batch only.  Used to
 * replay node sets sampled elsewhere (e.g. by the reference's numpy RNG). */
int gm_batch_from_nodes(const gm_store_t* store, const gm_seed_t* seeds, int32_t n_seeds,
                        const int32_t* set_offsets, int32_t n_sets, const int32_t* nodes_flat,
                        const int64_t* nodes_off, int32_t link_pred, void* stream, gm_batch_t** out);
/* dgl.batch over already-built batches (sets are appended in order).  Inputs stay valid. */
int gm_batch_concat(const gm_batch_t* const* parts, int32_t n_parts, void* stream, gm_batch_t** out);
/* Receptive-field tables for gm_hparams_t.cone with an n_gcn-layer model (built on `stream`, cached in the
 * batch; gm_meta_ws_bytes/gm_meta_step build them on first use otherwise).  level_rows/level_edges
 * (host int64[n_gcn+1]): rows of level l and edges from level l-1 into level l; *ok = 0 when the batch
 * cannot use the schedule.  gm_batch_cone_read copies one table to the host (tests): what = 0 rows,
 * 1 indptr, 2 indices, 3 indptr_t, 4 indices_t, 5 set offsets. */
int gm_batch_prepare_cone(const gm_batch_t* b, int32_t n_gcn, void* stream);
/* The same for the two batches of a meta-batch (support, query) with ONE pair of host round trips for both
 * (the builder thread of Subgraphs.batches calls this; results identical to two gm_batch_prepare_cone calls). */
int gm_batch_prepare_cone_pair(const gm_batch_t* a, const gm_batch_t* b, int32_t n_gcn, void* stream);
int gm_batch_cone_dims(const gm_batch_t* b, int32_t n_gcn, int32_t* ok, int64_t* level_rows, int64_t* level_edges);
int gm_batch_cone_read(const gm_batch_t* b, int32_t n_gcn, int32_t level, int32_t what, void* host, int64_t host_bytes);
void gm_batch_destroy(gm_batch_t* b);

/* Sizes: rows = total nodes, edges = total induced edges, subs = subgraphs, sets = task sets,
 * centres = 1 (node-clf) or 2 (link-pred). */
int gm_batch_dims(const gm_batch_t* b, int64_t* rows, int64_t* edges, int32_t* subs, int32_t* sets,
                  int32_t* centres);
enum gm_field {
    GM_F_SUB_OFF = 0,   /* int32[subs+1]  row offset of each subgraph == cumsum(batch_num_nodes) (learner.py:161-163) */
    GM_F_SET_SUB_OFF,   /* int32[sets+1]  subgraph range of each set                                                */
    GM_F_PARENT,        /* int32[rows]    parent node id of each row == sub.parent_nid (sdp.py:317)                  */
    GM_F_GRAPH,         /* int32[subs]    parent graph of each subgraph                                              */
    GM_F_INDPTR,        /* int32[rows+1]  in-edge CSR of the batched induced graph                                   */
    GM_F_INDICES,       /* int32[edges]   source ROW of every in-edge                                                */
    GM_F_INDPTR_T,      /* int32[rows+1]  by-source CSR (for the backward aggregate)                                 */
    GM_F_INDICES_T,     /* int32[edges]   destination ROW of every out-edge                                          */
    GM_F_CENTRE,        /* int32[subs*centres] local index of the centre(s) inside each subgraph (sdp.py:318-319)    */
    GM_F_NORM,          /* float[rows]    in_degree.clamp(1)^-0.5 (learner.py:29)                                    */
    GM_F_FEAT_ROW       /* int32[rows]    row of the store's feature matrix for each batch row                       */
};
/* Copies a field to host memory (synchronises `stream` internally). */
int gm_batch_read(const gm_batch_t* b, int32_t field, void* host_dst, int64_t bytes);
/* Device address of a field (valid until gm_batch_destroy). */
int gm_batch_device_ptr(const gm_batch_t* b, int32_t field, void** dptr);

/* ---- Feature gather: replaces np.vstack([feat[g][ids] ...]) + H2D (meta.py:119-120,193-194).
 * x_out: device fp32 [rows, feat_dim]. */
int gm_gather_features(const gm_batch_t* b, float* x_out, void* stream);

/* ---- GCN building blocks (GraphConv.forward, learner.py:25-56), exported for tests/profiling.
 * out[v,:] = s_out[v] * sum_{u in row v} s_in[u] * x[u,:]   (s_in / s_out may be NULL = 1).
 * transposed != 0 runs on the by-source CSR (autograd backward of update_all).  If gather != 0,
 * x is ignored and rows are read from the store's features through GM_F_FEAT_ROW. */
int gm_aggregate(const gm_batch_t* b, int32_t transposed, int32_t gather, const float* x, int32_t width,
                 const float* s_in, const float* s_out, float* out, void* stream);
int64_t gm_aggregate_bytes(const gm_batch_t* b, int32_t width); /* algorithmic HBM bytes of one call */

/* ---- Classifier.forward / backward (learner.py:134-175) for a batch whose set s uses the
 * parameter vector params + s*param_stride (param_stride = 0: every set shares one vector, the
 * `vars=None` case).  x0: device [rows, dims[0]] or NULL (gather from the store).
 * centre_local: device int32 [subs*centres] `to_fetch` override or NULL (use the batch's own).
 * logits: device fp32 [subs, n_out].  ws must hold gm_gcn_ws_bytes(); it carries the activations
 * from forward to backward (pass the same x0 / centre_local to both).  dlogits: device [subs, n_out];
 * dparams: device, set s written at dparams + s*dparam_stride (dparam_stride >= P).
 * params + s*param_stride must be 16-byte aligned for the vectorised weight loads (else a scalar path runs). */
int64_t gm_model_param_count(const gm_model_t* m);
int64_t gm_gcn_ws_bytes(const gm_batch_t* b, const gm_model_t* m);
int gm_gcn_forward(const gm_batch_t* b, const gm_model_t* m, const float* params, int64_t param_stride,
                   const float* x0, const int32_t* centre_local, float* logits, void* ws, int64_t ws_bytes,
                   void* stream);
int gm_gcn_backward(const gm_batch_t* b, const gm_model_t* m, const float* params, int64_t param_stride,
                    const float* x0, const int32_t* centre_local, const float* dlogits, float* dparams,
                    int64_t dparam_stride, void* ws, int64_t ws_bytes, void* stream);

/* torch.matmul(feat, weight) of GraphConv.forward (learner.py:36,47) over the rows of a batch: out[rows, N] = x[rows, K] @ W_t with
 * W_t = W + set * w_stride ([K, N] row-major; w_stride = 0: one matrix for every set).  Exported for numerics tests of the GEMM kernels.
 * mode -1: what the library would pick (gm_set_gemm_mode + launch size), 0: exact-fp32 MFMA kernels, 1: split-bf16 kernel (three pieces, N = 128 / 256),
 * 2: split-fp16 kernel (two pieces; the bounds of x and of every W_t are taken by the call itself). */
int gm_dense_update(const gm_batch_t* b, const float* x, int32_t K, const float* W, int64_t w_stride, int32_t N, float* out, int32_t mode,
                    void* stream);

/* ---- Prototypical losses (meta.py:28-54 proto_loss_spt, 56-79 proto_loss_qry), per set.
 * y: HOST int32 [subs] labels.  Outputs (device): loss[sets], acc[sets], protos[sets, c_task, n_out]
 * (c_task = the LARGEST number of classes of any set; every set keeps its own class layout -- classes, rows per class --
 * like the per-task calls of meta.py:118-157; set t uses the first classes_t rows of its [c_task, n_out] block),
 * dlogits[subs, n_out] (may be NULL), dprotos[sets, c_task, n_out] (qry only, may be NULL). */
int gm_proto_loss_spt(const gm_batch_t* b, const float* logits, int32_t n_out, const int32_t* y, int32_t n_support,
                      float* loss, float* acc, float* protos, float* dlogits, void* stream);
int gm_proto_loss_qry(const gm_batch_t* b, const float* logits, int32_t n_out, const int32_t* y, const float* protos,
                      int32_t c_task, float* loss, float* acc, float* dlogits, float* dprotos, void* stream);

/* ---- The fused hot path: Meta.forward_ProtoMAML (meta.py:101-173) when need_meta_grad = 1,
 * Meta.finetunning_ProtoMAML (meta.py:175-234) when 0, for ALL sets (tasks) of spt/qry at once.
 * spt and qry must have the same number of sets; set t of each is task t.  y_spt / y_qry: HOST
 * int32 labels per subgraph.  theta: device fp32 [P] (read only).
 * out: device fp32 [P + 2*(K+1) + 1 + sets*(K+1) + 1] (= gm_meta_out_floats; out_floats is the capacity the caller allocated, checked):
 *   [0,P)            SUM over tasks of the first-order meta-gradient (query path at fw_K + prototype
 *                    path through the support forward at fw_{K-1}); zeros when need_meta_grad = 0
 *   [P, P+K+1)       SUM over tasks of losses_q[k]   (meta.py:133,140,155)
 *   [P+K+1, P+2K+2)  SUM over tasks of corrects[k]   (meta.py:134,141,157)
 *   [P+2K+2]         the number of tasks (sets) in this call, as a float (rides along with the all-reduce)
 *   [P+2K+3, ...)    per-task query accuracy [sets, K+1]
 *   [last]           violation word of the opt-in two-piece kernels as a float (0 = none; gm_set_split_pieces): when non-zero, losses_q[K]
 *                    above is NaN -- the step is skipped like a NaN loss on every rank -- and the caller re-runs it with three pieces
 * The caller divides by the (global) task count, applies the NaN guard (meta.py:163) and the
 * optimiser -- after the RCCL all-reduce when tasks are sharded over GPUs. */
int64_t gm_meta_ws_bytes(const gm_batch_t* spt, const gm_batch_t* qry, const gm_model_t* m, const gm_hparams_t* hp);
int64_t gm_meta_out_floats(const gm_batch_t* spt, const gm_model_t* m, const gm_hparams_t* hp);
int gm_meta_step(const gm_batch_t* spt, const gm_batch_t* qry, const int32_t* y_spt, const int32_t* y_qry,
                 const gm_model_t* m, const gm_hparams_t* hp, const float* theta, float* out, int64_t out_floats,
                 void* ws, int64_t ws_bytes, void* stream);

/* After the (optional) all-reduce of out[0 .. P + 2*(K+1)] over the ranks: the mean meta-gradient and the NaN guard of
 * meta.py:161-163, on the device.  head: device, the reduced block; grad: device fp32 [P] <- head[0..P) / task count;
 * found_inf: device fp32 [1] <- 1.0 if losses_q[K] / task count is NaN else 0.0 (a fused Adam skips its step on 1.0, which
 * is the reference's `if torch.isnan(loss_q): pass`).  K1 = update_step + 1. */
int gm_meta_finish(const float* head, int64_t P, int32_t K1, float* grad, float* found_inf, void* stream);
/* gm_meta_finish AND the Adam step of meta.py:97,161-169 (optim.Adam(lr = meta_lr): betas (0.9, 0.999), eps 1e-8, no weight decay) in one launch.
 * theta / exp_avg / exp_avg_sq: device fp32 [P], updated in place -- the parameters and the optimiser state torch.optim.Adam keeps (the host mirror
 * makes them views of flat buffers); steps: device fp32 [n_steps], the optimiser's per-parameter step counters (all equal), incremented together;
 * grad <- head[0..P) / task count as in gm_meta_finish; a NaN reduced query loss leaves theta, the state and the counters untouched and sets
 * found_inf = 1 (`if torch.isnan(loss_q): pass`).  ticket: device uint32 [1], zero before the first call (the kernel leaves it zero). */
int gm_meta_finish_adam(const float* head, int64_t P, int32_t K1, float* theta, float* exp_avg, float* exp_avg_sq, float* grad, float* steps, int32_t n_steps,
                        float lr, float beta1, float beta2, float eps, float* found_inf, uint32_t* ticket, void* stream);

/* Update-GEMM arithmetic (the reference multiplies in fp32: torch.matmul(feat, weight), learner.py:36,47).
 * mode 0: exact fp32 on v_mfma_f32_32x32x2_f32 everywhere.  mode 1 (default): large N = 128 / 256 launches run on the bf16 matrix cores
 * with every fp32 operand split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits: all 24 bits of both operands enter the
 * product) and the six products of weight >= 2^-16 accumulated in fp32 (dropped terms <= 1.2e-7 |a||b|: within one fp32 rounding per
 * product of the exact one; measured error against fp64 <= the fmaf chain's).  Also settable with GM_GEMM_MODE=f32|split.
 * gm_set_split_pieces(2) / GM_SPLIT_PIECES=2 is an OPT-IN fast mode, never the default: inside gm_meta_step (dense schedule,
 * aggregate-first layers) the split kernels then take TWO fp16 pieces per operand -- 22 significand bits, i.e. NARROWER than the
 * reference's fp32 operands -- with three products a_h b_h + a_h b_m + a_m b_h under per-task power-of-two scales derived from magnitude
 * bounds that the producing kernels record on the device (csrc/gm_bound.h).  Its guards: a feature table whose largest entry sits more
 * than 2^14 above its mean magnitude keeps the three-piece kernels for layer 1; a fast weight that outgrows the step's weight bound is
 * detected on the device and reported in the last float of gm_meta_step's `out` (the host mirror then re-runs the step three-piece).
 * gm_get_split_pieces() = 2 or 3. */
void gm_set_gemm_mode(int32_t mode);
int32_t gm_get_gemm_mode(void);
void gm_set_split_pieces(int32_t pieces);   /* 2, 3, or -1 = back to the environment variable (default 3) */
int32_t gm_get_split_pieces(void);
/* Tuning knob by the name of its environment variable, after start-up; GM_EINVAL for unknown names.  Tests use it to
 * force the large-launch kernels onto small fixtures (GM_GEMM_SPLIT_MIN_TILES, GM_SPLIT16_MIN_ROWS); not synchronised with concurrent calls. */
int gm_set_tuning(const char* name, int32_t value);
int32_t gm_get_tuning(const char* name);      /* current value of such a knob (0 for an unknown name) */
int32_t gm_tuning_epoch(void);   /* number of gm_set_tuning changes so far (cache key for sizes that depend on the knobs) */

/* Fused aggregate + update for forward passes nobody differentiates (the query evaluations of the inner steps in gm_meta_step,
 * i.e. meta.py:129-141,152-154 before the last step, and every query pass of finetunning): rows with one or two sources are
 * aggregated inside the GEMM's operand feeders, in the aggregate kernel's own fma order -- same floats, bit for bit -- and their
 * Z rows never travel through HBM.  on = 1 / 0, -1 = back to the environment variable GM_FUSE_AGG (default 1). */
void gm_set_fuse_agg(int32_t on);
int32_t gm_get_fuse_agg(void);

/* Profiling aid for bench.py: HIP-event time (ms) of the aggregate launches of the last
 * gm_meta_step on this thread, their count and their summed algorithmic bytes.  Events are only
 * recorded when gm_profile_enable(1) was called (they add a few microseconds per launch). */
void gm_profile_enable(int32_t on);
int gm_profile_aggregate(double* total_ms, int64_t* launches, int64_t* algorithmic_bytes);
/* Same for one launch category: 0 = aggregate (work = algorithmic bytes), 1 = grouped GEMM (forward and dZ; work =
 * flops 2*rows*K*N), 2 = weight gradient incl. its reduction (work = flops), 3 = the aggregate launches of category 0
 * priced at their COMPULSORY HBM bytes (a layer-1 launch that gathers from the store's feature table reads at most the
 * whole table, not rows*width; total_ms is 0 for this category -- use category 0's), 4 / 5 = the grouped GEMMs / weight
 * gradients that ran on the split-bf16 kernels (work = flops of the fp32 product; the kernels issue 6 bf16 MFMA flops per
 * fp32 flop) -- categories 1 / 2 then hold only the launches on the exact-fp32 MFMA kernels; 6 / 7 = the grouped GEMMs / weight
 * gradients on the two-piece fp16 split kernels (3 fp16 MFMA flops per fp32 flop).  Categories 8 / 9 / 10 belong to gm_extract on this thread
 * (not reset by gm_meta_step; gm_profile_enable resets them): 8 = k_nodes (h-hop expansion, sampling, node lists, induced degrees),
 * 9 = k_fill (the batched CSR in both orientations), work = subgraphs; 10 = batch finalisation (GPU span including its host round trips).
 * 12 = work-only: bytes by which the rounds-2/3 pricing of the partial aggregate launches (sources = min(edges, rows)) exceeds the exact count.
 * 11 = work-only shadow of categories 4 + 6: compulsory HBM bytes of the split GEMM launches, 4 rows (K + N) (A read once, C written once).
 * 13 / 14 = work-only: compulsory HBM bytes of EVERY grouped GEMM launch (A read once + the C rows stored) / of every weight-gradient launch
 * (A and G read once); 15 = the head + prototypical-loss launches (k_head_loss; work = subgraphs).  Under the receptive-field schedule
 * (gm_hparams_t.cone) category 0 is priced on the rows each level-to-level aggregate touches: destination-level row bounds and norms, the
 * edges between the two levels, every source-level row read once, every destination row written once. */
int gm_profile_read(int32_t category, double* total_ms, int64_t* launches, int64_t* work);
/* The same per launch: ms[k] / work[k] of the k-th timed launch of the category since the last reset (at most cap); returns their number (< 0: error). */
int gm_profile_read_launches(int32_t category, double* ms, int64_t* work, int32_t cap);
/* Timeline / phase probes used by tools/ (gm_debug_stamp, gm_stream_debug, gm_head_loss_debug) are NOT part of this library: they exist only in
 * the probe build, libgmeta_hip_probes.so (`python g-meta_amd/build.py --probes`), and are declared in include/gmeta_hip_probes.h. */

#ifdef __cplusplus
}
#endif
#endif /* GMETA_HIP_H */
