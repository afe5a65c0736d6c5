// GraphStore: parent graphs (in-edge CSR + derived out-edge CSR) and node features resident in HBM.
// Replaces the pickled DGLGraph list and the `feat` list of train.py:41-44,63-65.
#include <math.h>
#include "gm_internal.h"

extern "C" int gm_store_create(int32_t n_graphs, const int64_t* n_nodes, const int64_t* const* indptr,
                               const int32_t* const* indices, const float* const* feat, int32_t feat_dim,
                               gm_store_t** out) {
    GM_REQUIRE(out, GM_EINVAL, "gm_store_create: out is NULL");
    *out = nullptr;
    GM_REQUIRE(n_graphs >= 1 && n_nodes && indptr && indices && feat && feat_dim >= 1, GM_EINVAL,
               "gm_store_create: bad arguments");
    gm_store* s = new gm_store();
    s->n_graphs = n_graphs; s->feat_dim = feat_dim;
    s->node_off.assign(n_graphs + 1, 0); s->edge_off.assign(n_graphs + 1, 0);
    for (int g = 0; g < n_graphs; ++g) {
        if (n_nodes[g] < 1 || n_nodes[g] > (int64_t)INT32_MAX - 2) { delete s; gm_set_error("graph %d: bad node count", g); return GM_ERANGE; }
        if (indptr[g][0] != 0) { delete s; gm_set_error("graph %d: indptr[0] != 0", g); return GM_EINVAL; }
        s->node_off[g + 1] = s->node_off[g] + n_nodes[g];
        s->edge_off[g + 1] = s->edge_off[g] + indptr[g][n_nodes[g]];
        if (n_nodes[g] > s->max_nodes) s->max_nodes = n_nodes[g];
    }
    s->total_nodes = s->node_off[n_graphs]; s->total_edges = s->edge_off[n_graphs];
    if (s->total_nodes > (int64_t)INT32_MAX - 2) { delete s; gm_set_error("store: more than 2^31 nodes in total"); return GM_ERANGE; }

    // host staging: global in-CSR, derived out-CSR (stable counting sort: destinations ascending per source)
    std::vector<int64_t> in_ptr(s->total_nodes + 1), out_ptr(s->total_nodes + 1, 0);
    std::vector<int32_t> in_idx(s->total_edges ? s->total_edges : 1), out_idx(s->total_edges ? s->total_edges : 1);
    for (int g = 0; g < n_graphs; ++g) {
        const int64_t n = n_nodes[g], no = s->node_off[g], eo = s->edge_off[g];
        for (int64_t v = 0; v < n; ++v) {
            if (indptr[g][v + 1] < indptr[g][v]) { delete s; gm_set_error("graph %d: indptr not monotone", g); return GM_EINVAL; }
            in_ptr[no + v] = eo + indptr[g][v];
        }
        const int64_t ne = indptr[g][n];
        for (int64_t e = 0; e < ne; ++e) {
            const int32_t u = indices[g][e];
            if (u < 0 || u >= n) { delete s; gm_set_error("graph %d: edge source %d out of range", g, u); return GM_EINVAL; }
            in_idx[eo + e] = u;
            out_ptr[no + u + 1] += 1;
        }
    }
    in_ptr[s->total_nodes] = s->total_edges;
    for (int64_t v = 0; v < s->total_nodes; ++v) out_ptr[v + 1] += out_ptr[v];
    {
        std::vector<int64_t> cur(out_ptr.begin(), out_ptr.end() - 1);
        for (int g = 0; g < n_graphs; ++g) {
            const int64_t n = n_nodes[g], no = s->node_off[g];
            for (int64_t v = 0; v < n; ++v)
                for (int64_t e = in_ptr[no + v]; e < in_ptr[no + v + 1]; ++e) out_idx[cur[no + in_idx[e]]++] = (int32_t)v;
        }
    }
    s->symmetric = s->total_edges > 0 && in_ptr == out_ptr && in_idx == out_idx && getenv("GM_EXTRACT_NO_SYM") == nullptr;
    hipStream_t st = nullptr;
    int rc = GM_OK;
#define UP(dptr, vec, T)                                                                              \
    if (rc == GM_OK) {                                                                                \
        rc = gm_dev_alloc((void**)&(dptr), (vec).size() * sizeof(T), st);                             \
        if (rc == GM_OK && hipMemcpy((dptr), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { \
            gm_set_error("gm_store_create: upload failed"); rc = GM_EHIP;                             \
        }                                                                                             \
    }
    UP(s->d_node_off, s->node_off, int64_t)
    UP(s->d_in_ptr, in_ptr, int64_t)
    UP(s->d_in_idx, in_idx, int32_t)
    UP(s->d_out_ptr, out_ptr, int64_t)
    UP(s->d_out_idx, out_idx, int32_t)
#undef UP
    {
        const int pad_on = gm_knob().feat_pad;
        // 1 (default): pad only widths the vector kernels cannot take as they are (not a multiple of 4: 50, 5, ...); 2: always; 0: never.
        // Aligned widths keep their native stride: padding is exact (zeros) but changes which kernels run, i.e. the fp summation order.
        s->feat_ld = (pad_on == 2 || (pad_on == 1 && feat_dim % 4 != 0)) ? gm_pad_feat(feat_dim) : feat_dim;
    }
    const size_t feat_bytes = (size_t)s->total_nodes * s->feat_ld * sizeof(float);
    if (rc == GM_OK) rc = gm_dev_alloc((void**)&s->d_feat, feat_bytes, st);
    if (rc == GM_OK && s->feat_ld != feat_dim && hipMemset(s->d_feat, 0, feat_bytes) != hipSuccess) { gm_set_error("gm_store_create: feature memset failed"); rc = GM_EHIP; }
    for (int g = 0; g < n_graphs && rc == GM_OK; ++g) {
        if (hipMemcpy2D(s->d_feat + s->node_off[g] * s->feat_ld, (size_t)s->feat_ld * sizeof(float), feat[g], (size_t)feat_dim * sizeof(float),
                        (size_t)feat_dim * sizeof(float), (size_t)n_nodes[g], hipMemcpyHostToDevice) != hipSuccess) {
            gm_set_error("gm_store_create: feature upload failed"); rc = GM_EHIP;
        }
    }
    s->h_feat_amax.assign(n_graphs, 0.f); s->h_feat_mean.assign(n_graphs, 0.f);
    for (int g = 0; g < n_graphs; ++g) {
        float mx = 0.f; double sum = 0.0; int64_t nz = 0;
        const float* f = feat[g];
        for (int64_t i = 0, n = n_nodes[g] * (int64_t)feat_dim; i < n; ++i) {
            const float a = fabsf(f[i]);
            if (!(a <= 3.0e38f)) { mx = INFINITY; continue; }                  // inf / NaN entry: never the two-piece kernels
            if (a > 0.f) { mx = a > mx ? a : mx; sum += a; ++nz; }
        }
        s->h_feat_amax[g] = mx; s->h_feat_mean[g] = nz ? (float)(sum / (double)nz) : 0.f;
    }
    if (rc == GM_OK) rc = gm_dev_alloc((void**)&s->d_feat_amax, sizeof(unsigned), st);
    if (rc == GM_OK && hipMemset(s->d_feat_amax, 0, sizeof(unsigned)) != hipSuccess) { gm_set_error("gm_store_create: memset failed"); rc = GM_EHIP; }
    if (rc == GM_OK) rc = gm_amax(s->d_feat, 0, 0, (int64_t)s->total_nodes * s->feat_ld, 1, s->d_feat_amax, 0, st);
    if (rc == GM_OK && hipStreamSynchronize(st) != hipSuccess) { gm_set_error("gm_store_create: feature bound failed"); rc = GM_EHIP; }
    if (rc != GM_OK) { gm_store_destroy(s); return rc; }
    *out = s;
    return GM_OK;
}

extern "C" void gm_store_destroy(gm_store_t* s) {
    if (!s) return;
    (void)hipDeviceSynchronize();
    gm_dev_free(s->d_node_off, nullptr); gm_dev_free(s->d_in_ptr, nullptr); gm_dev_free(s->d_in_idx, nullptr);
    gm_dev_free(s->d_out_ptr, nullptr); gm_dev_free(s->d_out_idx, nullptr); gm_dev_free(s->d_feat, nullptr); gm_dev_free(s->d_feat_amax, nullptr);
    delete s;
}
