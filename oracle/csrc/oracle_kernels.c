/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the message-passing reduce that the
 * reference delegates to DGL 0.4.3 (`graph.update_all(fn.copy_src, fn.sum)`,
 * /root/reference/G-Meta/learner.py:38-39,44-45): out[v] = sum over in-edges (u->v) of x[u].
 * Used by oracle/gmeta_oracle.py (checker) and by bench.py's cpu_baseline leg (kind "port").
 * Never linked into the product library. */
#include <stdint.h>
#include <string.h>

void oracle_agg_f32(int64_t n, int64_t F, const int64_t* indptr, const int64_t* indices,
                    const float* x, float* out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t v = 0; v < n; ++v) {
        float* o = out + v * F;
        memset(o, 0, sizeof(float) * (size_t)F);
        for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            const float* s = x + indices[e] * F;
            for (int64_t f = 0; f < F; ++f) o[f] += s[f];
        }
    }
}

/* sdp.py:300-311 restated as a BFS over the in-edge CSR: marks {i} U <=h-step predecessors. */
int64_t oracle_khop_mark(int64_t n, const int64_t* indptr, const int32_t* indices, int64_t seed,
                         int h, uint8_t* seen, int32_t* frontier_a, int32_t* frontier_b) {
    int64_t na = 1, count = 1;
    memset(seen, 0, (size_t)n);
    seen[seed] = 1; frontier_a[0] = (int32_t)seed;
    for (int hop = 0; hop < h; ++hop) {
        int64_t nb = 0;
        for (int64_t k = 0; k < na; ++k) {
            int64_t v = frontier_a[k];
            for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                int32_t u = indices[e];
                if (!seen[u]) { seen[u] = 1; frontier_b[nb++] = u; ++count; }
            }
        }
        int32_t* t = frontier_a; frontier_a = frontier_b; frontier_b = t; na = nb;
    }
    return count;
}

/* autograd backward of update_all(copy_src, sum): grad_x[u] = sum over out-edges (u->v) of g[v].
 * indptr_t/indices_t = the same edges grouped by SOURCE (destinations in edge order), so every
 * output row is owned by one thread and the summation order is fixed. */
void oracle_agg_t_f32(int64_t n, int64_t F, const int64_t* indptr_t, const int64_t* indices_t,
                      const float* g, float* out) {
    oracle_agg_f32(n, F, indptr_t, indices_t, g, out);
}
