"""GPU (-m gpu): parent graph too large for the LDS bitmap pair (> ~650k nodes) -> the global-memory bitmap path of the
extraction kernels; bit-exact against the oracle's BFS/sampler/induce on a sample of subgraphs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_extraction_on_a_800k_node_graph():
    import gmeta_amd
    import gmeta_oracle as orc
    from gmeta_amd import synth
    from gmeta_amd.subgraphs import SubgraphBatch
    n = 800_000
    rng = np.random.default_rng(3)
    e = synth.pa_edges(n, 3, rng)
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    store = gmeta_amd.GraphStore([(n, src, dst)], [np.zeros((n, 4), np.float32)])
    seeds = np.stack([np.zeros(24, np.int64), np.concatenate([rng.integers(0, n, 20), np.arange(4)]), -np.ones(24, np.int64)], 1)   # incl. 4 hubs
    B = SubgraphBatch.extract(store, seeds, [0, 24], 2, 500, 222, False)
    G = orc.Graph(n, src, dst)
    par, sub = B.parent(), B.sub_off
    ip, ix = B.csr()
    n_sampled = 0
    for k, (_, i, _) in enumerate(seeds):
        full = orc.khop_nodes(G, int(i), 2)
        want = orc.sample_nodes(full, 500, 222, 0, int(i))
        n_sampled += int(len(full) > 500)
        assert np.array_equal(par[sub[k]:sub[k + 1]], want), k
        oip, oix = orc.induce(G, want)
        assert np.array_equal(ip[sub[k]:sub[k + 1] + 1] - ip[sub[k]], oip)
        assert np.array_equal(ix[ip[sub[k]]:ip[sub[k + 1]]] - sub[k], oix)
    assert n_sampled >= 4
