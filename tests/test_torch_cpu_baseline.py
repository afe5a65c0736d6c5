"""CPU: the torch-CPU variant of bench.py's cpu_baseline (oracle/torch_cpu_baseline.py: index_add_ + matmul + autograd, the
closest analogue of the reference's DGL-CPU path) computes what the numpy oracle computes -- and therefore what the
reference computes (the oracle is pinned to the reference's golden outputs in tests/test_oracle_golden.py).  Also pins the
OpenMP transposed aggregate the numpy oracle's backward uses."""
import numpy as np
import pytest

import gmeta_oracle as orc
import torch_cpu_baseline as tcb
from golden_util import Fixture


@pytest.mark.parametrize('case', ['g0_disjoint_h1', 'g2_shared', 'g3_linkpred', 'g5_in_gt_out'])
def test_torch_cpu_variant_equals_oracle(case):
    fx = Fixture(case)
    graphs = fx.graphs()
    n_gcn = len([1 for n, _ in fx.config if n == 'GraphConv'])
    for t in range(fx.T):
        spt = orc.extract_batch(graphs, fx.z['spt_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('spt', t))
        qry = orc.extract_batch(graphs, fx.z['qry_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('qry', t))
        xs, xq = spt.features(fx.feats), qry.features(fx.feats)
        lq, aq, mg = orc.task_inner_loop(spt, qry, xs, xq, fx.z['y_spt'][t], fx.z['y_qry'][t], fx.vars0, fx.config, fx.args['k_spt'],
                                         fx.args['update_lr'], fx.K, True)
        lq2, aq2, mg2 = tcb.task_inner_loop(spt, qry, xs, xq, fx.z['y_spt'][t], fx.z['y_qry'][t], fx.vars0, n_gcn, fx.args['k_spt'],
                                            fx.args['update_lr'], fx.K, True)
        np.testing.assert_allclose(lq2, lq, atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(aq2, aq, atol=1e-6)
        for a, b in zip(mg2, mg):
            np.testing.assert_allclose(a, b, atol=1e-4, rtol=1e-3)


def test_openmp_transposed_aggregate_equals_scatter_add():
    rng = np.random.default_rng(0)
    fx = Fixture('g1_sampled_h2')
    graphs = fx.graphs()
    b = orc.extract_batch(graphs, fx.z['qry_seeds'][0], fx.args['h'], fx.args['sample_nodes'], 222, fx.link)
    g = rng.standard_normal((b.n, 24)).astype(np.float32)
    want = np.zeros_like(g)
    np.add.at(want, b.indices, g[b.dst])
    assert orc._load_c() and hasattr(orc._load_c(), 'oracle_agg_t_f32')
    np.testing.assert_allclose(orc.agg_t(b, g), want, atol=1e-5, rtol=1e-5)
    x = rng.standard_normal((b.n, 24)).astype(np.float32)
    lhs = float((orc.agg(b.indptr, b.indices, x).astype(np.float64) * g).sum()); rhs = float((x.astype(np.float64) * orc.agg_t(b, g)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))                      # <A x, g> == <x, A^T g>
