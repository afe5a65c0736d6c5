"""CPU: pins the restated DGL 0.4.3 surface (oracle/dgl_shim) that oracle/make_golden.py puts under
the reference's modules.  DGL is third-party and absent, and the reference has no tests, so these
known answers are OURS: hand-computed on tiny graphs plus cross-checks against scipy.sparse and
networkx.  They also pin the oracle's own Graph/induce/agg against the same answers."""
import os
import sys

import networkx as nx
import numpy as np
import pytest
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'dgl_shim'))
import dgl                     # noqa: E402  (restated surface)
import dgl.function as fn      # noqa: E402
import gmeta_oracle as orc     # noqa: E402


def _g(n, edges):
    g = dgl.DGLGraph(); g.add_nodes(n)
    e = np.array(edges, np.int64).reshape(-1, 2)
    g.add_edges(e[:, 0], e[:, 1])
    return g


# triangle 0->1->2->0 plus a parallel edge 0->1, a self loop 2->2 and an isolated node 3
TRI = [(0, 1), (1, 2), (2, 0), (0, 1), (2, 2)]


def test_in_edges_in_degrees_known_answers():
    g = _g(4, TRI)
    assert g.in_edges(1)[0].tolist() == [0, 0]            # duplicates kept
    assert sorted(g.in_edges(2)[0].tolist()) == [1, 2]    # self loop counted
    assert g.in_edges(3)[0].tolist() == []
    assert g.in_degrees().tolist() == [1, 2, 2, 0]


def test_update_all_known_answer_and_grad():
    g = _g(4, TRI)
    h = torch.tensor([[1., 10.], [2., 20.], [3., 30.], [4., 40.]], dtype=torch.float64, requires_grad=True)
    g.ndata['h'] = h
    g.update_all(fn.copy_src(src='h', out='m'), fn.sum(msg='m', out='h'))
    out = g.ndata['h']
    assert out.tolist() == [[3., 30.], [2., 20.], [5., 50.], [0., 0.]]   # row1 = h0+h0, row2 = h1+h2, row3 isolated
    out.sum().backward()
    assert h.grad[:, 0].tolist() == [2., 1., 2., 0.]                      # out-degrees incl. parallel/self


def test_subgraph_order_and_edges():
    g = _g(4, TRI)
    sub = g.subgraph(np.array([2, 0]))            # order preserved: local 0 = parent 2
    assert sub.parent_nid.tolist() == [2, 0]
    e = sorted(zip(sub._src.tolist(), sub._dst.tolist()))
    assert e == [(0, 0), (0, 1)]                  # 2->2 and 2->0
    b = dgl.batch([sub, g.subgraph(np.array([0, 1]))])
    assert b.batch_num_nodes == [2, 2] and isinstance(b.batch_num_nodes, list)
    assert sorted(zip(b._src.tolist(), b._dst.tolist())) == [(0, 0), (0, 1), (2, 3), (2, 3)]


@pytest.mark.parametrize('seed', range(5))
def test_against_scipy_and_networkx(seed):
    rng = np.random.default_rng(seed)
    n, e = 40, 160
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    g = _g(n, np.stack([src, dst], 1))
    A = sp.coo_matrix((np.ones(e), (dst, src)), shape=(n, n)).tocsr()   # A[v,u] = #edges u->v
    X = rng.standard_normal((n, 7))
    g.ndata['h'] = torch.tensor(X)
    g.update_all(fn.copy_src(src='h', out='m'), fn.sum(msg='m', out='h'))
    np.testing.assert_allclose(g.ndata['h'].numpy(), A @ X, atol=1e-12)
    assert np.array_equal(g.in_degrees().numpy(), np.asarray(A.sum(1)).reshape(-1).astype(np.int64))
    M = nx.MultiDiGraph(); M.add_nodes_from(range(n)); M.add_edges_from(zip(src.tolist(), dst.tolist()))
    nodes = np.sort(rng.choice(n, 12, replace=False))
    sub = g.subgraph(nodes)
    ref = sorted((u, v) for u, v, _ in M.subgraph(nodes.tolist()).edges(keys=True))
    got = sorted(zip(nodes[sub._src.numpy()].tolist(), nodes[sub._dst.numpy()].tolist()))
    assert got == ref
    for v in range(n):
        assert sorted(g.in_edges(v)[0].tolist()) == sorted(u for u, _ in M.in_edges(v))
    # the oracle's own containers agree with the same ground truth
    G = orc.Graph(n, src, dst)
    for v in range(n):
        assert sorted(G.preds(v).tolist()) == sorted(u for u, _ in M.in_edges(v))
    ip, ix = orc.induce(G, nodes)
    got2 = sorted(zip(nodes[ix].tolist(), np.repeat(nodes, np.diff(ip)).tolist()))
    assert got2 == ref
    np.testing.assert_allclose(orc.agg(G.indptr, G.indices.astype(np.int64), X.astype(np.float32)), (A @ X), atol=1e-4)
    for h in (1, 2, 3):
        bfs = {int(nodes[0])}
        fr = {int(nodes[0])}
        for _ in range(h):
            fr = {u for v in fr for u, _ in M.in_edges(v)}
            bfs |= fr
        assert set(orc.khop_nodes(G, int(nodes[0]), h).tolist()) == bfs


def test_update_all_gradcheck():
    g = _g(4, TRI)

    def f(h):
        gg = g.local_var(); gg.ndata['h'] = h
        gg.update_all(fn.copy_src(src='h', out='m'), fn.sum(msg='m', out='h'))
        return gg.ndata['h']
    assert torch.autograd.gradcheck(f, (torch.randn(4, 3, dtype=torch.float64, requires_grad=True),))
