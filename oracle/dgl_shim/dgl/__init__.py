"""TEST INFRASTRUCTURE ONLY -- restatement of the slice of the third-party package
``dgl == 0.4.3post2`` (pinned in the reference's requirements.txt:2) that the reference's
hot path calls.  DGL is NOT vendored in /root/reference and is not installable in the build
container, so its semantics are restated here from DGL 0.4.x's published behaviour and pinned
by our own known-answer tests (tests/test_dgl_restatement.py: hand-computed answers plus
scipy.sparse / networkx cross-checks).  PARITY AT THE DGL BOUNDARY IS THEREFORE UNPINNED by
the reference itself (the reference has no tests); everything above the boundary (learner.py,
meta.py, subgraph_data_processing.py) is executed unmodified from /root/reference by
oracle/make_golden.py.

This package exists only so that oracle/make_golden.py can import the reference modules in
the build container.  It never travels into the product path; nothing under g-meta_amd/
imports it.

Reference call sites covered (file:line relative to /root/reference/G-Meta):
  learner.py:9,27,29,37-40,43-46,154,161   local_var / in_degrees / ndata / update_all / batch_num_nodes
  subgraph_data_processing.py:301-311,327-333  G.in_edges(v)[0]
  subgraph_data_processing.py:316-317,341-342  G.subgraph(nodes), sub.parent_nid
  subgraph_data_processing.py:399-400,405-406  dgl.batch(list)
  meta.py:122,131,...                      g.to(device)
"""
import numpy as np
import torch

from . import function  # noqa: F401


def _as_long(x):
    if isinstance(x, torch.Tensor):
        return x.long().reshape(-1)
    return torch.as_tensor(np.asarray(x).reshape(-1).astype(np.int64))


class DGLGraph(object):
    """Directed multigraph: edge k goes src[k] -> dst[k]; parallel edges and self loops kept."""

    def __init__(self):
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.long)
        self._dst = torch.zeros(0, dtype=torch.long)
        self.ndata = {}
        self.batch_num_nodes = None   # python list for batched graphs (learner.py:161-162)
        self.parent_nid = None        # LongTensor for subgraphs (sdp.py:317)

    # -- construction --------------------------------------------------------------------
    def add_nodes(self, n):
        self._n += int(n)

    def add_edges(self, u, v):
        u, v = _as_long(u), _as_long(v)
        assert u.numel() == v.numel()
        self._src = torch.cat([self._src, u])
        self._dst = torch.cat([self._dst, v])

    # -- queries -------------------------------------------------------------------------
    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.numel())

    def in_edges(self, v):
        """All edges with dst == v, in edge-id order, as (src, dst) LongTensors."""
        v = int(v)
        sel = (self._dst == v).nonzero().reshape(-1)
        return self._src[sel], self._dst[sel]

    def in_degrees(self):
        return torch.bincount(self._dst, minlength=self._n).long()

    def subgraph(self, nodes):
        """Node-induced subgraph; node k of the result is nodes[k] (order preserved); every
        parent edge with both endpoints inside is kept, in parent edge-id order."""
        nodes = _as_long(nodes)
        lut = torch.full((self._n,), -1, dtype=torch.long)
        lut[nodes] = torch.arange(nodes.numel())
        ls, ld = lut[self._src], lut[self._dst]
        keep = (ls >= 0) & (ld >= 0)
        sub = DGLGraph()
        sub.add_nodes(nodes.numel())
        sub.add_edges(ls[keep], ld[keep])
        sub.parent_nid = nodes.clone()
        return sub

    # -- frame plumbing ------------------------------------------------------------------
    def local_var(self):
        g = DGLGraph()
        g._n, g._src, g._dst = self._n, self._src, self._dst
        g.ndata = dict(self.ndata)
        g.batch_num_nodes = self.batch_num_nodes
        g.parent_nid = self.parent_nid
        return g

    def to(self, device):
        return self

    # -- message passing -----------------------------------------------------------------
    def update_all(self, message_func, reduce_func):
        """Only copy_src + sum is used (learner.py:38-39,44-45): out[v] = sum_{(u->v)} h[u];
        zero rows for in-degree 0.  Differentiable through torch autograd."""
        kind_m, src_field, msg_field = message_func
        kind_r, msg_field_r, out_field = reduce_func
        assert kind_m == 'copy_src' and kind_r == 'sum' and msg_field == msg_field_r
        h = self.ndata[src_field]
        out = torch.zeros((self._n,) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
        out = out.index_add(0, self._dst.to(h.device), h[self._src.to(h.device)])
        self.ndata[out_field] = out


def batch(graph_list):
    """Disjoint union; nodes relabelled consecutively in list order (sdp.py:399-406)."""
    g = DGLGraph()
    off = 0
    srcs, dsts, sizes = [], [], []
    for sub in graph_list:
        srcs.append(sub._src + off)
        dsts.append(sub._dst + off)
        sizes.append(sub._n)
        off += sub._n
    g.add_nodes(off)
    if srcs:
        g.add_edges(torch.cat(srcs), torch.cat(dsts))
    g.batch_num_nodes = [int(s) for s in sizes]
    return g
