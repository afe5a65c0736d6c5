"""GraphStore: parent graphs + node features resident in HBM.  Replaces the pickled list of DGLGraph
objects and the `feat` list that train.py:41-44,63-65 load and hand to Subgraphs / Meta."""
import ctypes as C

import numpy as np

from . import _lib


def edges_to_in_csr(n, src, dst):
    """Directed multigraph edge list (edge k: src[k] -> dst[k]) -> in-edge CSR.  Within a row the
    parent edge-id order is kept (DGL's in_edges order, sdp.py:301)."""
    src = np.asarray(src, np.int64).reshape(-1); dst = np.asarray(dst, np.int64).reshape(-1)
    if src.shape != dst.shape:
        raise ValueError('src and dst must have the same length')
    if len(src) and (src.min() < 0 or src.max() >= n or dst.min() < 0 or dst.max() >= n):
        raise ValueError('edge endpoint out of range')
    order = np.argsort(dst, kind='stable')
    indptr = np.zeros(n + 1, np.int64)
    np.add.at(indptr, dst + 1, 1)
    return np.cumsum(indptr), np.ascontiguousarray(src[order].astype(np.int32))


class GraphStore:
    """`graphs`: list of (n_nodes, src, dst) edge lists or (indptr, indices) in-CSR pairs;
    `feats`: list of float arrays [n_nodes, F0] (one per graph)."""

    def __init__(self, graphs, feats):
        _lib.require_gpu()
        if len(graphs) != len(feats) or not graphs:
            raise ValueError('need one feature matrix per graph')
        ptrs, idxs, ns = [], [], []
        for g in graphs:
            if len(g) == 3:
                n, src, dst = g
                ip, ix = edges_to_in_csr(int(n), src, dst)
            else:
                ip, ix = np.ascontiguousarray(g[0], np.int64), np.ascontiguousarray(g[1], np.int32)
                n = len(ip) - 1
            ptrs.append(ip); idxs.append(ix); ns.append(int(n))
        F0 = int(np.asarray(feats[0]).shape[1])
        fl = []
        for n, f in zip(ns, feats):
            f = np.ascontiguousarray(f, np.float32)
            if f.shape != (n, F0):
                raise ValueError('feature matrix shape %s does not match (%d, %d)' % (f.shape, n, F0))
            fl.append(f)
        G = len(ns)
        n_arr = (C.c_int64 * G)(*ns)
        p_arr = (C.c_void_p * G)(*[a.ctypes.data for a in ptrs])
        i_arr = (C.c_void_p * G)(*[a.ctypes.data for a in idxs])
        f_arr = (C.c_void_p * G)(*[a.ctypes.data for a in fl])
        h = C.c_void_p()
        _lib.check(_lib.lib().gm_store_create(G, n_arr, p_arr, i_arr, f_arr, F0, C.byref(h)), 'gm_store_create')
        self.handle = h
        self.n_graphs, self.n_nodes, self.feat_dim = G, ns, F0
        self.n_edges = [int(p[-1]) for p in ptrs]
        self.host_csr = list(zip(ptrs, idxs))      # host copy of the in-edge CSR (Subgraphs(sample_mode='reference') walks it like sdp.py:301)

    def symmetric(self):
        """True when every out-edge list equals the in-edge list element for element (undirected graphs stored in both directions with
        ascending rows): extraction then walks the adjacency lists once for both CSR orientations (csrc/extract.hip)."""
        for ptr, ix in self.host_csr:
            n = len(ptr) - 1
            dst = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
            order = np.argsort(ix, kind='stable')                      # out-CSR: by source, destinations in in-CSR (= ascending destination) order
            if not (np.array_equal(np.bincount(ix, minlength=n), np.diff(ptr)) and np.array_equal(dst[order], ix.astype(np.int64))):
                return False
        return True

    def close(self):
        if getattr(self, 'handle', None):
            _lib.lib().gm_store_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
