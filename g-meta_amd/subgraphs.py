"""Task / subgraph data: mirror of G-Meta/subgraph_data_processing.py (Subgraphs, collate) with the
h-hop extraction, sampling, induced-subgraph build and batching done by HIP kernels
(gm_extract, include/gmeta_hip.h) on HBM-resident CSR instead of DGL + Python loops."""
import collections
import concurrent.futures
import csv
import ctypes as C
import itertools
import os
import random
import threading

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib
from .graphstore import GraphStore


class _NodeIds(collections.abc.Sequence):
    """Parent node ids of one subgraph (== list(sub.parent_nid), sdp.py:317); fetched from HBM on first use."""

    def __init__(self, owner, k):
        self._o, self._k = owner, k

    def __len__(self):
        off = self._o.sub_off
        return int(off[self._k + 1] - off[self._k])

    def _data(self):
        p, off = self._o.parent(), self._o.sub_off
        return p[off[self._k]:off[self._k + 1]]

    def __getitem__(self, i):
        return self._data()[i]

    def __array__(self, dtype=None, copy=None):
        a = self._data()
        return a.astype(dtype) if dtype is not None else a


class _NodeLists(collections.abc.Sequence):
    """The list of per-subgraph id lists of slots 6/7 (sdp.py:402,408), materialised on access."""

    def __init__(self, owner, a, b):
        self._o, self._a, self._b = owner, a, b

    def __len__(self):
        return self._b - self._a

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return _NodeIds(self._o, self._a + k)


class SubgraphBatch:
    """Device-resident batched induced subgraphs: stands where the reference has a batched DGLGraph
    (slots 0 and 2 of the task tuple, sdp.py:399-408).  `sets` > 1 when it holds several tasks.
    A view (parent, index) addresses one set of a multi-set batch without copying."""

    def __init__(self, handle, store, owner=True, view_of=None, view_index=None):
        self.handle, self.store, self._owner = handle, store, owner
        self.view_of, self.view_index = view_of, view_index
        self._cache = {}
        if view_of is not None:          # a view shares the parent's handle and dimensions
            self.rows, self.edges, self.subs, self.sets, self.centres = view_of.rows, view_of.edges, view_of.subs, view_of.sets, view_of.centres
            return
        rows, edges, subs, sets, cen = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().gm_batch_dims(handle, C.byref(rows), C.byref(edges), C.byref(subs), C.byref(sets), C.byref(cen)))
        self.rows, self.edges, self.subs, self.sets, self.centres = rows.value, edges.value, subs.value, sets.value, cen.value

    # ---- construction
    @staticmethod
    def _seed_array(seeds):
        # gm_seed_t is three packed int32 (graph, i, j): a C-contiguous int32 [n, 3] array has the same layout
        return np.ascontiguousarray(np.asarray(seeds, np.int32).reshape(-1, 3))

    @classmethod
    def extract(cls, store, seeds, set_offsets, h, sample_nodes, rng_seed, link_pred):
        arr = cls._seed_array(seeds)
        so = np.ascontiguousarray(set_offsets, np.int32)
        out = C.c_void_p()
        _lib.check(_lib.lib().gm_extract(store.handle, _lib.ptr(arr), len(arr), _lib.ptr(so), len(so) - 1, int(h), int(sample_nodes),
                                         C.c_uint64(int(rng_seed) & (2 ** 64 - 1)), int(bool(link_pred)), _lib.stream_ptr(), C.byref(out)),
                   'gm_extract')
        b = cls(out, store)
        # what the host already knows is never read back from the device (a gm_batch_read synchronises the batch's stream)
        b._cache[(_lib.F_SET_SUB_OFF,)] = so
        b._cache[(_lib.F_GRAPH,)] = np.ascontiguousarray(arr[:, 0])
        return b

    @classmethod
    def extract_pair(cls, store, seeds_a, set_offsets_a, seeds_b, set_offsets_b, h, sample_nodes, rng_seed, link_pred):
        """The two batches extract() would return for (seeds_a, set_offsets_a) and (seeds_b, set_offsets_b), from one build (gm_extract_pair)."""
        aa, ab = cls._seed_array(seeds_a), cls._seed_array(seeds_b)
        sa, sb = np.ascontiguousarray(set_offsets_a, np.int32), np.ascontiguousarray(set_offsets_b, np.int32)
        oa, ob = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().gm_extract_pair(store.handle, _lib.ptr(aa), len(aa), _lib.ptr(sa), len(sa) - 1, _lib.ptr(ab), len(ab), _lib.ptr(sb), len(sb) - 1,
                                              int(h), int(sample_nodes), C.c_uint64(int(rng_seed) & (2 ** 64 - 1)), int(bool(link_pred)), _lib.stream_ptr(),
                                              C.byref(oa), C.byref(ob)), 'gm_extract_pair')
        out = []
        for h_, arr, so in ((oa, aa, sa), (ob, ab, sb)):
            b = cls(h_, store)
            b._cache[(_lib.F_SET_SUB_OFF,)] = so
            b._cache[(_lib.F_GRAPH,)] = np.ascontiguousarray(arr[:, 0])
            out.append(b)
        return out[0], out[1]

    @classmethod
    def from_nodes(cls, store, seeds, set_offsets, node_lists, link_pred):
        arr = cls._seed_array(seeds)
        so = np.ascontiguousarray(set_offsets, np.int32)
        lists = [np.unique(np.asarray(x, np.int32)) for x in node_lists]
        flat = np.ascontiguousarray(np.concatenate(lists), np.int32)
        off = np.ascontiguousarray(np.cumsum([0] + [len(x) for x in lists]), np.int64)
        out = C.c_void_p()
        _lib.check(_lib.lib().gm_batch_from_nodes(store.handle, _lib.ptr(arr), len(arr), _lib.ptr(so), len(so) - 1, _lib.ptr(flat), _lib.ptr(off),
                                                  int(bool(link_pred)), _lib.stream_ptr(), C.byref(out)), 'gm_batch_from_nodes')
        return cls(out, store)

    @classmethod
    def concat(cls, parts):
        """dgl.batch over batches; if `parts` are the consecutive views of one multi-set batch it is returned as is."""
        p0 = parts[0]
        if p0.view_of is not None and all(p.view_of is p0.view_of and p.view_index == k for k, p in enumerate(parts)) \
                and len(parts) == p0.view_of.sets:
            return p0.view_of
        if len(parts) == 1 and parts[0].view_of is None:
            return parts[0]
        real = [p._materialize() for p in parts]
        arr = (C.c_void_p * len(real))(*[p.handle.value for p in real])
        out = C.c_void_p()
        _lib.check(_lib.lib().gm_batch_concat(arr, len(real), _lib.stream_ptr(), C.byref(out)), 'gm_batch_concat')
        return cls(out, p0.store)

    def views(self):
        return [SubgraphBatch(self.handle, self.store, owner=False, view_of=self, view_index=k) for k in range(self.sets)]

    def _materialize(self):
        if self.view_of is None:
            return self
        raise NotImplementedError('concatenating a strict subset of a multi-set batch is not supported; '
                                  'pass all of its task views, in order')

    # ---- DGL-compatible surface used by meta.py:122 / learner.py:161
    def to(self, device):
        return self

    def _read(self, field, n, dtype):
        if self.view_of is not None:
            return self.view_of._read(field, n, dtype)
        key = (field,)
        if key not in self._cache:
            a = np.empty(n, dtype)
            _lib.check(_lib.lib().gm_batch_read(self.handle, field, _lib.ptr(a), a.nbytes), 'gm_batch_read')
            self._cache[key] = a
        return self._cache[key]

    @property
    def sub_off(self):
        return self._read(_lib.F_SUB_OFF, self.subs + 1, np.int32)

    @property
    def set_sub_off(self):
        return self._read(_lib.F_SET_SUB_OFF, self.sets + 1, np.int32)

    def _sub_range(self):
        if self.view_of is None:
            return 0, self.subs
        o = self.set_sub_off
        return int(o[self.view_index]), int(o[self.view_index + 1])

    @property
    def batch_num_nodes(self):
        a, b = self._sub_range()
        return [int(x) for x in np.diff(self.sub_off)[a:b]]

    def parent(self):
        return self._read(_lib.F_PARENT, self.rows, np.int32)

    def centres_local(self):
        a, b = self._sub_range()
        c = self._read(_lib.F_CENTRE, self.subs * self.centres, np.int32).reshape(self.subs, self.centres)[a:b]
        return c[:, 0] if self.centres == 1 else c

    def graph_ids(self):
        a, b = self._sub_range()
        return self._read(_lib.F_GRAPH, self.subs, np.int32)[a:b]

    def node_lists(self):
        a, b = self._sub_range()
        return _NodeLists(self, a, b)

    def csr(self, transposed=False):
        ip = self._read(_lib.F_INDPTR_T if transposed else _lib.F_INDPTR, self.rows + 1, np.int32)
        ix = self._read(_lib.F_INDICES_T if transposed else _lib.F_INDICES, self.edges, np.int32)
        return ip, ix

    def device_ptr(self, field):
        p = C.c_void_p()
        _lib.check(_lib.lib().gm_batch_device_ptr(self.handle, field, C.byref(p)))
        return p

    def __del__(self):
        try:
            if self._owner and self.handle:
                _lib.lib().gm_batch_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def collate(samples):
    """sdp.py:414-419 / train.py:26-29: list of task tuples -> tuple of 10 lists."""
    return tuple(map(list, zip(*samples)))


class Subgraphs(Dataset):
    """Mirror of subgraph_data_processing.Subgraphs (sdp.py:14-412).  `adjs` is a GraphStore (the
    HBM-resident counterpart of the reference's list of DGLGraph objects).  Task sampling
    (create_batch_*) follows sdp.py:150-292; `tables=` can replace the CSV files by in-memory
    {'train': (names, labels)} style dictionaries (names 'g_i' or 'g_i_j', labels as in the CSV)."""

    def __init__(self, root, mode, subgraph2label, n_way, k_shot, k_query, batchsz, args, adjs, h, tables=None, verbose=True, sample_mode=None):
        self.batchsz, self.n_way, self.k_shot, self.k_query = batchsz, n_way, k_shot, k_query
        # 'device' (default): neighbourhoods above sample_nodes are thinned by the keyed permutation in gm_extract.
        # 'reference': the node sets the REFERENCE would draw for the same global-RNG history (sdp.py:312-314,337-339):
        # the oversize neighbourhood is rebuilt on the host in the reference's list order, passed through a CPython set
        # and np.random.choice exactly like sdp.py:300-313, memoised per name like sdp.py:296-297,319; the GPU then builds
        # the induced subgraphs from those sets (gm_batch_from_nodes).  For reproducing reference runs, not for speed.
        self.sample_mode = sample_mode or getattr(args, 'sample_mode', 'device')
        if self.sample_mode not in ('device', 'reference'):
            raise ValueError("sample_mode must be 'device' or 'reference'")
        self._ref_memo = {}
        self.setsz, self.querysz = n_way * k_shot, n_way * k_query            # sdp.py:20-21
        self.h = h
        self.sample_nodes = args.sample_nodes
        self.rng_seed = int(getattr(args, 'sample_seed', 222))
        if verbose:
            print('shuffle DB :%s, b:%d, %d-way, %d-shot, %d-query, %d-hops' % (mode, batchsz, n_way, k_shot, k_query, h))
        self.subgraph2label = subgraph2label
        self.link_pred_mode = args.link_pred_mode == 'True'                   # string booleans (train.py:175)
        self.task_setup = args.task_setup
        if not isinstance(adjs, GraphStore):
            raise TypeError('adjs must be a gmeta_amd.GraphStore (graphs + features resident in HBM)')
        self.G = adjs
        self.subgraphs = {}     # kept for source compatibility; extraction is not memoised on the host

        def load(name):
            if tables is not None:
                return self._group(*tables[name])
            return self.loadCSV(os.path.join(root, name + '.csv'))
        if self.link_pred_mode:                                               # sdp.py:35-38
            _, dG_s, dGL_s = load(mode + '_spt')
            _, dG_q, dGL_q = load(mode + '_qry')
        dictLabels, dictGraphs, dictGraphsLabels = load(mode)
        if self.task_setup == 'Disjoint':                                     # sdp.py:51-58
            self.data = [v for v in dictLabels.values()]
            self.cls_num = len(self.data)
            self.create_batch_disjoint(batchsz)
        elif self.task_setup == 'Shared':
            if self.link_pred_mode:                                           # sdp.py:61-95
                self.data_label_spt = [list(dGL_s[k].values()) for k in dG_s]
                self.data_label_qry = [list(dGL_q[k].values()) for k in dG_q]
                self.graph_num_spt = len(self.data_label_spt)
                self.create_batch_LinkPred(batchsz)
            else:                                                             # sdp.py:97-116
                self.data_label = [list(dictGraphsLabels[k].values()) for k in dictGraphs]
                self.graph_num = len(self.data_label)
                self.cls_num = len(self.data_label[0])
                self.create_batch_shared(batchsz)
        else:
            raise ValueError("task_setup must be 'Disjoint' or 'Shared'")
        self._task_memo = {}          # per-task seed / raw-label arrays (the names never change after create_batch_*)
        # The name -> (graph, i, j) parsing of sdp.py:355-362 is pure Python (~3 ms per 32-task meta-batch): done here, once, with the task
        # lists -- in the prefetch thread it would hold the GIL against the training thread between two meta-steps
        if len(self.support_x_batch) <= 65536:
            for i in range(len(self.support_x_batch)):
                self._task_arrays(i)

    # ---- CSV index (sdp.py:119-148): columns (pandas index, name, label); label kept as string
    @staticmethod
    def _group(names, labels):
        dictGraphsLabels, dictLabels, dictGraphs = {}, {}, {}
        for filename, label in zip(names, labels):
            label = str(label)
            g_idx = int(filename.split('_')[0])
            dictGraphs.setdefault(g_idx, []).append(filename)
            dictGraphsLabels.setdefault(g_idx, {}).setdefault(label, []).append(filename)
            dictLabels.setdefault(label, []).append(filename)
        return dictLabels, dictGraphs, dictGraphsLabels

    def loadCSV(self, csvf):
        names, labels = [], []
        with open(csvf) as f:
            rd = csv.reader(f, delimiter=',')
            next(rd, None)
            for row in rd:
                names.append(row[1]); labels.append(row[2])
        return self._group(names, labels)

    # ---- task sampling (host bookkeeping on names; numpy/python global RNGs like the reference)
    def _pick(self, pool, k_shot, k_query):
        idx = np.random.choice(len(pool), k_shot + k_query, False)
        np.random.shuffle(idx)
        arr = np.array(pool)
        return arr[idx[:k_shot]].tolist(), arr[idx[k_shot:]].tolist()

    def create_batch_disjoint(self, batchsz):                                 # sdp.py:150-182
        self.support_x_batch, self.query_x_batch = [], []
        for _ in range(batchsz):
            selected_cls = np.random.choice(self.cls_num, self.n_way, False)
            np.random.shuffle(selected_cls)
            support_x, query_x = [], []
            for cls in selected_cls:
                s, q = self._pick(self.data[cls], self.k_shot, self.k_query)
                support_x.append(s); query_x.append(q)
            random.shuffle(support_x); random.shuffle(query_x)
            self.support_x_batch.append(support_x); self.query_x_batch.append(query_x)

    def create_batch_shared(self, batchsz):                                   # sdp.py:184-247
        self.support_x_batch, self.query_x_batch = [], []
        for _ in range(batchsz):
            data = self.data_label[np.random.choice(self.graph_num, 1, False)[0]]
            selected_cls = np.arange(len(data)); np.random.shuffle(selected_cls)
            support_x, query_x = [], []
            for cls in selected_cls:
                pool = data[cls]
                if len(pool) >= self.k_shot + self.k_query:
                    s, q = self._pick(pool, self.k_shot, self.k_query)
                elif len(pool) >= self.k_shot:
                    # the reference's short-class branch (sdp.py:218-238, "not used in practice"): every entity of the class is used, the first
                    # k_shot of a shuffle as support, and the query list is topped up with k_shot + k_query - len + 1 entities (`count <=
                    # num_more`) of randomly chosen classes of the same graph, drawn with the same global-RNG calls.  Such a task carries
                    # k_query + 1 query entries with mixed labels: the query loss then raises on unequal class counts, in the reference
                    # (torch.stack, meta.py:65) and here (gm_meta_step) alike.
                    idx = np.arange(len(pool)); np.random.shuffle(idx)
                    arr = np.array(pool)
                    s, q = arr[idx[:self.k_shot]].tolist(), arr[idx[self.k_shot:]].tolist()
                    for _ in range(self.k_shot + self.k_query - len(pool) + 1):
                        sub_cls = np.random.choice(selected_cls, 1)[0]
                        q.append(str(np.array(data[sub_cls])[np.random.choice(len(data[sub_cls]), 1)[0]]))
                else:
                    print('each class in a graph must have larger than k_shot entities in the current model')      # sdp.py:240: class skipped
                    continue
                support_x.append(s); query_x.append(q)
            random.shuffle(support_x); random.shuffle(query_x)
            self.support_x_batch.append(support_x); self.query_x_batch.append(query_x)

    def create_batch_LinkPred(self, batchsz):                                 # sdp.py:249-292
        self.support_x_batch, self.query_x_batch = [], []
        for _ in range(batchsz):
            g = np.random.choice(self.graph_num_spt, 1, False)[0]
            data_spt, data_qry = self.data_label_spt[g], self.data_label_qry[g]
            cs = np.arange(len(data_spt)); np.random.shuffle(cs)
            cq = np.arange(len(data_qry)); np.random.shuffle(cq)
            support_x, query_x = [], []
            for cls in cs:
                idx = np.random.choice(len(data_spt[cls]), self.k_shot, False); np.random.shuffle(idx)
                support_x.append(np.array(data_spt[cls])[idx].tolist())
            for cls in cq:
                idx = np.random.choice(len(data_qry[cls]), self.k_query, False); np.random.shuffle(idx)
                query_x.append(np.array(data_qry[cls])[idx].tolist())
            random.shuffle(support_x); random.shuffle(query_x)
            self.support_x_batch.append(support_x); self.query_x_batch.append(query_x)

    # ---- extraction
    _seed_memo = {}

    @classmethod
    def _seeds(cls, names):
        memo = cls._seed_memo          # 'g_i' / 'g_i_j' -> (g, i, j): names recur across tasks and epochs
        out = []
        for item in names:
            v = memo.get(item)
            if v is None:
                p = [int(x) for x in item.split('_')]
                v = memo[item] = tuple(p + [-1] if len(p) == 2 else p)
            out.append(v)
        return np.array(out, np.int32).reshape(-1, 3)

    def _task_names(self, index):
        spt = [item for sub in self.support_x_batch[index] for item in sub]
        qry = [item for sub in self.query_x_batch[index] for item in sub]
        return spt, qry

    def _task_arrays(self, index):
        """Per-task host tables, computed once: seeds int32 [n,3] and raw int labels of the support and query names."""
        c = self._task_memo.get(index)
        if c is None:
            spt, qry = self._task_names(index)
            lab = self.subgraph2label
            c = self._task_memo[index] = (self._seeds(spt), self._seeds(qry), np.array([lab[i] for i in spt]).astype(np.int32),
                                          np.array([lab[i] for i in qry]).astype(np.int32))
        return c

    @staticmethod
    def _labels_lists(support_y, query_y):
        """sdp.py:389-397 on plain lists: a handful of labels per task, where plain Python beats five numpy calls (this runs per task between two
        meta-steps, under the GIL)."""
        sl, ql = support_y.tolist(), query_y.tolist()
        unique = sorted(set(sl))                                              # np.unique(support_y)
        # random.shuffle(unique) of the reference, applied to an index list: the same draws, the same permutation
        order = list(range(len(unique)))
        random.shuffle(order)
        rank = {unique[o]: idx for idx, o in enumerate(order)}                # class unique[order[idx]] -> idx
        get = rank.get
        return [rank[c] for c in sl], [get(c, 0) for c in ql]                 # a query class absent from the support keeps 0 (np.zeros, sdp.py:393)

    def _labels(self, support_y, query_y):
        if self.task_setup == 'Disjoint':                                     # sdp.py:389-397
            ys, yq = self._labels_lists(support_y, query_y)
            return torch.tensor(ys, dtype=torch.int64), torch.tensor(yq, dtype=torch.int64)
        return torch.from_numpy(support_y.astype(np.int64)), torch.from_numpy(query_y.astype(np.int64))

    def _tuple(self, bs, bq, ys, yq):
        return (bs, ys, bq, yq, torch.from_numpy(bs.centres_local().astype(np.int64)), torch.from_numpy(bq.centres_local().astype(np.int64)),
                bs.node_lists(), bq.node_lists(), bs.graph_ids().tolist(), bq.graph_ids().tolist())

    # ---- reference-order sampling replay (sample_mode='reference')
    def _in(self, g, v):
        ip, ix = self.G.host_csr[g]
        return ix[ip[v]:ip[v + 1]].tolist()                                   # [n.item() for n in G.in_edges(v)[0]] (sdp.py:301)

    def _reference_nodes(self, name, g, i, j):
        """The sorted node array the reference keeps for an OVERSIZE neighbourhood; consumes np.random like sdp.py:313."""
        if name in self._ref_memo:
            return self._ref_memo[name]
        chain = itertools.chain
        if self.link_pred_mode:                                               # sdp.py:327-335 (j side: 1 hop, the sdp.py:332 quirk)
            f_hop = self._in(g, i)
            n_l = [self._in(g, u) for u in f_hop]
            a1 = np.array(list(set([x for sub in n_l for x in sub] + f_hop + [i])), np.int64)
            f_hop = self._in(g, j)
            n_l = [self._in(g, j) for _ in f_hop]
            a2 = np.array(list(set([x for sub in n_l for x in sub] + f_hop + [j])), np.int64)
            arr, keep = np.union1d(a1, a2), [i, j]
        else:
            f_hop = self._in(g, i)
            if self.h == 1:                                                   # sdp.py:304-306
                arr = np.array(list(set(f_hop + [i])), np.int64)
            elif self.h == 2:                                                 # sdp.py:300-303
                n_l = [self._in(g, u) for u in f_hop]
                arr = np.array(list(set(list(chain(*n_l)) + f_hop + [i])), np.int64)
            elif self.h == 3:                                                 # sdp.py:307-311
                n_2 = [self._in(g, u) for u in f_hop]
                n_3 = [self._in(g, u) for u in list(chain(*n_2))]
                arr = np.array(list(set(list(chain(*n_2)) + list(chain(*n_3)) + f_hop + [i])), np.int64)
            else:
                raise ValueError('the reference defines h in {1, 2, 3} (sdp.py:300-311)')
            keep = [i]
        if arr.shape[0] <= self.sample_nodes:
            raise RuntimeError('reference replay: %s is not oversize on the host (%d nodes) but was on the device' % (name, arr.shape[0]))
        arr = np.random.choice(arr, self.sample_nodes, replace=False)         # sdp.py:313 / 338 -- the global numpy RNG
        arr = np.unique(np.append(arr, keep)).astype(np.int32)                # sdp.py:314 / 339
        self._ref_memo[name] = arr
        return arr

    def _extract_reference(self, tasks):
        """tasks: [(spt_seeds, spt_names, qry_seeds, qry_names)] in visiting order.  Returns (S, Q) batches holding the
        reference's node sets; the RNG is consumed task by task, support items before query items, like __getitem__
        (sdp.py:363-386), once per name."""
        seeds = np.concatenate([np.concatenate([t[0], t[2]]) for t in tasks])
        names = [n for t in tasks for n in (list(t[1]) + list(t[3]))]
        # sizing pass: with a threshold of sample_nodes + 1 every neighbourhood the reference would sample comes back with more than
        # sample_nodes nodes (thinned or not) and every other one comes back exact -- without per-subgraph buffers of graph size
        full = SubgraphBatch.extract(self.G, seeds, [0, len(seeds)], self.h, self.sample_nodes + 1, self.rng_seed, self.link_pred_mode)
        off, par = full.sub_off, full.parent()
        lists = []
        for k, (name, (g, i, j)) in enumerate(zip(names, seeds.tolist())):
            if off[k + 1] - off[k] > self.sample_nodes:
                lists.append(self._reference_nodes(name, g, i, j))
            else:
                lists.append(par[off[k]:off[k + 1]])
        pos, ls, lq = 0, [], []
        for t in tasks:
            ls += lists[pos:pos + len(t[0])]; pos += len(t[0])
            lq += lists[pos:pos + len(t[2])]; pos += len(t[2])
        off_s = np.cumsum([0] + [len(t[0]) for t in tasks]); off_q = np.cumsum([0] + [len(t[2]) for t in tasks])
        S = SubgraphBatch.from_nodes(self.G, np.concatenate([t[0] for t in tasks]), off_s, ls, self.link_pred_mode)
        Q = SubgraphBatch.from_nodes(self.G, np.concatenate([t[2] for t in tasks]), off_q, lq, self.link_pred_mode)
        return S, Q

    # The support and the query batch of a meta-batch are independent builds (two gm_extract calls, each with two host round trips and ~0.1-0.3 ms
    # of host-side table work between its kernels): the support batch is built by a helper thread on a stream of its own while the calling thread
    # builds the query batch -- ctypes releases the GIL for the length of the call.  The caller's stream then waits for the helper stream's event, so
    # consumers see both batches complete; gm_batch_destroy orders its frees behind the consumers' work (gm_batch_mark_use) as for prefetched batches.
    # One helper per calling thread (the training thread and every prefetch worker have their own).  (GMETA_EXTRACT_MODE=threads; the default since is one
    # joint build of both batches, gm_extract_pair.)
    _tls = threading.local()

    def _helper(self):
        h = getattr(self._tls, 'helper', None)
        dev = torch.cuda.current_device()
        if h is None or h[2] != dev:
            ex = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='gmeta-extract')
            h = self._tls.helper = (ex, ex.submit(self._helper_init, dev).result(), dev)
        return h

    @staticmethod
    def _helper_init(dev):
        torch.cuda.set_device(dev)
        return torch.cuda.Stream(priority=Subgraphs._PREFETCH_PRIORITY)

    def _extract_on(self, stream, seeds, off):
        with torch.cuda.stream(stream):
            b = SubgraphBatch.extract(self.G, seeds, off, self.h, self.sample_nodes, self.rng_seed, self.link_pred_mode)
            ev = torch.cuda.Event()
            ev.record(stream)
        return b, ev

    @staticmethod
    def _pack_seeds(arrs):
        """Seed tables and set offsets of the support / query batch of a meta-batch (the arguments of gm_extract / gm_extract_pair)."""
        off_s = np.cumsum([0] + [len(a[0]) for a in arrs]); off_q = np.cumsum([0] + [len(a[1]) for a in arrs])
        return np.concatenate([a[0] for a in arrs]), off_s, np.concatenate([a[1] for a in arrs]), off_q

    def _extract_tasks(self, indices, arrs=None, packed=None):
        if arrs is None:
            arrs = [self._task_arrays(i) for i in indices]
        if self.sample_mode == 'reference':
            names = [self._task_names(i) for i in indices]
            S, Q = self._extract_reference([(a[0], n[0], a[1], n[1]) for a, n in zip(arrs, names)])
            return arrs, S, Q
        seeds_s, off_s, seeds_q, off_q = packed if packed is not None else self._pack_seeds(arrs)
        # default: ONE build for both batches (gm_extract_pair: one launch of each extraction kernel over all subgraphs, one round trip for both finalisations);
        # GMETA_EXTRACT_MODE=threads: two gm_extract calls, the support batch on a helper thread / stream; =serial: one after the other
        mode = os.environ.get('GMETA_EXTRACT_MODE', 'pair')
        if mode == 'pair':
            S, Q = SubgraphBatch.extract_pair(self.G, seeds_s, off_s, seeds_q, off_q, self.h, self.sample_nodes, self.rng_seed, self.link_pred_mode)
            return arrs, S, Q
        helper = self._helper() if (mode == 'threads' and len(seeds_q) >= 64) else None
        if helper is None:
            S = SubgraphBatch.extract(self.G, seeds_s, off_s, self.h, self.sample_nodes, self.rng_seed, self.link_pred_mode)
            Q = SubgraphBatch.extract(self.G, seeds_q, off_q, self.h, self.sample_nodes, self.rng_seed, self.link_pred_mode)
            return arrs, S, Q
        fut = helper[0].submit(self._extract_on, helper[1], seeds_s, off_s)
        try:
            Q = SubgraphBatch.extract(self.G, seeds_q, off_q, self.h, self.sample_nodes, self.rng_seed, self.link_pred_mode)
        finally:
            S, ev = fut.result()                                              # (also re-raises the helper's failure)
        torch.cuda.current_stream().wait_event(ev)
        return arrs, S, Q

    def __getitem__(self, index):
        """One task (sdp.py:348-408): the 10-tuple with SubgraphBatch handles in slots 0 and 2."""
        arrs, bs, bq = self._extract_tasks([index])
        ys, yq = self._labels(arrs[0][2], arrs[0][3])
        return self._tuple(bs, bq, ys, yq)

    def get_batch(self, indices):
        """MI355X-first counterpart of DataLoader(..., collate_fn=collate): the subgraphs of ALL tasks of a
        meta-batch are extracted by two launches (support / query); returns the collated 10-tuple of lists."""
        if len(indices) == 0:           # an empty task shard (more ranks than tasks in a short trailing meta-batch)
            return tuple([] for _ in range(10))
        return self._build(self._prepare(indices))

    def _prepare(self, indices):
        """Host half of get_batch that consumes the global Python RNG (the per-task label shuffle of sdp.py:390-397): per-task arrays and
        relabelled targets, in task order.  batches(workers > 1) runs it in the consumer's thread, in meta-batch order, so that the draws do
        not depend on which worker builds which meta-batch."""
        arrs = [self._task_arrays(i) for i in indices]
        if self.task_setup == 'Disjoint' and len({(len(a[2]), len(a[3])) for a in arrs}) == 1:
            # the usual case, every task of the same shape: the relabelled targets of all tasks as two tensors, one row view per task (a torch.tensor per
            # task and side was a third of this function)
            sy, qy = [], []
            for a in arrs:
                ys, yq = self._labels_lists(a[2], a[3])
                sy.append(ys); qy.append(yq)
            sy, qy = torch.tensor(sy, dtype=torch.int64), torch.tensor(qy, dtype=torch.int64)
            labels = list(zip(sy.unbind(0), qy.unbind(0)))
        else:
            labels = [self._labels(a[2], a[3]) for a in arrs]
        # (the packed seed tables too: a builder thread of batches() then goes from its job's first line straight into the library)
        return list(indices), arrs, labels, None if self.sample_mode == 'reference' else self._pack_seeds(arrs)

    def _build(self, prep):
        indices, arrs, labels, packed = prep
        arrs, S, Q = self._extract_tasks(indices, arrs, packed)
        # the ten slots of every task (what _tuple builds one view at a time), from ONE read of each batch's centre table and host-side slices
        cols = [S.views(), [y[0] for y in labels], Q.views(), [y[1] for y in labels], None, None, None, None, None, None]
        for b, k in ((S, 0), (Q, 1)):
            off = b.set_sub_off.tolist()
            cen = torch.from_numpy(b._read(_lib.F_CENTRE, b.subs * b.centres, np.int32).astype(np.int64).reshape(b.subs, b.centres))
            if b.centres == 1:
                cen = cen[:, 0]
            gid = b._read(_lib.F_GRAPH, b.subs, np.int32).tolist()
            cols[4 + k] = [cen[off[t]:off[t + 1]] for t in range(b.sets)]
            cols[6 + k] = [_NodeLists(b, off[t], off[t + 1]) for t in range(b.sets)]
            cols[8 + k] = [gid[off[t]:off[t + 1]] for t in range(b.sets)]
        return tuple(cols)

    # The builder threads' streams are created ONCE per (device, builder slot, priority) and kept for the life of the process.  HIP deals streams of one
    # priority round-robin onto a small set of hardware queues (GPU_MAX_HW_QUEUES, 4), so which queue a NEW stream gets depends on how many streams the
    # process has created before; every batches() call starts new threads, and a stream per call meant that sooner or later a builder landed on the
    # hardware queue of the meta-step's own stream and its build serialised with the step (bench.py after its builder-pool phases, a training run in
    # its later epochs: 4.77 instead of 4.13 ms per 4-task meta-step with a build per step).  Builder slot 0 is the third stream of a typical process
    # (after the caller's and the meta-step's query stream): a queue of its own.  GMETA_BUILD_PRIORITY (torch convention: lower = more urgent; default
    # 0): -1 also gives the builders queues of their own, but their wide kernels then run ahead of a short meta-step's small ones (receptive-field
    # schedule, build per step: 2.90 instead of 2.6-2.7 ms).
    _PREFETCH_PRIORITY = int(os.environ.get('GMETA_BUILD_PRIORITY', '0'))
    _builder_streams = {}
    _builder_streams_lock = threading.Lock()

    @classmethod
    def _builder_stream(cls, dev, slot, priority):
        key = (int(dev), int(slot), int(priority))
        with cls._builder_streams_lock:
            st = cls._builder_streams.get(key)
            if st is None:
                torch.cuda.set_device(dev)
                st = cls._builder_streams[key] = torch.cuda.Stream(device=dev, priority=int(priority))
            return st

    def batches(self, index_lists, prefetch=1, cone_layers=0, priority=None, workers=1):
        """Iterate get_batch(idx) for idx in index_lists with the NEXT `prefetch` meta-batches being prepared (host half: one thread) and
        extracted (a builder thread on its own HIP stream) while the caller runs the meta-step on the current one -- what
        DataLoader(num_workers>0) does for the reference (train.py:96,173), minus the per-worker copy of the memo cache
        (sdp.py:296-297,319).  cone_layers = n_gcn also builds the receptive-field tables (gm_hparams_t.cone) there.
        workers > 1: that many builder threads (each with its own stream), meta-batches delivered in order -- for schedules whose
        meta-step is shorter than one batch build (the receptive-field schedule at task_num 32: 2.2 ms against ~3 ms)."""
        index_lists = [list(int(i) for i in idx) for idx in index_lists]
        if prefetch <= 0 or len(index_lists) <= 1:
            for idx in index_lists:
                yield self.get_batch(idx)
            return
        yield from self._batches_pool(index_lists, max(prefetch, workers), cone_layers, priority, max(1, workers))

    def _batches_pool(self, index_lists, depth, cone_layers, priority, workers):
        """Three stages, meta-batches delivered in order: ONE thread runs the host halves (_prepare: task arrays, the label shuffles -- every draw of
        the global Python RNG, in meta-batch order, whichever builder takes the batch), `workers` builder threads (own streams) run the GPU builds
        (_build + the receptive-field tables), the caller consumes.
        Who starts what matters more than how long it takes (tools/e2e_timeline.py, receptive-field schedule at task_num 32: a 2.2 ms meta-step):
        the interpreter lock is handed over only when its holder blocks, so a host half started by the CALLER's next() ran its 0.4 ms of pure Python
        exactly while the caller wanted to get to its meta-step and the builder to its build -- both waited for all of it.  (A builder that is faster
        than the caller idles until the caller's next() queues a job: it is in step with the caller, and so is whatever it starts at the top of a job.)
        Here a host half is started by a BUILDER when its batches exist and it goes back into the library for the tables / the final wait -- the
        caller is inside its meta-step then -- `depth` + 1 meta-batches ahead of the build it feeds; the caller's next() only queues closures."""
        dev = torch.cuda.current_device()
        tls = threading.local()
        n = len(index_lists)
        slots = [[threading.Event(), None] for _ in range(n)]        # host half k: done flag, result / exception
        go = threading.Semaphore(depth + 1)                           # host halves allowed to start
        stop = threading.Event()

        trash = collections.deque()              # delivered meta-batches the caller has let go of: taken apart HERE, not in the caller's loop

        def host():
            for k, idx in enumerate(index_lists):
                go.acquire()
                if stop.is_set():
                    break
                while trash:
                    trash.popleft()
                try:
                    slots[k][1] = ('ok', self._prepare(idx) if idx else ([], [], [], None))
                except BaseException as e:      # surfaces in the builder that takes this meta-batch, then in the caller
                    slots[k][1] = ('err', e)
                slots[k][0].set()

        def job(k):
            slots[k][0].wait()
            kind, prep = slots[k][1]
            slots[k][1] = None
            if kind == 'err':
                raise prep
            if not prep[0]:
                go.release()
                return tuple([] for _ in range(10))
            side = getattr(tls, 'side', None)
            if side is None:
                torch.cuda.set_device(dev)
                with slot_lock:
                    slot = slot_next[0]; slot_next[0] += 1
                side = tls.side = self._builder_stream(dev, slot, self._PREFETCH_PRIORITY if priority is None else int(priority))
            with torch.cuda.stream(side):
                b = self._build(prep)
                go.release()                 # the next host half starts while this thread is inside the library again (tables / the wait below)
                if cone_layers:
                    roots = [x.view_of if x.view_of is not None else x for x in (b[0][0], b[2][0])]
                    _lib.check(_lib.lib().gm_batch_prepare_cone_pair(roots[0].handle, roots[1].handle, int(cone_layers), _lib.stream_ptr()), 'gm_batch_prepare_cone_pair')
                side.synchronize()
            return b

        th = threading.Thread(target=host, name='gmeta-prepare', daemon=True)
        th.start()
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix='gmeta-batches')
        pending = collections.deque()
        held = collections.deque()
        nxt = [0]
        slot_next, slot_lock = [0], threading.Lock()

        def top_up():
            while len(pending) < depth and nxt[0] < n:
                pending.append(pool.submit(job, nxt[0]))
                nxt[0] += 1
        try:
            while True:
                top_up()
                if not pending:
                    break
                f = pending.popleft()
                top_up()                    # `depth` builds queued or running while the caller works on this one
                # The last reference to a meta-batch decides which thread takes it apart (64 views, two batches, their tables, the events of their
                # slabs).  The caller drops its reference to batch k when it takes batch k + 1 -- in the same breath in which it queues the next
                # build, i.e. while a builder thread wants the interpreter lock for the top of its job: 0.2 ms per step of the receptive-field loop.
                # This generator keeps a reference until the caller asks for batch k + 2 and then leaves it to the host thread, which runs while
                # the caller is inside its meta-step: one more meta-batch alive in HBM.
                held.append(f.result())
                del f
                if len(held) > 2:
                    trash.append(held.popleft())
                yield held[-1]
        finally:
            held.clear(); trash.clear()
            stop.set()
            for f in pending:
                f.cancel()
            for _ in range(n + 1):
                go.release()
            pool.shutdown(wait=True)
            th.join()

    def __len__(self):
        return self.batchsz
