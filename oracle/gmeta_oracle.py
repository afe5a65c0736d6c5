"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float32) of G-Meta's inner-loop hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / reported baseline.  The product (g-meta_amd/) never imports it.

PINNING.  Every function below is checked against tests/golden/*.npz, which were produced in
the build container by running the reference's own learner.py / meta.py /
subgraph_data_processing.py UNMODIFIED (oracle/make_golden.py).  The reference delegates its
graph primitives to the third-party package dgl==0.4.3post2 (requirements.txt:2), which is NOT
under /root/reference and is not installable here; its published semantics are restated in
oracle/dgl_shim and pinned by known-answer tests (tests/test_dgl_restatement.py).  So:
  * reference-owned arithmetic (GraphConv normalisation/ordering, centre gather, linear head,
    prototypical losses, first-order ProtoMAML step, Adam): PINNED by the reference's outputs;
  * the DGL boundary (update_all copy_src/sum, in_edges, subgraph, batch): parity UNPINNED by
    the reference (it ships no tests/vectors); restated + known-answer tested by us.

File:line citations are relative to /root/reference/G-Meta/ (sdp = subgraph_data_processing.py).

Two deliberate, documented deviations from the reference:
  * node order inside a subgraph is ASCENDING parent id (the reference uses CPython set order
    for unsampled node-clf subgraphs, sdp.py:303; sampled/link-pred ones are already ascending,
    sdp.py:314,335,339).  Node/edge SETS are compared bit-exactly; logits are order-invariant
    up to fp summation order.
  * the node sampler (sdp.py:312-314,337-339 use the global numpy RNG) is a counter-based
    keyed permutation (`sample_keys`): uniform without replacement, reproducible per
    (seed, graph, i, j), identical on CPU and GPU.  `replay` entry points accept the reference's
    own sampled node lists so downstream float parity is still checked on sampled tasks.
"""
import ctypes
import os
import numpy as np

f32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))


# =========================================================================== graph containers
class Graph:
    """Parent graph as in-edge CSR (by destination).  Replaces dgl.DGLGraph (train.py:43-44)."""

    def __init__(self, n, src, dst):
        src = np.asarray(src, np.int64); dst = np.asarray(dst, np.int64)
        order = np.argsort(dst, kind='stable')          # keep parent edge-id order inside a row
        self.n = int(n)
        self.indices = src[order].astype(np.int32)
        self.indptr = np.zeros(self.n + 1, np.int64)
        np.add.at(self.indptr, dst + 1, 1)
        self.indptr = np.cumsum(self.indptr)

    def preds(self, v):
        """G.in_edges(v)[0] (sdp.py:301): sources of all edges into v, duplicates kept."""
        return self.indices[self.indptr[v]:self.indptr[v + 1]]


# =========================================================================== extraction (a1, a2)
def khop_nodes(G, i, h):
    """sdp.py:300-311: {i} U 1..h-step in-predecessors, as a sorted unique array.
    The reference unions the exactly-k-step predecessor lists; that equals BFS distance <= h."""
    seen = np.zeros(G.n, bool); seen[i] = True
    frontier = np.array([i], np.int64)
    for _ in range(h):
        nxt = np.unique(np.concatenate([G.preds(v) for v in frontier]) if len(frontier) else np.zeros(0, np.int64))
        nxt = nxt[~seen[nxt]]
        seen[nxt] = True
        frontier = nxt
    return np.nonzero(seen)[0].astype(np.int32)


def linkpred_nodes(G, i, j):
    """sdp.py:327-335 INCLUDING the reference bug at sdp.py:332 (`G.in_edges(j)` inside
    `for i in f_hop`): i side is 2 hops, j side is only {j} U preds(j); --h is ignored."""
    a = khop_nodes(G, i, 2)
    b = np.union1d(G.preds(j), [j])
    return np.union1d(a, b).astype(np.int32)


def lowbias32(x):
    x = np.asarray(x, np.uint32).copy()
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def sample_salt(seed, g, i, j):
    """Per-subgraph salt of the keyed permutation; 32-bit ops only (mirrors the HIP kernel)."""
    with np.errstate(over='ignore'):
        s = lowbias32(np.uint32(seed & 0xffffffff) ^ np.uint32(0x9E3779B9))
        s = lowbias32(s ^ np.uint32((seed >> 32) & 0xffffffff))
        s = lowbias32(s + np.uint32(g) * np.uint32(0x85EBCA6B))
        s = lowbias32(s ^ np.uint32(i))
        s = lowbias32(s + np.uint32(j + 1) * np.uint32(0xC2B2AE35))
    return np.uint32(s)


def sample_keys(nodes, salt):
    """key(node) = lowbias32(node ^ salt): a bijection of node for a fixed salt => no ties."""
    return lowbias32(np.asarray(nodes, np.uint32) ^ np.uint32(salt))


def sample_nodes(nodes, k, seed, g, i, j=-1):
    """Build-owned replacement for np.random.choice(nodes, k, replace=False) followed by
    np.unique(np.append(., centres)) (sdp.py:312-314 / 337-339): keep the k nodes with the
    smallest keys, then add the centre(s); result ascending, size k, k+1 (or k+2)."""
    nodes = np.asarray(nodes, np.int32)
    if len(nodes) <= k:                                   # strict '>' at sdp.py:312,337
        return nodes
    keys = sample_keys(nodes, sample_salt(seed, g, i, j))
    keep = nodes[np.argsort(keys, kind='stable')[:k]]
    cen = [i] if j < 0 else [i, j]
    return np.unique(np.append(keep, cen)).astype(np.int32)


def induce(G, nodes):
    """G.subgraph(nodes) (sdp.py:316,341) for ASCENDING `nodes`: local in-edge CSR; row r lists
    local sources of every parent edge into nodes[r] whose source is inside, parent order kept."""
    nodes = np.asarray(nodes, np.int64)
    lut = np.full(G.n, -1, np.int64); lut[nodes] = np.arange(len(nodes))
    indptr = np.zeros(len(nodes) + 1, np.int64); cols = []
    for r, v in enumerate(nodes):
        s = lut[G.preds(v)]
        s = s[s >= 0]
        cols.append(s); indptr[r + 1] = indptr[r] + len(s)
    indices = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
    return indptr, indices


class Batch:
    """One batched set of subgraphs == dgl.batch(list) (sdp.py:399-406) + everything
    learner.py/meta.py derive from it.  Rows are numbered consecutively across subgraphs."""

    def __init__(self, graphs, seeds, node_lists):
        """seeds [S,3] (graph, i, j|-1); node_lists: ascending parent ids per subgraph."""
        self.S = len(seeds)
        self.link = bool(len(seeds) and seeds[0][2] >= 0)
        ptr, idx, par, gid, cen = [np.zeros(1, np.int64)], [], [], [], []
        off, eoff = 0, 0
        self.sub_off = np.zeros(self.S + 1, np.int64)
        for s, ((g, i, j), nodes) in enumerate(zip(seeds, node_lists)):
            nodes = np.asarray(nodes, np.int64)
            ip, ix = induce(graphs[g], nodes)
            ptr.append(ip[1:] + eoff); idx.append(ix.astype(np.int64) + off)
            par.append(nodes); gid.append(np.full(len(nodes), g, np.int64))
            ci = off + np.searchsorted(nodes, i)
            cen.append([ci, off + np.searchsorted(nodes, j)] if j >= 0 else [ci])
            off += len(nodes); eoff += len(ix)
            self.sub_off[s + 1] = off
        self.n = off
        self.indptr = np.concatenate(ptr)
        self.indices = np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)
        self.parent = np.concatenate(par) if par else np.zeros(0, np.int64)
        self.graph_id = np.concatenate(gid) if gid else np.zeros(0, np.int64)
        self.centre_rows = np.array(cen, np.int64)          # [S,1] or [S,2] (learner.py:165-170)
        deg = np.diff(self.indptr)                          # in-degree of the batched induced graph
        self.norm = np.power(np.maximum(deg, 1).astype(f32), f32(-0.5)).astype(f32)  # learner.py:29
        self.dst = np.repeat(np.arange(self.n), deg)        # edge list view for the transposed pass

    def features(self, feats):
        """meta.py:119-120 feature gather: vstack(feat[g][ids])."""
        out = np.empty((self.n, feats[0].shape[1]), f32)
        for g in np.unique(self.graph_id):
            m = self.graph_id == g
            out[m] = feats[g][self.parent[m]]
        return out


def extract_batch(graphs, seeds, h, sample_n, rng_seed, link_pred, replay_nodes=None):
    """a1+a2+a3 for one set: node sets (optionally replayed from the reference) -> Batch."""
    lists = []
    for s, (g, i, j) in enumerate(seeds):
        if replay_nodes is not None:
            nodes = np.unique(np.asarray(replay_nodes[s], np.int32))
        else:
            full = linkpred_nodes(graphs[g], i, j) if link_pred else khop_nodes(graphs[g], i, h)
            nodes = sample_nodes(full, sample_n, rng_seed, g, i, j if link_pred else -1)
        lists.append(nodes)
    return Batch(graphs, seeds, lists)


# =========================================================================== aggregate (a6/a12)
_clib = None


def _load_c():
    global _clib
    if _clib is None:
        p = os.path.join(_HERE, '_build', 'liboracle_kernels.so')
        if os.path.exists(p):
            _clib = ctypes.CDLL(p)
        else:
            _clib = False
    return _clib


def agg(indptr, indices, x):
    """update_all(copy_src, sum) (learner.py:38-39,44-45): out[v] = sum_{u in row v} x[u]."""
    lib = _load_c()
    x = np.ascontiguousarray(x, f32)
    n = len(indptr) - 1
    out = np.zeros((n, x.shape[1]), f32)
    if lib:
        ip = np.ascontiguousarray(indptr, np.int64); ix = np.ascontiguousarray(indices, np.int64)
        lib.oracle_agg_f32(ctypes.c_int64(n), ctypes.c_int64(x.shape[1]), ip.ctypes.data_as(ctypes.c_void_p),
                           ix.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p))
        return out
    deg = np.diff(indptr)
    np.add.at(out, np.repeat(np.arange(n), deg), x[indices])
    return out


def _by_source(batch):
    """The batch's edges grouped by source (stable: destinations keep their edge order); built once per batch."""
    c = getattr(batch, '_csr_t', None)
    if c is None:
        order = np.argsort(batch.indices, kind='stable')
        ptr = np.zeros(batch.n + 1, np.int64)
        np.add.at(ptr, np.asarray(batch.indices, np.int64) + 1, 1)
        c = batch._csr_t = (np.cumsum(ptr), np.ascontiguousarray(batch.dst[order], np.int64))
    return c


def agg_t(batch, g):
    """autograd backward of update_all: grad_x[u] = sum_{(u->v)} g[v] (transposed pass)."""
    lib = _load_c()
    if lib and hasattr(lib, 'oracle_agg_t_f32'):
        ptr, idx = _by_source(batch)
        g = np.ascontiguousarray(g, f32)
        out = np.zeros_like(g)
        lib.oracle_agg_t_f32(ctypes.c_int64(batch.n), ctypes.c_int64(g.shape[1]), ptr.ctypes.data_as(ctypes.c_void_p),
                             idx.ctypes.data_as(ctypes.c_void_p), g.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        return out
    out = np.zeros_like(g)
    np.add.at(out, batch.indices, g[batch.dst])
    return out


# =========================================================================== model (a6, a7)
def parse_config(config):
    """train.py:67-75 config list -> (gcn dims, linear dims, link_pred)."""
    gcn = [tuple(p) for n, p in config if n == 'GraphConv']
    lin = [tuple(p) for n, p in config if n == 'Linear'][0]
    link = config[-1][0] == 'LinkPred'                     # learner.py:78-79
    return gcn, lin, link


def classifier_forward(batch, x0, vars_, config):
    """Classifier.forward (learner.py:134-175) with GraphConv.forward (learner.py:25-56).
    Returns logits [S,C] and a cache for the backward."""
    gcn, lin, link = parse_config(config)
    norm = batch.norm[:, None]
    h = np.asarray(x0, f32)
    cache = []
    for l, (fi, fo) in enumerate(gcn):
        W, b = vars_[2 * l], vars_[2 * l + 1]
        xs = h * norm                                       # learner.py:32
        if fi > fo:                                         # learner.py:34-40 matmul first
            y = xs @ W
            z = agg(batch.indptr, batch.indices, y)
            pre = z
        else:                                               # learner.py:41-47 aggregate first
            z = agg(batch.indptr, batch.indices, xs)
            pre = z @ W
        q = pre * norm + b                                  # learner.py:49-51
        hn = np.maximum(q, 0)                               # relu on every GCN layer (learner.py:97)
        cache.append((xs, z, hn, fi > fo))
        h = hn
    rows = batch.centre_rows
    hc = h[rows[:, 0]] if rows.shape[1] == 1 else np.concatenate([h[rows[:, 0]], h[rows[:, 1]]], 1)  # learner.py:165-170
    Wl, bl = vars_[2 * len(gcn)], vars_[2 * len(gcn) + 1]
    logits = hc @ Wl.T + bl                                 # F.linear (learner.py:174)
    return logits.astype(f32), (cache, hc, h.shape)


def classifier_backward(batch, vars_, config, fcache, dlogits):
    """Manual reverse pass of classifier_forward; returns grads in `vars` order."""
    gcn, lin, link = parse_config(config)
    cache, hc, hshape = fcache
    L = len(gcn)
    grads = [None] * len(vars_)
    Wl = vars_[2 * L]
    dlogits = np.asarray(dlogits, f32)
    grads[2 * L] = dlogits.T @ hc
    grads[2 * L + 1] = dlogits.sum(0)
    dhc = dlogits @ Wl
    dh = np.zeros(hshape, f32)
    rows = batch.centre_rows
    H = hshape[1]
    np.add.at(dh, rows[:, 0], dhc[:, :H])
    if rows.shape[1] == 2:
        np.add.at(dh, rows[:, 1], dhc[:, H:])
    norm = batch.norm[:, None]
    for l in range(L - 1, -1, -1):
        W = vars_[2 * l]
        xs, z, hn, mm_first = cache[l]
        dq = dh * (hn > 0)
        grads[2 * l + 1] = dq.sum(0)
        dpre = dq * norm
        if mm_first:
            dy = agg_t(batch, dpre)
            grads[2 * l] = xs.T @ dy
            dxs = dy @ W.T if l > 0 else None
        else:
            grads[2 * l] = z.T @ dpre
            dxs = agg_t(batch, dpre @ W.T) if l > 0 else None
        dh = dxs * norm if l > 0 else None
    return [g.astype(f32) for g in grads]


# =========================================================================== losses (a8, a9)
def _log_softmax(a):
    m = a.max(1, keepdims=True)
    return a - m - np.log(np.exp(a - m).sum(1, keepdims=True))


def _class_rows(y, limit=None):
    classes = np.unique(y)                                  # sorted (meta.py:35,60)
    rows = []
    for c in classes:
        r = np.nonzero(y == c)[0]
        rows.append(r[:limit] if limit is not None else r)
    cnt = {len(r) for r in rows}
    if len(cnt) != 1:
        raise ValueError('classes with unequal row counts (torch.stack at meta.py:42/65 fails)')
    return classes, np.stack(rows)                          # [C_task, n]


def proto_loss(logits, rows, protos):
    """Shared tail of meta.py:44-53 / 68-78.  rows [C_task,n]; returns loss, acc, G=dL/d(-dist)."""
    C, n = rows.shape
    q = logits[rows.reshape(-1)]                            # grouped by class
    if q.shape[1] != protos.shape[1]:
        raise Exception('feature-dim mismatch (meta.py:20-21)')
    d = ((q[:, None, :] - protos[None, :, :]) ** 2).sum(2)  # euclidean_dist meta.py:14-26
    logp = _log_softmax(-d)
    tgt = np.repeat(np.arange(C), n)
    loss = -logp[np.arange(C * n), tgt].mean()
    acc = (logp.argmax(1) == tgt).astype(f32).mean()
    G = np.exp(logp); G[np.arange(C * n), tgt] -= 1; G /= f32(C * n)   # dL/da, a = -d
    return f32(loss), f32(acc), G.astype(f32), q, tgt


def proto_loss_spt(logits, y, n_support, need_grad=True):
    """meta.py:28-54.  Returns loss, acc, prototypes, dlogits (both roles summed)."""
    classes, rows = _class_rows(y, n_support)
    protos = np.stack([logits[r].mean(0) for r in rows]).astype(f32)
    loss, acc, G, q, tgt = proto_loss(logits, rows, protos)
    dl = None
    if need_grad:
        dl = np.zeros_like(logits)
        diff = q[:, None, :] - protos[None, :, :]           # [Q,C,D]
        dq = (G[:, :, None] * (-2 * diff)).sum(1)           # query role
        dp = (G[:, :, None] * (2 * diff)).sum(0)            # prototype role [C,D]
        np.add.at(dl, rows.reshape(-1), dq)
        for c in range(len(classes)):
            dl[rows[c]] += dp[c] / f32(rows.shape[1])
    return loss, acc, protos, dl


def proto_loss_qry(logits, y, protos, need_grad=False):
    """meta.py:56-79.  Returns loss, acc, (dlogits, dprotos) if need_grad."""
    classes, rows = _class_rows(y)
    loss, acc, G, q, tgt = proto_loss(logits, rows, protos)
    if not need_grad:
        return loss, acc, None, None
    diff = q[:, None, :] - protos[None, :, :]
    dl = np.zeros_like(logits)
    np.add.at(dl, rows.reshape(-1), (G[:, :, None] * (-2 * diff)).sum(1))
    dp = (G[:, :, None] * (2 * diff)).sum(0)
    return loss, acc, dl.astype(f32), dp.astype(f32)


def protos_to_dlogits(y, n_support, dprotos, shape):
    """Prototype path back into the support logits: prototype_c = mean of first n_support rows."""
    classes, rows = _class_rows(y, n_support)
    dl = np.zeros(shape, f32)
    for c in range(len(classes)):
        dl[rows[c]] += dprotos[c] / f32(rows.shape[1])
    return dl


# =========================================================================== ProtoMAML (a10, a11)
def task_inner_loop(spt, qry, x_spt, x_qry, y_spt, y_qry, theta, config, k_spt, update_lr, K, need_meta_grad, trace=None):
    """One task of forward_ProtoMAML (meta.py:118-157) / finetunning_ProtoMAML (meta.py:197-229).
    Returns losses_q[K+1], accs_q[K+1], meta-grad list (first-order: query path at fw_K plus
    prototype path through the support forward at fw_{K-1})."""
    if K < 2 and need_meta_grad:
        raise ValueError('update_step must be >= 2 (meta.py:129-141 are no_grad)')
    lq, aq = np.zeros(K + 1, f32), np.zeros(K + 1, f32)
    fw = [v.copy() for v in theta]
    log = (lambda t, v: trace.append((t, v))) if trace is not None else (lambda t, v: None)
    # step 0 (meta.py:122-126)
    logit_s, cs = classifier_forward(spt, x_spt, fw, config); log('logits', logit_s)
    loss_s, _, protos, dls = proto_loss_spt(logit_s, y_spt, k_spt); log('loss_s', loss_s)
    g = classifier_backward(spt, fw, config, cs, dls)
    fw1 = [w - f32(update_lr) * gg for w, gg in zip(fw, g)]
    # query before / after first update (meta.py:129-141), both against step-0 prototypes
    for k, w in ((0, fw), (1, fw1)):
        logit_q, _ = classifier_forward(qry, x_qry, w, config); log('logits', logit_q)
        lq[k], aq[k], _, _ = proto_loss_qry(logit_q, y_qry, protos); log('loss_q', lq[k])
    fw_prev, fw = fw, fw1
    mg = None
    for k in range(1, K):                                   # meta.py:143-157
        logit_s, cs = classifier_forward(spt, x_spt, fw, config); log('logits', logit_s)
        loss_s, _, protos, dls = proto_loss_spt(logit_s, y_spt, k_spt); log('loss_s', loss_s)
        g = classifier_backward(spt, fw, config, cs, dls)
        fw_next = [w - f32(update_lr) * gg for w, gg in zip(fw, g)]
        logit_q, cq = classifier_forward(qry, x_qry, fw_next, config); log('logits', logit_q)
        last = need_meta_grad and k == K - 1
        lq[k + 1], aq[k + 1], dlq, dpr = proto_loss_qry(logit_q, y_qry, protos, need_grad=last)
        log('loss_q', lq[k + 1])
        if last:
            g_q = classifier_backward(qry, fw_next, config, cq, dlq)                    # d L_q / d fw_K
            dls_p = protos_to_dlogits(y_spt, k_spt, dpr, logit_s.shape)
            g_p = classifier_backward(spt, fw, config, cs, dls_p)                        # d L_q / d fw_{K-1}
            mg = [a + b for a, b in zip(g_q, g_p)]
        fw = fw_next
    return lq, aq, mg


def adam_step(theta, grad, state, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (meta.py:97)."""
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    out = []
    for k, (p, g) in enumerate(zip(theta, grad)):
        m = state.setdefault(('m', k), np.zeros_like(p)); v = state.setdefault(('v', k), np.zeros_like(p))
        m[:] = b1 * m + (1 - b1) * g
        v[:] = b2 * v + (1 - b2) * g * g
        denom = np.sqrt(v) / np.sqrt(1 - b2 ** t) + eps
        out.append((p - (lr / (1 - b1 ** t)) * m / denom).astype(f32))
    return out


def meta_step(graphs, feats, spt_batches, qry_batches, y_spt, y_qry, theta, config, k_spt,
              update_lr, meta_lr, K, adam_state=None, trace=None):
    """Meta.forward_ProtoMAML (meta.py:101-173) over T tasks.  Returns accs[K+1], grad, theta'."""
    T = len(spt_batches)
    lq_sum, aq_sum = np.zeros(K + 1, np.float64), np.zeros(K + 1, np.float64)
    gsum = [np.zeros_like(v) for v in theta]
    for t in range(T):
        xs, xq = spt_batches[t].features(feats), qry_batches[t].features(feats)
        lq, aq, mg = task_inner_loop(spt_batches[t], qry_batches[t], xs, xq, y_spt[t], y_qry[t], theta, config,
                                     k_spt, update_lr, K, True, trace)
        lq_sum += lq; aq_sum += aq
        gsum = [a + b for a, b in zip(gsum, mg)]
    grad = [(g / f32(T)).astype(f32) for g in gsum]
    loss_q = lq_sum[-1] / T
    new_theta = theta
    if not np.isnan(loss_q):                               # meta.py:163-169 NaN guard
        new_theta = adam_step(theta, grad, adam_state if adam_state is not None else {}, meta_lr)
    return aq_sum / T, grad, new_theta, lq_sum / T


def finetune(graphs, feats, spt, qry, y_spt, y_qry, theta, config, k_spt, update_lr, K_test, trace=None):
    """Meta.finetunning_ProtoMAML (meta.py:175-234) for one task; theta untouched."""
    lq, aq, _ = task_inner_loop(spt, qry, spt.features(feats), qry.features(feats), y_spt, y_qry, theta, config,
                                k_spt, update_lr, K_test, False, trace)
    return aq.astype(np.float64)
