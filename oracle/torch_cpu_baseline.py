"""TEST INFRASTRUCTURE ONLY -- the reference's inner loop written with plain PyTorch CPU ops and autograd, the closest
analogue of the reference's own DGL-CPU path that can run without DGL: `update_all(copy_src, sum)` (learner.py:38-39,44-45)
becomes `zeros.index_add_(0, dst, h[src])` (what DGL 0.4's SpMM computes), everything else is the reference's torch code
shape for shape (GraphConv.forward learner.py:25-56, Classifier.forward learner.py:134-175, proto_loss_spt/qry meta.py:28-79,
the task loop meta.py:118-157 with torch.autograd.grad like meta.py:125,149).

Used only by bench.py's cpu_baseline leg (kind "port", variant "torch-cpu") and by tests/test_torch_cpu_baseline.py, which
pins it to the numpy oracle (itself pinned to the reference's golden outputs).  Never imported by the product."""
import numpy as np
import torch
import torch.nn.functional as F


class TorchBatch:
    """Edge-list view of an oracle Batch (gmeta_oracle.Batch) as torch tensors."""

    def __init__(self, b):
        self.n = b.n
        self.src = torch.from_numpy(np.asarray(b.indices, np.int64))
        self.dst = torch.from_numpy(np.asarray(b.dst, np.int64))
        deg = torch.from_numpy(np.diff(b.indptr).astype(np.float32))
        self.norm = torch.pow(deg.clamp(min=1), -0.5).unsqueeze(1)            # learner.py:29-31
        self.centre_rows = torch.from_numpy(np.asarray(b.centre_rows, np.int64))


def graph_conv(tb, feat, W, b):
    """learner.py:25-56."""
    feat = feat * tb.norm
    agg = lambda h: torch.zeros(tb.n, h.shape[1], dtype=h.dtype).index_add_(0, tb.dst, h[tb.src])   # noqa: E731
    if W.shape[0] > W.shape[1]:
        rst = agg(torch.matmul(feat, W))
    else:
        rst = torch.matmul(agg(feat), W)
    return rst * tb.norm + b


def classifier(tb, x, vars_, n_gcn):
    """learner.py:134-175 (relu after every GraphConv, learner.py:97)."""
    h = x
    for l in range(n_gcn):
        h = F.relu(graph_conv(tb, h, vars_[2 * l], vars_[2 * l + 1]))
    rows = tb.centre_rows
    hc = h[rows[:, 0]] if rows.shape[1] == 1 else torch.cat((h[rows[:, 0]], h[rows[:, 1]]), 1)
    return F.linear(hc, vars_[2 * n_gcn], vars_[2 * n_gcn + 1])


def euclidean_dist(x, y):                                                     # meta.py:14-26
    n, m, d = x.size(0), y.size(0), x.size(1)
    if d != y.size(1):
        raise Exception
    return torch.pow(x.unsqueeze(1).expand(n, m, d) - y.unsqueeze(0).expand(n, m, d), 2).sum(2)


def proto_loss_spt(logits, y, n_support):                                     # meta.py:28-54
    classes = torch.unique(y)
    n_classes = len(classes)
    n_query = n_support
    idxs = [y.eq(c).nonzero()[:n_support].squeeze(1) for c in classes]
    prototypes = torch.stack([logits[i].mean(0) for i in idxs])
    query_samples = logits[torch.stack(idxs).view(-1)]
    log_p_y = F.log_softmax(-euclidean_dist(query_samples, prototypes), dim=1).view(n_classes, n_query, -1)
    target = torch.arange(0, n_classes).view(n_classes, 1, 1).expand(n_classes, n_query, 1).long()
    loss = -log_p_y.gather(2, target).squeeze().view(-1).mean()
    acc = log_p_y.max(2)[1].eq(target.squeeze(2)).float().mean()
    return loss, acc, prototypes


def proto_loss_qry(logits, y, prototypes):                                    # meta.py:56-79
    classes = torch.unique(y)
    n_classes = len(classes)
    n_query = int(logits.shape[0] / n_classes)
    idxs = [y.eq(c).nonzero().squeeze(1) for c in classes]
    query_samples = logits[torch.stack(idxs).view(-1)]
    log_p_y = F.log_softmax(-euclidean_dist(query_samples, prototypes), dim=1).view(n_classes, n_query, -1)
    target = torch.arange(0, n_classes).view(n_classes, 1, 1).expand(n_classes, n_query, 1).long()
    loss = -log_p_y.gather(2, target).squeeze().view(-1).mean()
    acc = log_p_y.max(2)[1].eq(target.squeeze(2)).float().mean()
    return loss, acc


def task_inner_loop(spt, qry, x_spt, x_qry, y_spt, y_qry, theta, n_gcn, k_spt, update_lr, K, need_meta_grad):
    """One task of forward_ProtoMAML (meta.py:118-157) + its share of `loss_q.backward()` (meta.py:161-168).
    Returns losses_q [K+1], accs [K+1], the task's meta-gradient (list, `vars` order) or None."""
    ts, tq = TorchBatch(spt), TorchBatch(qry)
    xs, xq = torch.from_numpy(np.ascontiguousarray(x_spt, np.float32)), torch.from_numpy(np.ascontiguousarray(x_qry, np.float32))
    ys, yq = torch.from_numpy(np.asarray(y_spt, np.int64)), torch.from_numpy(np.asarray(y_qry, np.int64))
    net = [torch.from_numpy(np.array(v, np.float32)).requires_grad_(True) for v in theta]
    lq, aq = [None] * (K + 1), np.zeros(K + 1, np.float32)
    logits = classifier(ts, xs, net, n_gcn)
    loss, _, prototypes = proto_loss_spt(logits, ys, k_spt)
    grad = torch.autograd.grad(loss, net)                                      # meta.py:125 (first order: no create_graph)
    fast = [p - update_lr * g for p, g in zip(net, grad)]
    with torch.no_grad():                                                      # meta.py:129-141
        for k, w in ((0, net), (1, fast)):
            l_, a_ = proto_loss_qry(classifier(tq, xq, w, n_gcn), yq, prototypes)
            lq[k], aq[k] = l_, float(a_)
    for k in range(1, K):                                                      # meta.py:143-157
        logits = classifier(ts, xs, fast, n_gcn)
        loss, _, prototypes = proto_loss_spt(logits, ys, k_spt)
        grad = torch.autograd.grad(loss, fast, retain_graph=True)              # meta.py:149
        fast = [p - update_lr * g for p, g in zip(fast, grad)]
        l_, a_ = proto_loss_qry(classifier(tq, xq, fast, n_gcn), yq, prototypes)
        lq[k + 1], aq[k + 1] = l_, float(a_)
    mg = None
    if need_meta_grad:
        lq[K].backward()                                                       # meta.py:161-168: grads reach `net` through fast = p - lr*g (g detached)
        mg = [p.grad.numpy() for p in net]
    return np.array([float(v.detach()) for v in lq], np.float32), aq, mg
