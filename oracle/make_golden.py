#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden-vector generator.  Runs in the BUILD CONTAINER only.

Imports the reference's own modules UNMODIFIED from /root/reference/G-Meta
(learner.py, meta.py, subgraph_data_processing.py) against the restated DGL surface in
oracle/dgl_shim (DGL 0.4.3post2 is third-party, absent from /root/reference and from the
image), drives them exactly like train.py:33-35,67-79,89-108,118-120 does, and dumps small
.npz fixtures into tests/golden/.  The fixtures are DATA (inputs + the reference's outputs);
no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Cases:
  g0_disjoint_h1   cfg-0 plumbing: Disjoint, 1 graph, h=1, 2-way 1-shot 5-qry, T=4, K=5
  g1_sampled_h2    Disjoint, h=2, sample_nodes=30 so np.random.choice fires (replay fixture)
  g1_h3            Disjoint, h=3, no sampling
  g2_shared        Shared, 3 graphs, C=2
  g3_linkpred      Shared link-pred on directed graphs (captures the sdp.py:332 quirk)
  g5_in_gt_out     F0=96 > H=32: matmul-first branch of GraphConv (learner.py:34-40)
  g6_nan_skip      inf feature -> NaN query loss -> optimiser step skipped (meta.py:163-169)
  g7_wide_h2       hidden 128 (F0 = 64), h=2, sample_nodes=60: layers wide enough for the split-MFMA update kernels (N = 128), which the
                   GPU tests force onto this fixture (gm_set_tuning) -- the reference's own outputs for the arithmetic bench.py times
  g8_wide_scales   Shared, 3 graphs whose features are scaled by 2^-20, 2^-8, 2^4 (hidden 128): tasks of very different magnitude in one
                   meta-batch -- the case a per-tensor operand scale must not lose precision on
  g9_wide_nan      hidden 128 with an inf feature table: NaN query loss -> no optimiser step, through the wide kernels
Every case also records Meta.finetunning on task 0 (G4).
"""
import sys
sys.dont_write_bytecode = True
import os, json, csv, random, tempfile, argparse, warnings  # noqa: E401
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/G-Meta'
sys.path.insert(0, os.path.join(HERE, 'dgl_shim'))
sys.path.insert(0, REF)
warnings.filterwarnings('ignore', category=SyntaxWarning)

import dgl                                   # noqa: E402  (the restated surface)
import subgraph_data_processing as sdp       # noqa: E402  (reference, unmodified)
import meta as refmeta                       # noqa: E402  (reference, unmodified)

OUT = os.path.join(HERE, '..', 'tests', 'golden')


# ----------------------------------------------------------------------------- synthetic data
def pa_edges(n, m, rng):
    """Preferential-attachment undirected edge list (u<v) with numpy only."""
    targets = list(range(m))
    rep = []
    edges = []
    for v in range(m, n):
        for t in set(targets):
            edges.append((t, v))
        rep.extend(set(targets))
        rep.extend([v] * m)
        targets = [rep[k] for k in rng.integers(0, len(rep), size=m)]
    return np.array(sorted(set(edges)), dtype=np.int64)


def make_graph(n, src, dst):
    g = dgl.DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    return g


def write_csv(path, names, labels):
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['', 'name', 'label'])
        for k, (a, b) in enumerate(zip(names, labels)):
            w.writerow([k, a, b])


# ----------------------------------------------------------------------------- recording
class Recorder:
    def __init__(self):
        self.logits, self.loss_s, self.loss_q, self.acc_q = [], [], [], []


def run_case(name, graphs, feats, info, csvs, args, config, T, hub_inf=False):
    """graphs: list of (n, src, dst); feats: list of [n,F0]; info: name->label;
    csvs: {filename: (names, labels)}."""
    torch.manual_seed(222); np.random.seed(222); random.seed(222)   # train.py:33-35 (+random)
    root = tempfile.mkdtemp(prefix='gmeta_golden_') + '/'
    for fn, (nm, lb) in csvs.items():
        write_csv(root + fn, nm, lb)
    G = [make_graph(*g) for g in graphs]
    maml = refmeta.Meta(args, config)
    vars0 = [p.detach().numpy().copy() for p in maml.net.parameters()]
    db = sdp.Subgraphs(root, 'train', info, n_way=args.n_way, k_shot=args.k_spt, k_query=args.k_qry,
                       batchsz=T, args=args, adjs=G, h=args.h)
    samples = [db[t] for t in range(T)]
    batch = sdp.collate(samples)
    x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry = batch

    out = {'case': name, 'T': T}
    # ---- graphs / features
    out['n_graphs'] = len(graphs)
    for k, (n, s, d) in enumerate(graphs):
        out['g%d_n' % k] = n
        out['g%d_src' % k] = np.asarray(s, np.int32)
        out['g%d_dst' % k] = np.asarray(d, np.int32)
        out['g%d_feat' % k] = np.asarray(feats[k], np.float32)
    # ---- seeds (graph, i, j|-1) in the order __getitem__ flattens them (sdp.py:355-362)
    def seeds_of(batchlist):
        res = []
        for t in range(T):
            row = []
            for sub in batchlist[t]:
                for item in sub:
                    p = [int(x) for x in item.split('_')]
                    row.append(p + [-1] if len(p) == 2 else p)
            res.append(row)
        return np.array(res, np.int32)
    out['spt_seeds'] = seeds_of(db.support_x_batch)
    out['qry_seeds'] = seeds_of(db.query_x_batch)
    out['y_spt'] = np.stack([y.numpy() for y in y_spt]).astype(np.int32)
    out['y_qry'] = np.stack([y.numpy() for y in y_qry]).astype(np.int32)
    out['c_spt'] = np.stack([c.numpy() for c in c_spt]).astype(np.int32)   # reference-order local idx
    out['c_qry'] = np.stack([c.numpy() for c in c_qry]).astype(np.int32)

    # ---- per-subgraph node lists (reference order) and induced edges in parent ids
    def ragged(key, lists):
        flat = np.concatenate([np.asarray(x, np.int32).reshape(-1) for x in lists]) if lists else np.zeros(0, np.int32)
        off = np.cumsum([0] + [len(x) for x in lists]).astype(np.int64)
        out[key + '_flat'], out[key + '_off'] = flat, off
    for tag, nlist, glist, blist in (('spt', n_spt, g_spt, db.support_x_batch), ('qry', n_qry, g_qry, db.query_x_batch)):
        nodes, edges = [], []
        for t in range(T):
            names = [item for sub in blist[t] for item in sub]
            for k, item in enumerate(names):
                sub, _, h_c = db.subgraphs[item]
                par = np.asarray(h_c, np.int64)
                nodes.append(par)
                e = np.stack([par[sub._src.numpy()], par[sub._dst.numpy()]], 1) if sub.number_of_edges() else np.zeros((0, 2), np.int64)
                e = e[np.lexsort((e[:, 1], e[:, 0]))]
                edges.append(e.reshape(-1))
        ragged(tag + '_nodes', nodes)
        ragged(tag + '_edges', edges)          # pairs (src,dst), lexsorted, flattened

    # ---- instrument the reference's own functions (recording only)
    rec = Recorder()
    o_spt, o_qry = refmeta.proto_loss_spt, refmeta.proto_loss_qry
    def spt_w(logits, y, n):
        l, a, p = o_spt(logits, y, n); rec.loss_s.append(float(l)); return l, a, p
    def qry_w(logits, y, p):
        l, a = o_qry(logits, y, p); rec.loss_q.append(float(l)); rec.acc_q.append(float(a)); return l, a
    refmeta.proto_loss_spt, refmeta.proto_loss_qry = spt_w, qry_w
    hook = lambda m, i, o: rec.logits.append(o[0].detach().numpy().copy())  # noqa: E731
    maml.net.register_forward_hook(hook)

    feat_list = [np.asarray(f, np.float32) for f in feats]
    # ---- G4: finetunning on task 0 with the INITIAL weights (meta.py:175-234)
    one = [[b[0]] for b in batch]
    ft = maml.finetunning(*one, feat_list)
    out['ft_accs'] = np.asarray(ft, np.float64)
    out['ft_logits_flat'] = np.concatenate([x.reshape(-1) for x in rec.logits]).astype(np.float32)
    out['ft_loss_s'] = np.array(rec.loss_s, np.float32)
    out['ft_loss_q'] = np.array(rec.loss_q, np.float32)
    rec.__init__()

    # ---- the meta-training step (meta.py:101-173); capture theta.grad before Adam
    grads = {}
    o_step = maml.meta_optim.step
    def step_w(*a, **k):
        grads['g'] = [p.grad.detach().numpy().copy() for p in maml.net.parameters()]
        return o_step(*a, **k)
    maml.meta_optim.step = step_w
    accs = maml(*batch, feat_list)
    refmeta.proto_loss_spt, refmeta.proto_loss_qry = o_spt, o_qry
    K = args.update_step
    out['accs'] = np.asarray(accs, np.float64)
    out['logits_flat'] = np.concatenate([x.reshape(-1) for x in rec.logits]).astype(np.float32)
    out['loss_s'] = np.array(rec.loss_s, np.float32).reshape(T, K)
    out['loss_q'] = np.array(rec.loss_q, np.float32).reshape(T, K + 1)
    out['acc_q'] = np.array(rec.acc_q, np.float32).reshape(T, K + 1)
    out['stepped'] = int('g' in grads)
    for k, v in enumerate(vars0):
        out['vars0_%d' % k] = v
    for k, p in enumerate(maml.net.parameters()):
        out['vars1_%d' % k] = p.detach().numpy().copy()
        if 'g' in grads:
            out['grad_%d' % k] = grads['g'][k]
    out['n_vars'] = len(vars0)
    out['config'] = json.dumps(config)
    out['args'] = json.dumps(vars(args))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    sz = os.path.getsize(os.path.join(OUT, name + '.npz'))
    print('%-18s T=%d accs=%s  stepped=%d  %.1f KB' % (name, T, np.round(accs, 3), out['stepped'], sz / 1024))


def ns(**kw):
    base = dict(update_lr=0.01, meta_lr=1e-3, n_way=3, k_spt=2, k_qry=4, task_num=2, update_step=3,
                update_step_test=4, method='G-Meta', sample_nodes=1000, link_pred_mode='False',
                task_setup='Disjoint', h=2)
    base.update(kw)
    return argparse.Namespace(**base)


def node_case(name, n, m, F0, H, n_cls, args, T, seed, hub_inf=False):
    rng = np.random.default_rng(seed)
    e = pa_edges(n, m, rng)
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])  # both directions
    feat = rng.standard_normal((n, F0)).astype(np.float32)
    if hub_inf:
        feat[:] = np.inf        # every neighbourhood sees it -> NaN losses
    lab = rng.integers(0, n_cls, size=n)
    names = ['0_%d' % v for v in range(n)]
    info = {nm: int(l) for nm, l in zip(names, lab)}
    config = [('GraphConv', [F0, H])] + [('GraphConv', [H, H])] * (args.h - 1) + [('Linear', [H, args.n_way])]
    run_case(name, [(n, src, dst)], [feat], info, {'train.csv': (names, [str(l) for l in lab])}, args, config, T)


def shared_case(name, T, seed, F0=10, H=16, scales=None, update_lr=0.05):
    rng = np.random.default_rng(seed)
    graphs, feats, names, labels, info = [], [], [], [], {}
    for g in range(3):
        n = 120 + 20 * g
        e = pa_edges(n, 2, rng)
        graphs.append((n, np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])))
        feats.append((rng.standard_normal((n, F0)) * (scales[g] if scales else 1.0)).astype(np.float32))
        lab = rng.integers(0, 2, size=n)
        for v in range(n):
            nm = '%d_%d' % (g, v); names.append(nm); labels.append(str(lab[v])); info[nm] = int(lab[v])
    args = ns(task_setup='Shared', n_way=2, k_spt=3, k_qry=5, task_num=T, update_step=4, update_step_test=3, update_lr=update_lr)
    config = [('GraphConv', [F0, H]), ('GraphConv', [H, H]), ('Linear', [H, 2])]     # C = total_class (train.py:61)
    run_case(name, graphs, feats, info, {'train.csv': (names, labels)}, args, config, T)


def linkpred_case(name, T, seed):
    rng = np.random.default_rng(seed)
    graphs, feats, info = [], [], {}
    csv_all, csv_spt, csv_qry = ([], []), ([], []), ([], [])
    F0, H = 5, 16
    for g in range(2):
        n = 90 + 10 * g
        e = pa_edges(n, 3, rng)                       # positives, u<v only (link_process.py:32-47)
        neg = set()
        while len(neg) < len(e):
            a, b = rng.integers(0, n, size=2)
            if a != b and (min(a, b), max(a, b)) not in set(map(tuple, e)):
                neg.add((int(a), int(b)))
        neg = np.array(sorted(neg), np.int64)
        src = np.concatenate([e[:, 0], neg[:, 0]]); dst = np.concatenate([e[:, 1], neg[:, 1]])  # negatives injected (link_process.py:83-85)
        graphs.append((n, src, dst))
        feats.append(rng.standard_normal((n, F0)).astype(np.float32))
        for arr, lab in ((e, 1), (neg, 0)):
            perm = rng.permutation(len(arr))
            half = len(arr) // 2
            for k, idx in enumerate(perm):
                nm = '%d_%d_%d' % (g, arr[idx, 0], arr[idx, 1])
                info[nm] = lab
                csv_all[0].append(nm); csv_all[1].append(str(lab))
                tgt = csv_spt if k < half else csv_qry
                tgt[0].append(nm); tgt[1].append(str(lab))
    args = ns(task_setup='Shared', link_pred_mode='True', n_way=2, k_spt=4, k_qry=6, task_num=T,
              update_step=3, update_step_test=3, sample_nodes=25, update_lr=0.1)
    config = [('GraphConv', [F0, H]), ('GraphConv', [H, H]), ('Linear', [H, 2]), ('LinkPred', [True])]
    run_case(name, graphs, feats, info,
             {'train.csv': csv_all, 'train_spt.csv': csv_spt, 'train_qry.csv': csv_qry}, args, config, T)


if __name__ == '__main__':
    only = set(sys.argv[1:])            # optional: names of the cases to (re)generate; default all
    cases = [
        ('g0_disjoint_h1', lambda nm: node_case(nm, 300, 3, 32, 64, 10,
                                                ns(h=1, n_way=2, k_spt=1, k_qry=5, task_num=4, update_step=5, update_step_test=10, update_lr=0.001), 4, 1)),
        ('g1_sampled_h2', lambda nm: node_case(nm, 400, 4, 16, 24, 6, ns(h=2, sample_nodes=30, update_step=3, update_lr=0.1), 2, 2)),
        ('g1_h3', lambda nm: node_case(nm, 250, 2, 12, 16, 5, ns(h=3, update_step=2, update_step_test=2, update_lr=0.05), 2, 3)),
        ('g2_shared', lambda nm: shared_case(nm, 3, 4)),
        ('g3_linkpred', lambda nm: linkpred_case(nm, 2, 5)),
        ('g5_in_gt_out', lambda nm: node_case(nm, 200, 3, 96, 32, 5, ns(h=2, update_step=3, update_lr=0.05), 2, 6)),
        ('g6_nan_skip', lambda nm: node_case(nm, 150, 3, 8, 8, 5, ns(h=1, update_step=2, update_step_test=2), 2, 7, hub_inf=True)),
        ('g7_wide_h2', lambda nm: node_case(nm, 500, 4, 64, 128, 6, ns(h=2, sample_nodes=60, update_step=3, update_lr=0.05), 2, 8)),
        ('g8_wide_scales', lambda nm: shared_case(nm, 3, 9, F0=32, H=128, scales=(2.0 ** -20, 2.0 ** -8, 2.0 ** 4), update_lr=1e-3)),
        ('g9_wide_nan', lambda nm: node_case(nm, 200, 3, 32, 128, 5, ns(h=2, update_step=2, update_step_test=2), 2, 10, hub_inf=True)),
    ]
    for nm, fn in cases:
        if not only or nm in only:
            fn(nm)
