"""GPU (-m gpu): HIP vs oracle at the SHAPES of BASELINE.json configs[3] (Tissue-PPI: Shared multi-graph, F0=50, hidden
128 -- feature width not a multiple of 4, so the scalar/generic kernels run) and configs[4] (FirstMM-DB: directed
multi-graph link prediction, F0=5, pair centres, head [2, 2H]), at sizes the oracle finishes in seconds.
Integer work bit-exact (also with sampling active); floats within 1e-4."""
import argparse

import numpy as np
import pytest
import torch

import gmeta_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _tissue(rng, n_graphs=4, n=700, F0=50):
    from gmeta_amd import synth
    graphs, feats, names, labels, info = [], [], [], [], {}
    for g in range(n_graphs):
        e = synth.pa_edges(n, 6, rng)
        graphs.append((n, np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])))
        feats.append(rng.standard_normal((n, F0)).astype(np.float32))
        lab = rng.integers(0, 2, size=n)
        for v in range(n):
            nm = '%d_%d' % (g, v); names.append(nm); labels.append(str(lab[v])); info[nm] = int(lab[v])
    return graphs, feats, info, {'train': (names, labels)}


def _firstmm(rng, n_graphs=3, n=500, F0=5):
    from gmeta_amd import synth
    graphs, feats, info = [], [], {}
    tabs = {'train': ([], []), 'train_spt': ([], []), 'train_qry': ([], [])}
    for g in range(n_graphs):
        e = synth.pa_edges(n, 3, rng)                                    # positives stored once, u < v (link_process.py:32-47)
        neg = rng.integers(0, n, size=(len(e), 2)); neg = neg[neg[:, 0] != neg[:, 1]]
        graphs.append((n, np.concatenate([e[:, 0], neg[:, 0]]), np.concatenate([e[:, 1], neg[:, 1]])))   # negatives injected
        feats.append(rng.standard_normal((n, F0)).astype(np.float32))
        for arr, lab in ((e, 1), (neg, 0)):
            for k, (a, b) in enumerate(arr[:200]):
                nm = '%d_%d_%d' % (g, a, b)
                if nm in info:
                    continue
                info[nm] = lab
                for key in ('train', 'train_spt' if k % 2 == 0 else 'train_qry'):
                    tabs[key][0].append(nm); tabs[key][1].append(str(lab))
    return graphs, feats, info, tabs


def _run_case(kind, sample_nodes):
    import random
    import gmeta_amd
    from gmeta_amd import synth
    rng = np.random.default_rng(5)
    np.random.seed(5); random.seed(5); torch.manual_seed(5)
    link = kind == 'firstmm'
    graphs, feats, info, tabs = (_firstmm if link else _tissue)(rng)
    F0, H = feats[0].shape[1], 128
    args = argparse.Namespace(update_lr=0.05, meta_lr=5e-3, n_way=2, k_spt=3 if not link else 4, k_qry=10 if not link else 8, task_num=2,
                              update_step=3, update_step_test=3, method='G-Meta', sample_nodes=sample_nodes,
                              link_pred_mode='True' if link else 'False', task_setup='Shared', h=2)
    config = synth.make_config(F0, H, 2, 2, link=link)
    store = gmeta_amd.GraphStore(graphs, feats)
    db = gmeta_amd.Subgraphs(None, 'train', info, n_way=2, k_shot=args.k_spt, k_query=args.k_qry, batchsz=2, args=args, adjs=store, h=2,
                             tables=tabs, verbose=False)
    batch = db.get_batch([0, 1])
    S, Q = batch[0][0].view_of, batch[2][0].view_of
    og = [orc.Graph(*g) for g in graphs]
    seeds = [(db._seeds(db._task_names(t)[0]), db._seeds(db._task_names(t)[1])) for t in range(2)]
    ospt = [orc.extract_batch(og, seeds[t][0], 2, sample_nodes, 222, link) for t in range(2)]
    oqry = [orc.extract_batch(og, seeds[t][1], 2, sample_nodes, 222, link) for t in range(2)]
    # ---- integer work: bit-exact
    assert np.array_equal(S.parent(), np.concatenate([b.parent for b in ospt]))
    assert np.array_equal(Q.parent(), np.concatenate([b.parent for b in oqry]))
    e0 = r0 = 0
    ip, ix = Q.csr()
    for b in oqry:
        assert np.array_equal(ip[r0:r0 + b.n + 1] - e0, b.indptr) and np.array_equal(ix[e0:e0 + len(b.indices)] - r0, b.indices)
        r0 += b.n; e0 += len(b.indices)
    return dict(args=args, config=config, batch=batch, feats=feats, og=og, ospt=ospt, oqry=oqry, link=link)


@pytest.mark.parametrize('kind', ['tissue', 'firstmm'])
def test_sampled_extraction_bit_exact(kind):
    c = _run_case(kind, sample_nodes=60)
    sizes = np.diff(c['batch'][2][0].view_of.sub_off)
    assert sizes.max() <= 62 and (sizes >= 60).any()          # sampling really fired (size k, k+1 or k+2)


@pytest.mark.parametrize('cone', [0, 1])
@pytest.mark.parametrize('kind', ['tissue', 'firstmm'])
def test_meta_step_matches_oracle(kind, cone):
    import gmeta_amd
    c = _run_case(kind, sample_nodes=100000)                  # no sampling: every centre keeps its neighbourhood (well conditioned)
    torch.manual_seed(11)
    m = gmeta_amd.Meta(c['args'], c['config']).to('cuda')
    m.cone = cone                                             # receptive-field schedule (link-pred: two centres per subgraph)
    theta0 = [p.detach().cpu().numpy().copy() for p in m.net.parameters()]
    grads = {}
    orig = m.meta_optim.step
    m.meta_optim.step = lambda *a, **k: (grads.setdefault('g', torch.cat([p.grad.reshape(-1) for p in m.net.parameters()]).cpu().numpy().copy()), orig(*a, **k))[1]
    b = c['batch']
    accs = m(*b, c['feats'])
    ys = [np.asarray(y) for y in b[1]]; yq = [np.asarray(y) for y in b[3]]
    oaccs, ograd, otheta, lq = orc.meta_step(c['og'], c['feats'], c['ospt'], c['oqry'], ys, yq, theta0, c['config'], c['args'].k_spt,
                                             c['args'].update_lr, c['args'].meta_lr, 3, adam_state={})
    np.testing.assert_allclose(accs, oaccs, atol=1e-6)
    np.testing.assert_allclose(m.last_stats['losses_q'], lq, atol=TOL)
    og = np.concatenate([g.reshape(-1) for g in ograd])
    np.testing.assert_allclose(grads['g'], og, atol=TOL, rtol=1e-4)
    # finetunning on every task in one call == the oracle's per-task loop (theta is the post-Adam one on both sides? no:
    # compare with the oracle run from the HIP side's own updated weights)
    theta1 = [p.detach().cpu().numpy().copy() for p in m.net.parameters()]
    ft = m.finetunning_batch(b[0], b[1], b[2], b[3])
    for t in range(2):
        o = orc.finetune(c['og'], c['feats'], c['ospt'][t], c['oqry'][t], ys[t], yq[t], theta1, c['config'], c['args'].k_spt,
                         c['args'].update_lr, 3)
        np.testing.assert_allclose(ft[t], o, atol=1e-6)


def test_dataloader_getitem_path_equals_batched_extraction():
    """The reference's own loop shape (train.py:96-108): DataLoader -> Subgraphs.__getitem__ per task -> collate ->
    Meta.forward (which dgl.batch-es the per-task handles with gm_batch_concat) must give bit-identical results to the
    batched get_batch fast path, and the 10-tuple must have the reference's slot types."""
    import random
    import gmeta_amd
    from gmeta_amd import synth
    from torch.utils.data import DataLoader
    rng = np.random.default_rng(9)
    np.random.seed(9); random.seed(9); torch.manual_seed(9)
    graphs, feats, info, tabs = _tissue(rng, n_graphs=3, n=400, F0=16)
    args = argparse.Namespace(update_lr=0.05, meta_lr=5e-3, n_way=2, k_spt=3, k_qry=6, task_num=3, update_step=3, update_step_test=3,
                              method='G-Meta', sample_nodes=50, link_pred_mode='False', task_setup='Shared', h=2)
    store = gmeta_amd.GraphStore(graphs, feats)
    db = gmeta_amd.Subgraphs(None, 'train', info, n_way=2, k_shot=3, k_query=6, batchsz=3, args=args, adjs=store, h=2, tables=tabs, verbose=False)
    config = synth.make_config(16, 32, 2, 2)
    batch_dl = next(iter(DataLoader(db, 3, shuffle=False, collate_fn=gmeta_amd.collate)))
    x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry = batch_dl
    assert len(batch_dl) == 10 and all(len(s) == 3 for s in batch_dl)
    assert isinstance(x_spt[0], gmeta_amd.SubgraphBatch) and x_spt[0].batch_num_nodes == [len(n) for n in n_spt[0]]
    assert y_spt[0].dtype == torch.int64 and c_spt[0].dtype == torch.int64 and isinstance(g_spt[0], list)
    par = x_spt[0].parent(); off = x_spt[0].sub_off
    for k in range(len(n_spt[0])):                                        # slot 6 == list(sub.parent_nid) per subgraph (sdp.py:317)
        ids = np.asarray(n_spt[0][k])
        assert np.array_equal(ids, par[off[k]:off[k + 1]])
        assert ids[int(c_spt[0][k])] == int(db._task_names(0)[0][k].split('_')[1])   # centre index points at the seed node
    batch_fast = db.get_batch([0, 1, 2])

    def run(b):
        torch.manual_seed(4)
        m = gmeta_amd.Meta(args, config).to('cuda')
        g = {}
        orig = m.meta_optim.step
        m.meta_optim.step = lambda *a, **k: (g.setdefault('g', torch.cat([p.grad.reshape(-1) for p in m.net.parameters()]).clone()), orig(*a, **k))[1]
        return m(*b, feats), g['g']
    a1, g1 = run(batch_dl)
    a2, g2 = run(batch_fast)
    assert np.array_equal(a1, a2) and torch.equal(g1, g2)


def test_prefetching_iterator_equals_sequential_get_batch():
    """Subgraphs.batches(prefetch=1): meta-batches extracted by a background thread on its own stream (the reference's
    DataLoader num_workers, train.py:96,173) are the same objects get_batch builds, in order, with the cone tables ready."""
    import ctypes as C
    import random
    import gmeta_amd
    from gmeta_amd import _lib, synth
    rng = np.random.default_rng(5)
    np.random.seed(5)
    graphs, feats, info, tabs = _tissue(rng, n_graphs=3, n=300, F0=16)
    args = argparse.Namespace(update_lr=0.05, meta_lr=5e-3, n_way=2, k_spt=3, k_qry=6, task_num=2, update_step=3, update_step_test=3,
                              method='G-Meta', sample_nodes=40, link_pred_mode='False', task_setup='Disjoint', h=2)
    store = gmeta_amd.GraphStore(graphs, feats)
    db = gmeta_amd.Subgraphs(None, 'train', info, n_way=2, k_shot=3, k_query=6, batchsz=8, args=args, adjs=store, h=2, tables=tabs, verbose=False)
    lists = [[0, 1], [2, 3], [4, 5], [6, 7]]
    random.seed(77)
    seq = [db.get_batch(idx) for idx in lists]
    random.seed(77)
    pre = list(db.batches(lists, prefetch=2, cone_layers=2))
    assert len(pre) == len(seq)
    for a, b in zip(seq, pre):
        A, B = a[0][0].view_of, b[0][0].view_of
        assert np.array_equal(A.parent(), B.parent()) and np.array_equal(A.csr()[1], B.csr()[1])
        assert all(torch.equal(x, y) for x, y in zip(a[1] + a[3], b[1] + b[3]))           # relabelled targets: same Python-RNG order
        ok = C.c_int32()
        _lib.check(_lib.lib().gm_batch_cone_dims(B.handle, 2, C.byref(ok), None, None), 'cone_dims')     # built by the worker
        assert ok.value == 1
    config = synth.make_config(16, 32, 2, 2)
    torch.manual_seed(1); m1 = gmeta_amd.Meta(args, config).to('cuda'); m1.cone = 1
    torch.manual_seed(1); m2 = gmeta_amd.Meta(args, config).to('cuda'); m2.cone = 1
    for a, b in zip(seq, pre):
        assert np.array_equal(m1(*a, feats), m2(*b, feats))
    # a failure inside the worker surfaces in the consumer
    with pytest.raises(Exception):
        list(db.batches([[0, 1], [999, 1000]], prefetch=1))
