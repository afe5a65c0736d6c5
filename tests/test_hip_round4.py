"""GPU (-m gpu): round-4 parity cases for the split-MFMA update kernels.

The reference multiplies in fp32 (torch.matmul(feat, weight), learner.py:36,47).  The library's large-launch update kernels are the
three-piece bf16 split kernels (24 significand bits per operand, the default) and -- opt-in -- the two-piece fp16 ones (22 bits).  The
reference-generated fixtures are far too small for either to be selected by launch size, so these tests FORCE them on
(gm_set_tuning: GM_GEMM_SPLIT_MIN_TILES = 0, GM_SPLIT16_MIN_ROWS = 0) over the three hidden-128 fixtures (g7 sampled h = 2, g8 three
graphs whose features are 2^-20 / 2^-8 / 2^4 apart, g9 inf features) and compare with the REFERENCE'S OWN outputs, and check the guards of
the opt-in mode: an outlier feature keeps the three-piece kernels, a fast weight that outgrows its bound re-runs the step three-piece."""
import argparse
import ctypes as C
import random

import numpy as np
import pytest
import torch

from golden_util import NAN_CASES, WIDE_CASES, Fixture

pytestmark = pytest.mark.gpu
TOL = 1e-4


class forced_split:
    """Split kernels for every eligible launch whatever its size; pieces = 2 / 3 per operand."""

    def __init__(self, pieces):
        self.pieces = pieces

    def __enter__(self):
        from gmeta_amd import _lib
        self.lib = lib = _lib.lib()
        _lib.check(lib.gm_set_tuning(b'GM_GEMM_SPLIT_MIN_TILES', 0), 'set_tuning')
        _lib.check(lib.gm_set_tuning(b'GM_SPLIT16_MIN_ROWS', 0), 'set_tuning')
        _lib.check(lib.gm_set_tuning(b'GM_WGRAD_SPLIT_MIN_CHUNKS', 0), 'set_tuning')
        lib.gm_set_split_pieces(self.pieces)
        lib.gm_profile_enable(1)
        return self

    def launches(self):
        """(three-piece GEMM, three-piece wgrad, two-piece GEMM, two-piece wgrad, exact-fp32 GEMM) launches of the last gm_meta_step"""
        out = []
        for cat in (4, 5, 6, 7, 1):
            ms, n, w = C.c_double(), C.c_int64(), C.c_int64()
            self.lib.gm_profile_read(cat, C.byref(ms), C.byref(n), C.byref(w))
            out.append(int(n.value))
        return out

    def __exit__(self, *exc):
        lib = self.lib
        lib.gm_profile_enable(0)
        lib.gm_set_split_pieces(-1)
        lib.gm_set_tuning(b'GM_GEMM_SPLIT_MIN_TILES', -1)
        lib.gm_set_tuning(b'GM_SPLIT16_MIN_ROWS', 65536)
        lib.gm_set_tuning(b'GM_WGRAD_SPLIT_MIN_CHUNKS', -1)
        return False


@pytest.mark.parametrize('case', WIDE_CASES)
@pytest.mark.parametrize('pieces', [3, 2])
def test_split_kernels_forced_onto_reference_fixtures(case, pieces):
    """accs, losses_q, theta.grad, post-Adam weights, NaN skip == the reference's own outputs with the split kernels doing every
    update GEMM they are eligible for (the arithmetic bench.py times)."""
    from hip_util import hip_meta_step
    fx = Fixture(case)
    with forced_split(pieces) as fs:
        res = hip_meta_step(fx, replay=True)
        g3, w3, g2, w2, g1 = fs.launches()
    assert g3 + g2 > 0, 'no split GEMM launch: the fixture did not exercise the kernels under test'
    if pieces == 2 and case not in NAN_CASES:
        assert g2 > 0 and w2 > 0, (g3, w3, g2, w2)
    if pieces == 2 and case in NAN_CASES:
        assert g2 == 0 and w2 == 0, 'an inf feature table must keep the three-piece kernels (looseness guard)'
    if pieces == 3:
        assert g2 == 0 and w2 == 0 and w3 > 0
    if case in NAN_CASES:
        assert np.isnan(res['stats']['loss_q'])
        assert 'rerun_three_piece' not in res['stats']                   # a TRUE NaN is skipped like the reference skips it, not re-run
        for a, b in zip(res['vars1'], fx.vars1):
            assert np.array_equal(a, b)
        return
    np.testing.assert_allclose(res['accs'], fx.z['accs'], atol=1e-6)
    np.testing.assert_allclose(res['stats']['losses_q'], fx.z['loss_q'].mean(0), atol=TOL)
    ref_g = np.concatenate([g.reshape(-1) for g in fx.grad])
    np.testing.assert_allclose(res['grad'], ref_g, atol=TOL, rtol=0)
    for a, b, g in zip(res['vars1'], fx.vars1, fx.grad):
        m = np.abs(g) > 1e-5
        np.testing.assert_allclose(a[m], b[m], atol=TOL, rtol=0)


@pytest.mark.parametrize('pieces', [3, 2])
def test_tasks_of_very_different_magnitude_keep_their_precision(pieces):
    """g8: the three tasks' features are 2^-20, 2^-8 and 2^4 times N(0,1).  Each task ALONE through the forced split kernels against the
    oracle: per-task query losses of every step and the per-task meta-gradient to 1e-4 RELATIVE to that task's own scale -- a per-tensor
    (instead of per-task) operand scale would leave the 2^-20 task with a handful of bits."""
    import gmeta_oracle as orc
    import gmeta_amd
    from hip_util import fixture_meta, make_store
    fx = Fixture('g8_wide_scales')
    graphs = fx.graphs()
    store = make_store(fx)

    def one(tag, t):
        return gmeta_amd.SubgraphBatch.from_nodes(store, fx.z[tag + '_seeds'][t], [0, fx.z[tag + '_seeds'].shape[1]], fx.replay_lists(tag, t), fx.link)
    for t in range(fx.T):
        bs = orc.extract_batch(graphs, fx.z['spt_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('spt', t))
        bq = orc.extract_batch(graphs, fx.z['qry_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('qry', t))
        lq, aq, mg = orc.task_inner_loop(bs, bq, bs.features(fx.feats), bq.features(fx.feats), fx.z['y_spt'][t], fx.z['y_qry'][t], fx.vars0, fx.config,
                                         fx.args['k_spt'], fx.args['update_lr'], fx.K, True)
        m = fixture_meta(fx)
        grads = {}
        m.meta_optim.step = lambda *a, **k: grads.setdefault('g', [p.grad.detach().cpu().numpy().copy() for p in m.net.parameters()])
        ys = [torch.from_numpy(fx.z['y_spt'][t].astype(np.int64))]; yq = [torch.from_numpy(fx.z['y_qry'][t].astype(np.int64))]
        with forced_split(pieces) as fs:
            accs = m([one('spt', t)], ys, [one('qry', t)], yq, None, None, None, None, None, None, fx.feats)
            g3, w3, g2, w2, g1 = fs.launches()
        assert (g2 if pieces == 2 else g3) > 0
        np.testing.assert_allclose(accs, aq, atol=1e-6)
        np.testing.assert_allclose(m.last_stats['losses_q'], lq, rtol=TOL, atol=1e-6)
        np.testing.assert_allclose(m.last_stats['losses_q'], fx.z['loss_q'][t], rtol=TOL, atol=1e-6)      # ... and the reference's own per-task losses
        for a, b in zip(grads['g'][:-1], mg[:-1]):      # (d loss / d b_linear is identically 0 up to fp noise: prototype distances are shift invariant)
            scale = float(np.abs(b).max())
            assert float(np.abs(a - b).max()) <= TOL * max(scale, 1e-30), (t, float(np.abs(a - b).max()), scale)


def _arxiv4(outlier=None):
    import gmeta_amd
    from gmeta_amd import synth
    np.random.seed(222); random.seed(222); torch.manual_seed(222)
    args, cfg = synth.make_args('arxiv', task_num=4)
    data = synth.make_dataset(cfg)
    if outlier is not None:
        data['feats'][0] = data['feats'][0].copy()
        data['feats'][0][12345, 7] = outlier
    store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
    db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=4, args=args,
                             adjs=store, h=cfg['h'], tables=data['tables'], verbose=False)
    batch = db.get_batch([0, 1, 2, 3])
    config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])
    return args, data, batch, config, store


def _run(args, data, batch, config, pieces, update_lr=None):
    import gmeta_amd
    from gmeta_amd import _lib
    lib = _lib.lib()
    torch.manual_seed(5)
    a = argparse.Namespace(**vars(args))
    if update_lr is not None:
        a.update_lr = update_lr
    m = gmeta_amd.Meta(a, config).to('cuda')
    theta0 = [p.detach().clone() for p in m.net.parameters()]
    lib.gm_set_split_pieces(pieces)
    lib.gm_profile_enable(1)
    try:
        accs = m(*batch, data['feats'])
        torch.cuda.synchronize()
        n = []
        for cat in (4, 6):
            ms, c, w = C.c_double(), C.c_int64(), C.c_int64()
            lib.gm_profile_read(cat, C.byref(ms), C.byref(c), C.byref(w))
            n.append(int(c.value))
    finally:
        lib.gm_profile_enable(0)
        lib.gm_set_split_pieces(-1)
    grads = [p.grad.detach().clone() if p.grad is not None else None for p in m.net.parameters()]
    moved = max(float((p.detach() - q).abs().max()) for p, q in zip(m.net.parameters(), theta0))
    return np.asarray(accs), grads, moved, m, n


def test_two_piece_mode_keeps_three_pieces_for_a_loose_feature_table():
    """One feature entry 2^20 above the typical magnitude: under ONE bound for the table every ordinary entry would keep
    min(22, 39 - 20) = 19 bits at best and fewer as the outlier grows -- silently.  The opt-in two-piece mode must notice (largest entry
    more than 2^14 above the mean magnitude) and run the three-piece kernels: bitwise the three-piece result."""
    args, data, batch, config, store = _arxiv4(outlier=float(2 ** 20))
    acc3, g3, moved3, _, n3 = _run(args, data, batch, config, 3)
    acc2, g2, moved2, _, n2 = _run(args, data, batch, config, 2)
    assert n2[1] == 0 and n2[0] > 0, 'the loose table must take the three-piece kernels: %r' % (n2,)
    assert np.array_equal(acc2, acc3) and moved2 == moved3
    for a, b in zip(g2, g3):
        assert torch.equal(a, b)


def test_two_piece_mode_reruns_a_step_whose_weights_outgrew_their_bound():
    """Fast weights that outgrow 1024 x theta's largest weight inside one inner loop (update_lr = 3e4) do not fit the two-piece planes made
    under the step's weight bound.  The reference's fp32 arithmetic has no such limit (meta.py:126,151), so the step must come back
    as the three-piece kernels compute it -- same accuracies, same decision about the optimiser step -- not as a silently skipped step."""
    args, data, batch, config, store = _arxiv4()
    acc3, g3, moved3, m3, _ = _run(args, data, batch, config, 3, update_lr=3e4)
    acc2, g2, moved2, m2, _ = _run(args, data, batch, config, 2, update_lr=3e4)
    assert m2.last_stats.get('rerun_three_piece', 0) != 0, 'the violation was not detected'
    assert 'rerun_three_piece' not in m3.last_stats
    np.testing.assert_array_equal(acc2, acc3)
    assert moved2 == moved3
    l2, l3 = m2.last_stats['loss_q'], m3.last_stats['loss_q']
    assert (np.isnan(l2) and np.isnan(l3)) or l2 == l3
    # and with an ordinary learning rate nothing is re-run
    _, _, moved, m, n = _run(args, data, batch, config, 2)
    assert 'rerun_three_piece' not in m.last_stats and moved > 0 and n[1] > 0


def test_symmetric_parent_walks_adjacency_once_and_matches_the_two_walk_path(monkeypatch):
    """A parent graph whose out-CSR is element for element its in-CSR (undirected, both directions stored) fills both orientations of the
    induced CSR from one walk of the adjacency lists (extract.hip, ExStore::sym).  Same arrays as the two-walk path (GM_EXTRACT_NO_SYM=1
    at store creation) and as the oracle, on a fixture and at the arxiv shape with sampling."""
    import gmeta_amd
    import gmeta_oracle as orc
    from gmeta_amd import synth
    from gmeta_amd.subgraphs import SubgraphBatch
    fx = Fixture('g1_sampled_h2')

    def arrays(store, seeds, off, h, sn, link):
        B = SubgraphBatch.extract(store, seeds, off, h, sn, 222, link)
        ip, ix = B.csr(); ipt, ixt = B.csr(transposed=True)
        return [np.asarray(a).copy() for a in (B.parent(), ip, ix, ipt, ixt, B.centres_local(), B.sub_off)]
    seeds = fx.z['qry_seeds'].reshape(-1, 3); off = np.arange(fx.T + 1) * fx.z['qry_seeds'].shape[1]
    a_sym = arrays(gmeta_amd.GraphStore(fx.edges, fx.feats), seeds, off, fx.args['h'], fx.args['sample_nodes'], fx.link)
    monkeypatch.setenv('GM_EXTRACT_NO_SYM', '1')
    a_two = arrays(gmeta_amd.GraphStore(fx.edges, fx.feats), seeds, off, fx.args['h'], fx.args['sample_nodes'], fx.link)
    monkeypatch.delenv('GM_EXTRACT_NO_SYM')
    for x, y in zip(a_sym, a_two):
        assert np.array_equal(x, y)
    assert np.array_equal(a_sym[1], a_sym[3]) and np.array_equal(a_sym[2], a_sym[4])      # symmetric: by-source CSR == by-destination CSR
    # arxiv shape: hubs (> EX_BIG_DEG neighbours) and sampling
    np.random.seed(222); random.seed(222)
    args, cfg = synth.make_args('arxiv', task_num=2)
    data = synth.make_dataset(cfg)
    n, src, dst = data['graphs'][0]
    deg = np.bincount(np.asarray(dst), minlength=n)
    hubs = np.argsort(-deg)[:3]
    rng = np.random.default_rng(7)
    cs = np.concatenate([hubs, rng.integers(0, n, 29)]).astype(np.int32)
    seeds = np.stack([np.zeros_like(cs), cs, -np.ones_like(cs)], 1)
    off = np.array([0, len(cs)])
    a_sym = arrays(gmeta_amd.GraphStore(data['graphs'], data['feats']), seeds, off, 2, 1000, False)
    monkeypatch.setenv('GM_EXTRACT_NO_SYM', '1')
    a_two = arrays(gmeta_amd.GraphStore(data['graphs'], data['feats']), seeds, off, 2, 1000, False)
    for x, y in zip(a_sym, a_two):
        assert np.array_equal(x, y)
    ob = orc.extract_batch([orc.Graph(n, src, dst)], seeds, 2, 1000, 222, False)
    assert np.array_equal(a_sym[0], ob.parent) and np.array_equal(a_sym[1], ob.indptr) and np.array_equal(a_sym[2], ob.indices)


def test_two_query_streams_are_bitwise_the_one_stream_step():
    """GM_QUERY_STREAMS=2: the K + 1 query evaluations alternate between two contexts / streams (own activations, own hub counters and
    partial rows).  Same arithmetic per evaluation -> accuracies, losses, meta-gradient and post-Adam weights bitwise those of the default."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    args, data, batch, config, store = _arxiv4()

    def run(streams):
        _lib.check(lib.gm_set_tuning(b'GM_QUERY_STREAMS', streams), 'set_tuning')
        try:
            return _run(args, data, batch, config, 3)
        finally:
            lib.gm_set_tuning(b'GM_QUERY_STREAMS', 0)
    acc1, g1, moved1, m1, _ = run(1)
    acc2, g2, moved2, m2, _ = run(2)
    assert np.array_equal(acc1, acc2) and moved1 == moved2 and moved1 > 0
    assert np.array_equal(m1.last_stats['losses_q'], m2.last_stats['losses_q'])
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


@pytest.mark.parametrize('case', [c for c in WIDE_CASES if c not in NAN_CASES])
@pytest.mark.parametrize('pieces', [3, 2])
def test_finetunning_through_forced_split_kernels_matches_reference(case, pieces):
    """Meta.finetunning (meta.py:175-234) on task 0 of the hidden-128 fixtures with the split kernels forced on: the reference's accuracies."""
    import gmeta_amd
    from hip_util import fixture_meta, make_store
    fx = Fixture(case)
    store = make_store(fx)
    m = fixture_meta(fx)

    def one(tag):
        return gmeta_amd.SubgraphBatch.from_nodes(store, fx.z[tag + '_seeds'][0], [0, fx.z[tag + '_seeds'].shape[1]], fx.replay_lists(tag, 0), fx.link)
    ys = [torch.from_numpy(fx.z['y_spt'][0].astype(np.int64))]; yq = [torch.from_numpy(fx.z['y_qry'][0].astype(np.int64))]
    with forced_split(pieces) as fs:
        accs = m.finetunning([one('spt')], ys, [one('qry')], yq, None, None, None, None, None, None, fx.feats)
        g3, w3, g2, w2, g1 = fs.launches()
    assert (g2 if pieces == 2 else g3) > 0
    np.testing.assert_allclose(accs, fx.z['ft_accs'], atol=1e-6)


@pytest.mark.parametrize('pieces', [3, 2])
def test_centre_only_stores_of_the_last_layer_are_bitwise_the_full_stores(pieces):
    """The last GraphConv's update computes every row and stores only the centre rows (the `h[to_fetch]` of Classifier.forward, learner.py,
    fused into the GEMM epilogue): the head is the only reader of that activation -- the backward pass takes relu' from the bit masks and the
    weight gradient from Z_L.  GM_CENTRE_STORE = 2 (default: every pass), 1 (forward-only passes), 0 (every row stored): accuracies, losses,
    the meta-gradient and the post-Adam weights are bitwise the same -- training step and finetunning, 4-task arxiv shard (split kernels by
    launch size, fused aggregate + GEMM) and a hidden-128 fixture with the split kernels forced on."""
    from gmeta_amd import _lib
    from hip_util import hip_meta_step
    lib = _lib.lib()
    args, data, batch, config, store = _arxiv4()

    def run(on):
        _lib.check(lib.gm_set_tuning(b'GM_CENTRE_STORE', on), 'set_tuning')
        try:
            acc, g, moved, m, n = _run(args, data, batch, config, pieces)
            with torch.no_grad():
                ft = np.asarray(m.finetunning_batch(batch[0], batch[1], batch[2], batch[3]))
            return acc, g, moved, m, ft
        finally:
            lib.gm_set_tuning(b'GM_CENTRE_STORE', 2)
    acc0, g0, moved0, m0, ft0 = run(0)
    for mode in (2, 1):
        acc1, g1, moved1, m1, ft1 = run(mode)
        assert np.array_equal(acc1, acc0) and moved1 == moved0 and moved1 > 0 and np.array_equal(ft1, ft0)
        assert np.array_equal(m1.last_stats['losses_q'], m0.last_stats['losses_q'])
        for a, b in zip(g1, g0):
            assert torch.equal(a, b)
    # a reference fixture (link prediction has two centres per subgraph: g7 is node classification; the forced split kernels take its 128-wide layers)
    fx = Fixture('g7_wide_h2')
    with forced_split(pieces):
        outs = []
        for on in (2, 0):
            _lib.check(lib.gm_set_tuning(b'GM_CENTRE_STORE', on), 'set_tuning')
            try:
                outs.append(hip_meta_step(fx, replay=True))
            finally:
                lib.gm_set_tuning(b'GM_CENTRE_STORE', 2)
    assert np.array_equal(outs[0]['accs'], outs[1]['accs']) and np.array_equal(outs[0]['grad'], outs[1]['grad'])
    for a, b in zip(outs[0]['vars1'], outs[1]['vars1']):
        assert np.array_equal(a, b)


def test_centre_only_stores_under_autograd_classifier():
    """gmeta_amd.Classifier under torch.autograd (`to_fetch=None`: the batch's own centres) with the split kernels forced on: logits and
    parameter gradients bitwise the same with the last layer's activation stored at the centre rows only (default) and at every row."""
    import gmeta_amd
    from gmeta_amd import _lib
    import hip_util as hu
    lib = _lib.lib()
    fx = Fixture('g7_wide_h2')
    store = hu.make_store(fx)
    one = gmeta_amd.SubgraphBatch.from_nodes(store, fx.z['spt_seeds'][0], [0, fx.z['spt_seeds'].shape[1]], fx.replay_lists('spt', 0), fx.link)
    net = gmeta_amd.Classifier(fx.config).cuda()
    with torch.no_grad():
        for p, v in zip(net.parameters(), fx.vars0):
            p.copy_(torch.from_numpy(v))
    outs = []
    with forced_split(3) as fs:
        for on in (2, 0):
            _lib.check(lib.gm_set_tuning(b'GM_CENTRE_STORE', on), 'set_tuning')
            try:
                logits, _ = net(one, None, None)
                w = torch.linspace(-1.0, 1.0, logits.numel(), device='cuda').view_as(logits)
                grads = torch.autograd.grad((logits * w).sum(), list(net.parameters()))
                torch.cuda.synchronize()
                outs.append((logits.detach().clone(), [g.clone() for g in grads]))
            finally:
                lib.gm_set_tuning(b'GM_CENTRE_STORE', 2)
        assert fs.launches()[0] > 0, 'no split GEMM launch'
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('hoist,sparse,cone', [(1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1)])
@pytest.mark.parametrize('case', ['g7_wide_h2', 'g8_wide_scales'])
def test_flagged_schedules_through_forced_split_kernels_match_reference(case, hoist, sparse, cone):
    """The flagged exact schedules (hoisted layer-1 aggregate, row-sparse backward, receptive-field cone) with the three-piece split kernels forced
    onto the hidden-128 fixtures -- and the last layer's centre-only stores active -- against the reference's own outputs."""
    from hip_util import hip_meta_step
    fx = Fixture(case)
    with forced_split(3):
        res = hip_meta_step(fx, replay=True, hoist=hoist, sparse_bwd=sparse, cone=cone)
    np.testing.assert_allclose(res['accs'], fx.z['accs'], atol=1e-6)
    np.testing.assert_allclose(res['stats']['losses_q'], fx.z['loss_q'].mean(0), atol=TOL)
    ref_g = np.concatenate([g.reshape(-1) for g in fx.grad])
    np.testing.assert_allclose(res['grad'], ref_g, atol=TOL, rtol=0)
