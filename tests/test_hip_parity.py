"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden fixtures.
Integer/index work bit-exact; floats within the north-star tolerance 1e-4 (written as TOL)."""
import ctypes as C

import numpy as np
import pytest
import torch

import gmeta_oracle as orc
from golden_util import CASES, NAN_CASES, Fixture, call_sizes

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _imports():
    import gmeta_amd  # noqa: F401
    import hip_util
    return hip_util


def _oracle_batches(fx, replay):
    graphs = fx.graphs()
    out = []
    for tag in ('spt', 'qry'):
        bs = []
        for t in range(fx.T):
            rp = fx.replay_lists(tag, t) if replay else None
            bs.append(orc.extract_batch(graphs, fx.z[tag + '_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=rp))
        out.append(bs)
    return graphs, out[0], out[1]


def _cat_csr(batches):
    """Concatenate oracle batches (one per task) the way gm_extract lays a multi-set batch out."""
    ptr, idx, par, cen, norm, sub = [np.zeros(1, np.int64)], [], [], [], [], [np.zeros(1, np.int64)]
    r0 = e0 = 0
    for b in batches:
        ptr.append(b.indptr[1:] + e0); idx.append(b.indices + r0); par.append(b.parent); norm.append(b.norm)
        cen.append(b.centre_rows - b.sub_off[:-1, None]); sub.append(b.sub_off[1:] + r0)
        r0 += b.n; e0 += len(b.indices)
    return (np.concatenate(ptr), np.concatenate(idx), np.concatenate(par), np.concatenate(cen), np.concatenate(norm), np.concatenate(sub))


def _transpose(indptr, indices, n):
    """By-source CSR with destinations ascending (stable) -- what the kernel must produce."""
    dst = np.repeat(np.arange(n), np.diff(indptr))
    order = np.lexsort((dst, indices))
    tp = np.zeros(n + 1, np.int64)
    np.add.at(tp, indices + 1, 1)
    return np.cumsum(tp), dst[order]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('replay', [False, True])
def test_extraction_bit_exact(case, replay):
    """a1-a3: node lists, induced CSR (both orientations), centres, sub offsets == oracle, bit for bit;
    with replay the node lists are the reference's own (so edges == reference's, see test_oracle_golden)."""
    hu = _imports()
    fx = Fixture(case)
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, replay)
    graphs, os_, oq_ = _oracle_batches(fx, replay)
    for hipb, ob in ((S, os_), (Q, oq_)):
        ptr, idx, par, cen, norm, sub = _cat_csr(ob)
        assert hipb.rows == len(par) and hipb.edges == len(idx)
        assert np.array_equal(hipb.parent(), par)
        assert np.array_equal(hipb.sub_off, sub)
        ip, ix = hipb.csr()
        assert np.array_equal(ip, ptr) and np.array_equal(ix, idx)
        tp, tx = hipb.csr(transposed=True)
        rp, rx = _transpose(ptr, idx, len(par))
        assert np.array_equal(tp, rp) and np.array_equal(tx, rx)
        c = hipb._read(8, hipb.subs * hipb.centres, np.int32).reshape(hipb.subs, hipb.centres)
        assert np.array_equal(c, cen)
        nrm = hipb._read(9, hipb.rows, np.float32)
        np.testing.assert_allclose(nrm, norm, rtol=2e-7)
        if not replay:
            # unsampled subgraphs must equal the reference's node SETS (order differs: CPython set order vs ascending)
            tag = 'spt' if hipb is S else 'qry'
            k = 0
            for t in range(fx.T):
                for s in range(fx.z[tag + '_seeds'].shape[1]):
                    ours = par[sub[k]:sub[k + 1]]
                    ref = np.sort(fx.ref_nodes(tag, t, s))
                    g, i, j = fx.z[tag + '_seeds'][t, s]
                    full = orc.linkpred_nodes(graphs[g], i, j) if fx.link else orc.khop_nodes(graphs[g], i, fx.args['h'])
                    if len(full) <= fx.args['sample_nodes']:
                        assert np.array_equal(ours, ref)
                    k += 1


@pytest.mark.parametrize('width', [256, 128, 64, 50, 24, 5, 1])
@pytest.mark.parametrize('transposed', [0, 1])
def test_aggregate_matches_oracle(width, transposed):
    """a6/a12: update_all(copy_src, sum) incl. fused scalings, on both CSR orientations."""
    hu = _imports()
    from gmeta_amd import _lib
    fx = Fixture('g1_sampled_h2')
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    _, _, oq = _oracle_batches(fx, True)
    ptr, idx, par, cen, norm, sub = _cat_csr(oq)
    n = len(par)
    rng = np.random.default_rng(width * 2 + transposed)
    x = rng.standard_normal((n, width)).astype(np.float32)
    s_in, s_out = rng.random(n).astype(np.float32) + 0.5, rng.random(n).astype(np.float32) + 0.5
    if transposed:
        ptr, idx = _transpose(ptr, idx, n)
    ref = orc.agg(ptr, idx.astype(np.int64), x * s_in[:, None]) * s_out[:, None]
    dx, dsi, dso = (torch.from_numpy(a).cuda() for a in (x, s_in, s_out))
    out = torch.empty(n, width, device='cuda')
    _lib.check(_lib.lib().gm_aggregate(Q.handle, transposed, 0, _lib.ptr(dx), width, _lib.ptr(dsi), _lib.ptr(dso), _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5, rtol=1e-5)


def test_aggregate_gather_and_feature_gather():
    """a5: feat[g][ids] gather, alone and fused into the layer-1 aggregate."""
    hu = _imports()
    from gmeta_amd import _lib
    fx = Fixture('g2_shared')
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    _, _, oq = _oracle_batches(fx, True)
    X = np.concatenate([b.features(fx.feats) for b in oq])
    F0 = X.shape[1]
    got = torch.empty(Q.rows, F0, device='cuda')
    _lib.check(_lib.lib().gm_gather_features(Q.handle, _lib.ptr(got), _lib.stream_ptr()))
    assert np.array_equal(got.cpu().numpy(), X)                      # pure copy: bit-exact
    ptr, idx, par, cen, norm, sub = _cat_csr(oq)
    ref = orc.agg(ptr, idx.astype(np.int64), X)
    out = torch.empty(Q.rows, F0, device='cuda')
    _lib.check(_lib.lib().gm_aggregate(Q.handle, 0, 1, None, F0, None, None, _lib.ptr(out), _lib.stream_ptr()))
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('case', [c for c in CASES if c not in NAN_CASES])
def test_classifier_forward_backward_autograd(case):
    """a6/a7: gmeta_amd.Classifier under torch.autograd.grad (the way meta.py:125 uses it) vs the oracle."""
    hu = _imports()
    import gmeta_amd
    fx = Fixture(case)
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    graphs, os_, _ = _oracle_batches(fx, True)
    net = gmeta_amd.Classifier(fx.config).cuda()
    with torch.no_grad():
        for p, v in zip(net.parameters(), fx.vars0):
            p.copy_(torch.from_numpy(v))
    one = gmeta_amd.SubgraphBatch.from_nodes(store, fx.z['spt_seeds'][0], [0, fx.z['spt_seeds'].shape[1]], fx.replay_lists('spt', 0), fx.link)
    logits, _ = net(one, None, None)
    ol, cache = orc.classifier_forward(os_[0], os_[0].features(fx.feats), fx.vars0, fx.config)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ol, atol=TOL, rtol=0)
    w = torch.randn_like(logits)
    grads = torch.autograd.grad((logits * w).sum(), list(net.parameters()))
    og = orc.classifier_backward(os_[0], fx.vars0, fx.config, cache, w.cpu().numpy())
    for a, b in zip(grads, og):
        np.testing.assert_allclose(a.cpu().numpy(), b, atol=TOL, rtol=1e-4)
    # explicit features + explicit to_fetch (the reference's calling convention, meta.py:122) give the same logits
    feats = torch.from_numpy(os_[0].features(fx.feats)).cuda()
    tf = torch.from_numpy(np.asarray(one.centres_local(), np.int64)).cuda()
    logits2, _ = net(one, tf, feats)
    np.testing.assert_allclose(logits2.detach().cpu().numpy(), logits.detach().cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('hoist,sparse,cone', [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1)])
def test_meta_step_matches_reference(case, hoist, sparse, cone):
    """a8-a10 against the reference's own outputs (golden): accs, theta.grad, post-Adam weights, NaN skip -- for the
    default schedule and for the flagged exact variants (hoisted layer-1 aggregate, row-sparse backward, receptive-field
    schedule with and without the hoist)."""
    hu = _imports()
    fx = Fixture(case)
    res = hu.hip_meta_step(fx, replay=True, hoist=hoist, sparse_bwd=sparse, cone=cone)
    if case in NAN_CASES:
        # meta.py:163-164 `if torch.isnan(loss_q): pass`: the guard runs on the device (gm_meta_finish sets found_inf, the fused
        # Adam skips the update and rolls its step counter back) -- weights bit-identical, optimiser state untouched
        assert np.isnan(res['stats']['loss_q'])
        for a, b in zip(res['vars1'], fx.vars1):
            assert np.array_equal(a, b)
        for st in res['meta'].meta_optim.state.values():
            assert float(st['step']) == 0.0 and float(st['exp_avg'].abs().max()) == 0.0 and float(st['exp_avg_sq'].abs().max()) == 0.0
        return
    np.testing.assert_allclose(res['accs'], fx.z['accs'], atol=1e-6)
    np.testing.assert_allclose(res['stats']['losses_q'], fx.z['loss_q'].mean(0), atol=TOL)
    ref_g = np.concatenate([g.reshape(-1) for g in fx.grad])
    np.testing.assert_allclose(res['grad'], ref_g, atol=TOL, rtol=0)
    for a, b, g in zip(res['vars1'], fx.vars1, fx.grad):
        m = np.abs(g) > 1e-5           # Adam's first step is sign(g)*lr: only comparable where g is well away from 0
        np.testing.assert_allclose(a[m], b[m], atol=TOL, rtol=0)


@pytest.mark.parametrize('case', [c for c in CASES if c not in NAN_CASES])
def test_finetunning_matches_reference(case):
    """a11 / G4."""
    hu = _imports()
    fx = Fixture(case)
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    m = hu.fixture_meta(fx)
    before = [p.detach().clone() for p in m.net.parameters()]
    ys = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_spt']]
    yq = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_qry']]
    one_s = gmeta_one(fx, store, 'spt'); one_q = gmeta_one(fx, store, 'qry')
    accs = m.finetunning([one_s], ys[:1], [one_q], yq[:1], None, None, None, None, None, None, fx.feats)
    np.testing.assert_allclose(accs, fx.z['ft_accs'], atol=1e-6)
    for a, b in zip(before, m.net.parameters()):
        assert torch.equal(a, b)                                    # finetunning never touches self.net (meta.py:181)
    allacc = m.finetunning_batch(S.views(), ys, Q.views(), yq)     # every task in one call
    assert allacc.shape == (fx.T, fx.K_test + 1)
    np.testing.assert_allclose(allacc[0], fx.z['ft_accs'], atol=1e-6)


def test_deepcopy_snapshot_like_train_py():
    """train.py:87,125-127 keeps the best model as copy.deepcopy(maml): the copy must be independent and usable."""
    import copy
    hu = _imports()
    fx = Fixture('g2_shared')
    res = hu.hip_meta_step(fx, replay=True)
    m = res['meta']
    snap = copy.deepcopy(m)
    for a, b in zip(m.net.parameters(), snap.net.parameters()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr()
    ys = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_spt']]
    yq = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_qry']]
    S, Q = res['S'], res['Q']
    before = [p.detach().clone() for p in snap.net.parameters()]
    a1 = m(S.views(), ys, Q.views(), yq, None, None, None, None, None, None, fx.feats)         # second step on the original
    for a, b in zip(before, snap.net.parameters()):
        assert torch.equal(a, b)                                                                # snapshot untouched
    f1 = snap.finetunning_batch(S.views(), ys, Q.views(), yq)
    f2 = copy.deepcopy(snap).finetunning_batch(S.views(), ys, Q.views(), yq)
    assert np.array_equal(f1, f2) and np.isfinite(a1).all()


def gmeta_one(fx, store, tag):
    import gmeta_amd
    return gmeta_amd.SubgraphBatch.from_nodes(store, fx.z[tag + '_seeds'][0], [0, fx.z[tag + '_seeds'].shape[1]], fx.replay_lists(tag, 0), fx.link)


def test_proto_losses_match_oracle():
    """a8/a9 through their own C entry points."""
    hu = _imports()
    from gmeta_amd import _lib
    fx = Fixture('g2_shared')
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    rng = np.random.default_rng(0)
    C_ = 4
    ls, lq = rng.standard_normal((S.subs, C_)).astype(np.float32), rng.standard_normal((Q.subs, C_)).astype(np.float32)
    ys, yq = fx.z['y_spt'].reshape(-1).astype(np.int32), fx.z['y_qry'].reshape(-1).astype(np.int32)
    T, k = fx.T, fx.args['k_spt']
    Ss, Sq = S.subs // T, Q.subs // T
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    loss, acc = torch.empty(T, device='cuda'), torch.empty(T, device='cuda')
    protos = torch.empty(T, 2, C_, device='cuda'); dl = torch.empty(S.subs, C_, device='cuda')
    dls = d(ls)
    _lib.check(_lib.lib().gm_proto_loss_spt(S.handle, _lib.ptr(dls), C_, _lib.ptr(ys), k, _lib.ptr(loss), _lib.ptr(acc), _lib.ptr(protos), _lib.ptr(dl), _lib.stream_ptr()))
    lossq, accq = torch.empty(T, device='cuda'), torch.empty(T, device='cuda')
    dq = torch.empty(Q.subs, C_, device='cuda'); dp = torch.empty(T, 2, C_, device='cuda')
    dlq = d(lq)
    _lib.check(_lib.lib().gm_proto_loss_qry(Q.handle, _lib.ptr(dlq), C_, _lib.ptr(yq), _lib.ptr(protos), 2, _lib.ptr(lossq), _lib.ptr(accq), _lib.ptr(dq), _lib.ptr(dp), _lib.stream_ptr()))
    torch.cuda.synchronize()
    for t in range(T):
        a, b = ls[t * Ss:(t + 1) * Ss], lq[t * Sq:(t + 1) * Sq]
        l, ac, pr, g = orc.proto_loss_spt(a, ys[t * Ss:(t + 1) * Ss], k)
        np.testing.assert_allclose(loss[t].item(), l, atol=1e-5); np.testing.assert_allclose(acc[t].item(), ac, atol=1e-6)
        np.testing.assert_allclose(protos[t].cpu().numpy(), pr, atol=1e-6)
        np.testing.assert_allclose(dl[t * Ss:(t + 1) * Ss].cpu().numpy(), g, atol=1e-5)
        l2, ac2, g2, p2 = orc.proto_loss_qry(b, yq[t * Sq:(t + 1) * Sq], pr, need_grad=True)
        np.testing.assert_allclose(lossq[t].item(), l2, atol=1e-5); np.testing.assert_allclose(accq[t].item(), ac2, atol=1e-6)
        np.testing.assert_allclose(dq[t * Sq:(t + 1) * Sq].cpu().numpy(), g2, atol=1e-5)
        np.testing.assert_allclose(dp[t].cpu().numpy(), p2, atol=1e-5)


def test_error_conventions():
    """8(b): update_step < 2 raises early; unequal query counts raise; link/non-link mismatch raises."""
    hu = _imports()
    fx = Fixture('g0_disjoint_h1')
    store = hu.make_store(fx)
    S, Q = hu.fixture_batches(fx, store, True)
    m = hu.fixture_meta(fx)
    ys = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_spt']]
    yq = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_qry']]
    m.update_step = 1
    with pytest.raises(ValueError):
        m(S.views(), ys, Q.views(), yq, None, None, None, None, None, None, fx.feats)
    m.update_step = 3
    bad = [y.clone() for y in yq]; bad[0][0] = 1 - bad[0][0]
    with pytest.raises(ValueError):
        m(S.views(), ys, Q.views(), bad, None, None, None, None, None, None, fx.feats)
