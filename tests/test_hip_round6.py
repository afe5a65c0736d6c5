"""GPU (-m gpu): round-6 parity cases.

* The differentiated passes of the dense schedule (every support pass, the last query pass) take the fused aggregate + GEMM too: Z_l is written at the rows of
  three or more sources only and the weight gradient's split kernel forms the other rows from the batch's per-row source table
  (k_wgrad_split<., ., 3, true>), in the aggregate kernel's fma order -- so the whole meta-step is BITWISE the step with GM_FUSE_DIFF = 0, and on the
  reference-generated fixtures (split kernels forced onto them) it equals the reference's own outputs (learner.py:41-47, meta.py:122-171).
* 100 two-stream meta-steps of the 8-task shape from identical state are bitwise identical (tools/repro_stress.py's check, the one that exposed the
  round-5 LDS-DMA race: determinism is the only net under kernels that hand-roll their memory ordering).
* Batch build: the meta-batch built with the support batch on a helper thread / stream is the batch built on one thread; builder pools deliver
  meta-batches in order with the same label draws."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

from golden_util import WIDE_CASES, Fixture

pytestmark = pytest.mark.gpu


class tuning:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        from gmeta_amd import _lib
        self.lib = _lib.lib()
        self.prev = {k: self.lib.gm_get_tuning(k.encode()) for k in self.kv}
        for k, v in self.kv.items():
            _lib.check(self.lib.gm_set_tuning(k.encode(), v), 'set_tuning')
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            self.lib.gm_set_tuning(k.encode(), v)
        return False


@pytest.fixture(scope='module')
def arxiv8():
    import gmeta_amd
    from gmeta_amd import synth
    np.random.seed(222); random.seed(222); torch.manual_seed(222)
    T = 8
    args, cfg = synth.make_args('arxiv', task_num=T)
    data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
    store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
    db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=3 * T, args=args, adjs=store, h=2,
                             tables={'train': (data['names'], data['labels'])}, verbose=False)
    return dict(args=args, cfg=cfg, data=data, store=store, db=db, T=T, batch=db.get_batch(list(range(T))))


def _meta(w, update_step=3, **kw):
    import argparse
    import gmeta_amd
    from gmeta_amd import synth
    a = argparse.Namespace(**vars(w['args']))
    for k, v in kw.items():
        setattr(a, k, v)
    a.update_step = update_step
    torch.manual_seed(222)
    cfg = w['cfg']
    return gmeta_amd.Meta(a, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')


def _step(m, batch):
    grads = {}
    orig = m.meta_optim.step
    m.meta_optim.step = lambda *a, **k: grads.setdefault('g', torch.cat([p.grad.reshape(-1) for p in m.net.parameters()]).clone())
    accs = m(*batch, None)
    m.meta_optim.step = orig
    return accs, grads['g'], np.asarray(m.last_stats['losses_q']).copy()


def _launch_counts(lib):
    out = []
    for cat in (0, 4, 5):          # aggregate, split GEMM, split weight gradient
        ms, n, w = C.c_double(), C.c_int64(), C.c_int64()
        lib.gm_profile_read(cat, C.byref(ms), C.byref(n), C.byref(w))
        out.append((int(n.value), int(w.value)))
    return out


@pytest.mark.parametrize('tasks', [8, 3, 1])
def test_fused_differentiated_passes_are_bitwise_the_unfused_ones(arxiv8, tasks):
    """GM_FUSE_DIFF 1 (default) vs 0 with everything else equal: accuracies, every query loss and the meta-gradient bitwise; the fused step's
    aggregate launches move fewer algorithmic bytes (the support passes' and the last query pass' full launches became partial ones)."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    b = arxiv8['batch'] if tasks == 8 else arxiv8['db'].get_batch(list(range(tasks)))
    out = []
    with tuning(GM_AGG_STREAM=0):          # (the stream kernel sums a hub row's parts in another association than the window kernel: compare like with like)
        for diff in (0, 1):                # (2, the default, is 1 wherever the stream aggregate does not run: here)
            with tuning(GM_FUSE_DIFF=diff):
                lib.gm_profile_enable(1)
                m = _meta(arxiv8, serialize=1 if tasks == 1 else 0)
                a, g, l = _step(m, b)
                torch.cuda.synchronize()
                out.append((a, g, l, _launch_counts(lib)))
                lib.gm_profile_enable(0)
    (a0, g0, l0, c0), (a1, g1, l1, c1) = out
    assert np.array_equal(a0, a1) and np.array_equal(l0, l1) and torch.equal(g0, g1)
    assert c0[0][0] == c1[0][0] and c1[0][1] < c0[0][1], (c0, c1)          # same number of aggregate launches, fewer bytes
    assert c0[2] == c1[2]                                                  # the same weight gradients on the split kernel


@pytest.mark.parametrize('case', WIDE_CASES)
def test_fused_differentiated_passes_equal_the_reference_on_its_fixtures(case):
    """The reference's own outputs (accs, losses, theta.grad) with the split kernels -- and with them the table-formed weight gradient -- forced onto the
    hidden-128 fixtures (ragged subgraphs, sampled h = 2, three graphs of different feature scales, inf features), against the same run with GM_FUSE_DIFF = 0."""
    from hip_util import hip_meta_step
    fx = Fixture(case)
    res = []
    with tuning(GM_GEMM_SPLIT_MIN_TILES=0, GM_WGRAD_SPLIT_MIN_CHUNKS=0):
        for diff in (1, 0):
            with tuning(GM_FUSE_DIFF=diff):
                res.append(hip_meta_step(fx, replay=True))
    for k in ('accs', 'grad'):
        assert np.array_equal(np.asarray(res[0][k]), np.asarray(res[1][k]), equal_nan=True), k
    if not np.isnan(np.asarray(fx.z['accs'], np.float64)).any():
        np.testing.assert_allclose(res[0]['accs'], fx.z['accs'], atol=1e-6)
    if fx.grad is not None:                 # (the NaN-skip fixture records no theta.grad: the optimiser step never ran, meta.py:163)
        rg = np.concatenate([g.reshape(-1) for g in fx.grad])
        ok = np.isfinite(rg)
        np.testing.assert_allclose(np.asarray(res[0]['grad'])[ok], rg[ok], atol=1e-4, rtol=0)


def test_hundred_two_stream_steps_from_identical_state_are_bitwise_identical(arxiv8):
    """tools/repro_stress.py in the suite: the 8-task two-stream shape, the same state every time, 100 meta-steps -- accuracies, losses and meta-gradient of every
    one bitwise those of the first (the check that exposed the round-5 race; ~1 launch in 500 differed then)."""
    b = arxiv8['batch']
    a0, g0, l0 = _step(_meta(arxiv8), b)
    bad = 0
    for _ in range(99):
        a, g, l = _step(_meta(arxiv8), b)
        bad += not (np.array_equal(a0, a) and np.array_equal(l0, l) and torch.equal(g0, g))
    assert bad == 0, '%d of 99 repeated steps differ from the first' % bad


@pytest.mark.parametrize('mode', ['serial', 'threads'])
def test_joint_build_of_both_batches_is_the_two_separate_builds(arxiv8, mode):
    """Subgraphs.get_batch builds the support and the query batch of a meta-batch in ONE gm_extract_pair call (one launch of each extraction kernel over all
    subgraphs, one round trip for both finalisations): same CSRs, parents, centres, relabelled targets as two gm_extract calls -- one after the other, or
    with the support batch on a helper thread / stream -- and a meta-step over either is bitwise the same."""
    db = arxiv8['db']
    idx = list(range(arxiv8['T'], 2 * arxiv8['T']))
    st = random.getstate()
    b2 = db.get_batch(idx)
    random.setstate(st)
    os.environ['GMETA_EXTRACT_MODE'] = mode
    try:
        b1 = db.get_batch(idx)
    finally:
        del os.environ['GMETA_EXTRACT_MODE']
    for side in (0, 2):
        x, y = b1[side][0].view_of, b2[side][0].view_of
        assert (x.rows, x.edges, x.subs, x.sets) == (y.rows, y.edges, y.subs, y.sets)
        assert np.array_equal(x.parent(), y.parent()) and np.array_equal(x.sub_off, y.sub_off)
        assert np.array_equal(x._read(8, x.subs * x.centres, np.int32), y._read(8, y.subs * y.centres, np.int32))
        assert np.array_equal(x._read(9, x.rows, np.float32), y._read(9, y.rows, np.float32))
        for tr in (False, True):
            assert all(np.array_equal(p, q) for p, q in zip(x.csr(tr), y.csr(tr)))
    for slot in (1, 3, 4, 5):
        assert all(torch.equal(p, q) for p, q in zip(b1[slot], b2[slot]))
    assert b1[8] == b2[8] and b1[9] == b2[9]
    r1, r2 = _step(_meta(arxiv8), b1), _step(_meta(arxiv8), b2)
    assert np.array_equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1]) and np.array_equal(r1[2], r2[2])


@pytest.mark.parametrize('case', ['g2_shared', 'g3_linkpred', 'g1_h3'])
def test_joint_build_on_reference_fixtures(case):
    """gm_extract_pair on ragged multi-graph, link-prediction and h = 3 fixtures: each of the two batches equals its own gm_extract build."""
    from hip_util import fixture_batches, make_store
    from gmeta_amd.subgraphs import SubgraphBatch
    fx = Fixture(case)
    store = make_store(fx)
    S, Q = fixture_batches(fx, store, replay=False)
    sp, sq = fx.z['spt_seeds'], fx.z['qry_seeds']
    T = sp.shape[0]
    S2, Q2 = SubgraphBatch.extract_pair(store, sp.reshape(-1, 3), np.arange(T + 1) * sp.shape[1], sq.reshape(-1, 3), np.arange(T + 1) * sq.shape[1],
                                        fx.args['h'], fx.args['sample_nodes'], 222, fx.link)
    for x, y in ((S, S2), (Q, Q2)):
        assert (x.rows, x.edges, x.subs, x.sets) == (y.rows, y.edges, y.subs, y.sets)
        assert np.array_equal(x.parent(), y.parent()) and np.array_equal(x.sub_off, y.sub_off)
        assert np.array_equal(x._read(8, x.subs * x.centres, np.int32), y._read(8, y.subs * y.centres, np.int32))
        for tr in (False, True):
            assert all(np.array_equal(p, q) for p, q in zip(x.csr(tr), y.csr(tr)))


def test_builder_pool_delivers_in_order_with_the_same_label_draws(arxiv8):
    """Subgraphs.batches(workers=3): meta-batches arrive in list order, and the Disjoint relabelling (a global-RNG shuffle per task, sdp.py:390-397) draws
    exactly what the sequential get_batch loop draws -- the draws happen on ONE host thread, in meta-batch order, whichever builder takes the batch."""
    db, T = arxiv8['db'], arxiv8['T']
    lists = [list(range(k * T, (k + 1) * T)) for k in range(3)] + [list(range(0, 3))]
    st = random.getstate()
    seq = [db.get_batch(i) for i in lists]
    random.setstate(st)
    for kw in (dict(prefetch=2, workers=3, cone_layers=2), dict(prefetch=1, workers=1), dict(prefetch=2, workers=1, cone_layers=2)):
        random.setstate(st)
        par = list(db.batches(lists, **kw))
        assert len(par) == len(seq)
        for s, p in zip(seq, par):
            assert np.array_equal(s[0][0].view_of.parent(), p[0][0].view_of.parent()) and np.array_equal(s[2][0].view_of.parent(), p[2][0].view_of.parent())
            assert all(torch.equal(x, y) for x, y in zip(s[1], p[1])) and all(torch.equal(x, y) for x, y in zip(s[3], p[3]))
            assert all(torch.equal(x, y) for x, y in zip(s[4], p[4])) and all(torch.equal(x, y) for x, y in zip(s[5], p[5]))       # centre tables (host copy of the build's round trip)


def _cone_tables(lib, b, L):
    from gmeta_amd import _lib
    ok = C.c_int32(); nrows = (C.c_int64 * (L + 1))(); nedges = (C.c_int64 * (L + 1))()
    _lib.check(lib.gm_batch_cone_dims(b.handle, L, C.byref(ok), nrows, nedges), 'cone_dims')
    out = [ok.value, list(nrows), list(nedges)]
    if not ok.value:
        return out
    for l in range(L + 1):
        n_lo = nrows[l - 1] if l else 0
        for what, n in ((0, nrows[l]), (1, nrows[l] + 1 if l else 0), (2, nedges[l] if l else 0), (3, n_lo + 1 if l else 0), (4, nedges[l] if l else 0), (5, b.sets + 1)):
            a = np.empty(max(int(n), 0), np.int32)
            if a.size:
                _lib.check(lib.gm_batch_cone_read(b.handle, L, l, what, a.ctypes.data, a.nbytes), 'cone_read')
            out.append(a)
    return out


def test_receptive_field_tables_of_both_batches_in_one_call(arxiv8):
    """gm_batch_prepare_cone_pair (one pair of host round trips for the support AND the query batch) builds the tables of two gm_batch_prepare_cone
    calls, and a step over them is bitwise the step over those."""
    import gmeta_amd
    from gmeta_amd import _lib
    lib = _lib.lib()
    db, T, L = arxiv8['db'], arxiv8['T'], arxiv8['cfg']['h']
    st = random.getstate()
    one = db.get_batch(list(range(T)))
    random.setstate(st)
    two = db.get_batch(list(range(T)))
    for x in (one[0][0].view_of, one[2][0].view_of):
        _lib.check(lib.gm_batch_prepare_cone(x.handle, L, _lib.stream_ptr()), 'prepare_cone')
    _lib.check(lib.gm_batch_prepare_cone_pair(two[0][0].view_of.handle, two[2][0].view_of.handle, L, _lib.stream_ptr()), 'prepare_cone_pair')
    _lib.check(lib.gm_batch_prepare_cone_pair(two[0][0].view_of.handle, two[2][0].view_of.handle, L, _lib.stream_ptr()), 'prepare_cone_pair')      # cached: a no-op
    for k in (0, 2):
        ta, tb = _cone_tables(lib, one[k][0].view_of, L), _cone_tables(lib, two[k][0].view_of, L)
        assert ta[:3] == tb[:3] and ta[0] == 1
        assert all(np.array_equal(x, y) for x, y in zip(ta[3:], tb[3:]))
    a1, g1, l1 = _step(_meta(arxiv8, cone=1), one)
    a2, g2, l2 = _step(_meta(arxiv8, cone=1), two)
    assert np.array_equal(a1, a2) and torch.equal(g1, g2) and np.array_equal(l1, l2)
    assert lib.gm_batch_prepare_cone_pair(one[0][0].view_of.handle, one[0][0].view_of.handle, L, None) != 0      # two distinct batches


@pytest.mark.parametrize('case', ['g1_sampled_h2', 'g2_shared', 'g3_linkpred', 'g1_h3'])
def test_sixteen_bit_prefix_words_build_the_same_batches(case):
    """The extraction kernels keep per-word prefix counts of the membership bitmap in 16 bits (four resident workgroups per CU instead of three) and, with two
    hops, drop the BFS's `expanded` bitmap: node lists, both CSRs, centres and norms are those of the 32-bit build (GM_EXTRACT_PREF16 = 0), on sampled h = 2,
    multi-graph, link-prediction and h = 3 fixtures (the last keeps the expanded bitmap)."""
    from hip_util import fixture_batches, make_store
    fx = Fixture(case)
    store = make_store(fx)
    out = []
    for p16 in (1, 0):
        with tuning(GM_EXTRACT_PREF16=p16):
            out.append(fixture_batches(fx, store, replay=False))
    for x, y in zip(out[0], out[1]):
        assert (x.rows, x.edges, x.subs) == (y.rows, y.edges, y.subs)
        assert np.array_equal(x.parent(), y.parent()) and np.array_equal(x.sub_off, y.sub_off)
        assert np.array_equal(x._read(8, x.subs * x.centres, np.int32), y._read(8, y.subs * y.centres, np.int32))
        for tr in (False, True):
            assert all(np.array_equal(p, q) for p, q in zip(x.csr(tr), y.csr(tr)))


_SLAB_SCRIPT = r'''
import hashlib, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.environ['GM_REPO'])
import gmeta_amd
from gmeta_amd import synth
np.random.seed(222); random.seed(222); torch.manual_seed(222)
T = 4
args, cfg = synth.make_args('arxiv', task_num=T)
args.update_step = 2
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=6 * T, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
m = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
m.cone = int(os.environ.get('SLAB_CONE', '0'))
h = hashlib.sha256()
lists = [list(range(k * T, (k + 1) * T)) for k in range(6)]
for b in db.batches(lists + lists, prefetch=2, cone_layers=cfg['h'] if m.cone else 0):       # batches of different sizes come and go
    accs = m(*b, None)
    h.update(np.asarray(accs, np.float64).tobytes())
    h.update(torch.cat([p.detach().reshape(-1) for p in m.net.parameters()]).cpu().numpy().tobytes())
torch.cuda.synchronize()
print('HASH', h.hexdigest())
'''


@pytest.mark.parametrize('cone', [0, 1])
def test_slab_cache_limits_do_not_change_results(tmp_path, cone):
    """Dropped batches go to the process-level slab cache (csrc/common.hip) instead of hipFreeAsync; a released slab is reused by a later build behind an event.
    Twelve prefetched meta-steps give bitwise the same accuracies and weights with the cache at its default size, with a 1-MiB cache (every release evicts: the
    hipFreeAsync path, with the evicted slab's event waited for) and with the cache off."""
    import subprocess
    import sys
    script = tmp_path / 'slab.py'
    script.write_text(_SLAB_SCRIPT)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mb in ('default', '1', '0'):
        env = dict(os.environ, GM_REPO=repo, SLAB_CONE=str(cone))
        env.pop('GM_SLAB_CACHE_MB', None)
        if mb != 'default':
            env['GM_SLAB_CACHE_MB'] = mb
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mb] = [l for l in r.stdout.splitlines() if l.startswith('HASH')][0]
    assert out['default'] == out['1'] == out['0'], out
