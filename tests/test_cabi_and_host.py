"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/gmeta_hip.h declares (no compute
calls without a GPU), the product path fails loudly without a GPU, and the host-side mirrors behave like the
reference's (task sampling, collate, config parsing, CSR construction, parameter init)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import gmeta_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'gmeta_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gm_[a-z_0-9]+)\s*\(', txt)))


def test_every_declared_symbol_is_exported_and_bound():
    import gmeta_amd  # noqa: F401
    from gmeta_amd import _lib
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'libgmeta_hip.so does not export %s' % n
    assert set(names) == set(_lib.PROTOTYPES), 'ctypes prototypes and header disagree: %s' % (set(names) ^ set(_lib.PROTOTYPES))
    assert _lib.lib().gm_version() >= 100
    m = _lib.make_model([('GraphConv', [128, 256]), ('GraphConv', [256, 256]), ('Linear', [256, 3])])
    assert _lib.lib().gm_model_param_count(ctypes.byref(m)) == 99587        # == the reference's printed count (test.ipynb)
    assert _lib.HParams._fields_[-1][0] == 'cone' and ctypes.sizeof(_lib.HParams) == 32
    assert ctypes.sizeof(_lib.Model) == 4 * (1 + 5 + 2) and ctypes.sizeof(_lib.Seed) == 12


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_path_fails_loudly_without_gpu():
    import gmeta_amd
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        gmeta_amd.GraphStore([(3, [0, 1], [1, 2])], [np.zeros((3, 4), np.float32)])
    net = gmeta_amd.Classifier([('GraphConv', [4, 8]), ('Linear', [8, 2])])
    with pytest.raises((RuntimeError, TypeError)):
        net(object(), None, None)


def test_config_parsing_mirrors_learner():
    from gmeta_amd import _lib
    m = _lib.make_model([('GraphConv', [5, 128]), ('GraphConv', [128, 128]), ('Linear', [128, 2]), ('LinkPred', [True])])
    assert (m.n_gcn, list(m.dims)[:3], m.n_out, m.link_pred) == (2, [5, 128, 128], 2, 1)
    with pytest.raises(ValueError):
        _lib.make_model([('GraphConv', [5, 8]), ('GraphConv', [9, 8]), ('Linear', [8, 2])])
    with pytest.raises(NotImplementedError):
        _lib.make_model([('GraphConv', [5, 8]), ('Attention', [8, 4, 8, 2, 3])])


def test_classifier_parameters_match_reference_layout():
    """learner.py:81-97: vars order/shapes, GraphConv W is [in,out], Linear W is [out,in(*2)], zero biases."""
    import gmeta_amd
    torch.manual_seed(222)
    net = gmeta_amd.Classifier([('GraphConv', [5, 16]), ('GraphConv', [16, 16]), ('Linear', [16, 2]), ('LinkPred', [True])])
    shapes = [tuple(p.shape) for p in net.parameters()]
    assert shapes == [(5, 16), (16,), (16, 16), (16,), (2, 32), (2,)]
    assert all(float(p.detach().abs().sum()) == 0 for p in list(net.parameters())[1::2])
    assert isinstance(net.parameters(), torch.nn.ParameterList)
    w = list(net.parameters())[0]
    bound = np.sqrt(6.0 / (5 + 16))                     # xavier_uniform_ on [in,out]
    assert float(w.abs().max()) <= bound + 1e-6


def test_edges_to_in_csr_matches_oracle():
    from gmeta_amd.graphstore import edges_to_in_csr
    rng = np.random.default_rng(0)
    n, e = 50, 300
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    ip, ix = edges_to_in_csr(n, src, dst)
    G = orc.Graph(n, src, dst)
    assert np.array_equal(ip, G.indptr) and np.array_equal(ix, G.indices)
    with pytest.raises(ValueError):
        edges_to_in_csr(3, [0, 5], [1, 2])


def test_collate_transposes_like_reference():
    from gmeta_amd.subgraphs import collate
    samples = [tuple('t%d_s%d' % (t, s) for s in range(10)) for t in range(3)]
    out = collate(samples)
    assert len(out) == 10 and all(isinstance(x, list) and len(x) == 3 for x in out)
    assert out[4] == ['t0_s4', 't1_s4', 't2_s4']


class _FakeStore:
    pass


def _subgraphs(monkeypatch, setup, link='False', **kw):
    """Subgraphs without a GPU: the store is only stored, task sampling is pure host logic."""
    import gmeta_amd.subgraphs as sg
    monkeypatch.setattr(sg, 'GraphStore', _FakeStore)
    args = argparse.Namespace(sample_nodes=1000, link_pred_mode=link, task_setup=setup)
    return sg.Subgraphs(None, 'train', kw.pop('info'), kw.pop('n_way'), kw.pop('k_shot'), kw.pop('k_query'), kw.pop('batchsz'), args,
                        _FakeStore(), 2, tables=kw.pop('tables'), verbose=False)


def test_disjoint_task_sampler_distribution(monkeypatch):
    """sdp.py:150-182: n_way distinct classes, k_shot + k_query distinct subgraphs per class, spt/qry disjoint."""
    np.random.seed(1)
    names = ['0_%d' % v for v in range(400)]
    labels = [str(v % 8) for v in range(400)]
    info = {n: int(l) for n, l in zip(names, labels)}
    db = _subgraphs(monkeypatch, 'Disjoint', info=info, n_way=3, k_shot=2, k_query=5, batchsz=20, tables={'train': (names, labels)})
    assert len(db) == 20
    for t in range(20):
        spt, qry = db._task_names(t)
        assert len(spt) == 6 and len(qry) == 15 and not set(spt) & set(qry)
        cls_s = [info[n] for n in spt]
        assert len(set(cls_s)) == 3 and all(cls_s.count(c) == 2 for c in set(cls_s))
        assert set(info[n] for n in qry) == set(cls_s)
        raw_s, raw_q = db._task_arrays(t)[2:]
        assert raw_s.tolist() == cls_s
        import random
        random.seed(100 + t)
        ys, yq = db._labels(raw_s, raw_q)
        # the reference's relabelling loop (sdp.py:389-397) on the same Python-RNG state gives the same mapping
        random.seed(100 + t)
        uniq = np.unique(raw_s); random.shuffle(uniq)
        ref_s, ref_q = np.zeros(len(raw_s)), np.zeros(len(raw_q))
        for idx, l in enumerate(uniq):
            ref_s[raw_s == l] = idx; ref_q[raw_q == l] = idx
        assert ys.dtype == torch.int64 and ys.tolist() == ref_s.tolist() and yq.tolist() == ref_q.tolist()
        assert sorted(set(ys.tolist())) == [0, 1, 2] and sorted(set(yq.tolist())) == [0, 1, 2]      # relabelled 0..n_way-1 (sdp.py:389-397)
        m = {}
        for n, y in zip(spt, ys.tolist()):
            assert m.setdefault(info[n], y) == y
        for n, y in zip(qry, yq.tolist()):
            assert m[info[n]] == y


def test_shared_and_linkpred_samplers(monkeypatch):
    np.random.seed(2)
    names = ['%d_%d' % (g, v) for g in range(3) for v in range(60)]
    labels = [str(v % 2) for g in range(3) for v in range(60)]
    info = {n: int(l) for n, l in zip(names, labels)}
    db = _subgraphs(monkeypatch, 'Shared', info=info, n_way=2, k_shot=3, k_query=10, batchsz=10, tables={'train': (names, labels)})
    for t in range(10):
        spt, qry = db._task_names(t)
        assert len({n.split('_')[0] for n in spt + qry}) == 1                  # one graph per task (sdp.py:198)
        ys, yq = db._labels(*db._task_arrays(t)[2:])
        assert ys.tolist() == [info[n] for n in spt]                           # raw labels in Shared (sdp.py:408)
    pn = ['%d_%d_%d' % (g, a, a + 1) for g in range(2) for a in range(40)]
    pl = [str((a // 2) % 2) for g in range(2) for a in range(40)]
    pinfo = {n: int(l) for n, l in zip(pn, pl)}
    tabs = {'train': (pn, pl), 'train_spt': (pn[::2], pl[::2]), 'train_qry': (pn[1::2], pl[1::2])}
    db = _subgraphs(monkeypatch, 'Shared', link='True', info=pinfo, n_way=2, k_shot=4, k_query=6, batchsz=5, tables=tabs)
    for t in range(5):
        spt, qry = db._task_names(t)
        assert len(spt) == 8 and len(qry) == 12
        assert set(spt) <= set(pn[::2]) and set(qry) <= set(pn[1::2])          # separate spt / qry CSVs (sdp.py:35-38)
        seeds = db._seeds(spt)
        assert seeds.shape == (8, 3) and (seeds[:, 2] >= 0).all()


def test_synthetic_generator_is_seeded_and_sane():
    from gmeta_amd import synth
    a = synth.pa_edges(3000, 5, np.random.default_rng(222))
    b = synth.pa_edges(3000, 5, np.random.default_rng(222))
    assert np.array_equal(a, b) and (a[:, 0] < a[:, 1]).all()
    deg = np.bincount(a.reshape(-1), minlength=3000)
    assert deg.min() >= 1 and deg.max() > 10 * deg.mean() / 2          # heavy tail
    args, cfg = synth.make_args('arxiv')
    assert (args.task_num, args.update_step, cfg['hidden'], cfg['F0']) == (32, 10, 256, 128)
    assert synth.make_config(128, 256, 2, 3) == [('GraphConv', [128, 256]), ('GraphConv', [256, 256]), ('Linear', [256, 3])]


def test_hot_kernels_do_not_spill():
    """Compile the two hot translation units with -Rpass-analysis=kernel-resource-usage and require ScratchSize == 0 for
    every specialised kernel (a silent spill made k_wgrad_fast<4,8,2> 15x slower once).  The generic fallback k_wgrad is
    exempt (cold path for odd shapes)."""
    import subprocess
    csrc = os.path.join(ROOT, 'g-meta_amd', 'csrc')
    bad = []
    for unit in ('gemm.hip', 'agg.hip'):
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', os.path.join(csrc, unit), '-o', os.devnull,
                            '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        name = None
        for line in r.stderr.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = m.group(1)
            m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
            if m and int(m.group(1)) > 0 and name and not name.startswith('_Z7k_wgrad6'):
                bad.append((name, int(m.group(1))))
    assert not bad, 'kernels spilling to scratch: %s' % bad
