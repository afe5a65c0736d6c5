"""Next rows N2/N3: the on-disk adapter round-trips on CPU; the train.py-compatible driver runs end to end on a small
synthetic G-Meta data directory on the GPU and learns a separable task."""
import os
import pickle

import numpy as np
import pytest


def _dataset(tmp, n=600, F0=16, classes=6, seed=0):
    from gmeta_amd import datadir, synth
    rng = np.random.default_rng(seed)
    per = n // classes
    lab = np.repeat(np.arange(classes), per)
    blocks = [synth.pa_edges(per, 4, rng) + c * per for c in range(classes)]        # homophilous: a PA graph inside every class ...
    cross = rng.integers(0, n, size=(n // 10, 2))                                     # ... plus a few random cross-class edges
    e = np.concatenate(blocks + [cross[cross[:, 0] != cross[:, 1]]])
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    proto = rng.standard_normal((classes, F0)).astype(np.float32) * 2
    feat = (proto[lab] + 0.3 * rng.standard_normal((n, F0))).astype(np.float32)       # class-separable features
    names = np.array(['0_%d' % v for v in range(n)])
    info = {nm: int(l) for nm, l in zip(names, lab)}
    cls_split = {'train': [0, 1, 2], 'val': [3, 4, 5], 'test': [3, 4, 5]}                # Disjoint label sets per split
    splits = {k: (names[np.isin(lab, c)].tolist(), [str(x) for x in lab[np.isin(lab, c)]]) for k, c in cls_split.items()}
    datadir.write_datadir(str(tmp), [(n, src, dst)], [feat], info, splits)
    return (n, src, dst), feat, info


def test_datadir_roundtrip(tmp_path, monkeypatch):
    import sys
    import gmeta_amd  # noqa: F401
    monkeypatch.setitem(sys.modules, 'dgl', None)          # the test-only DGL restatement must not satisfy the product's import
    from gmeta_amd import datadir
    (n, src, dst), feat, info = _dataset(tmp_path)
    g = datadir.load_graphs(str(tmp_path))
    assert g[0][0] == n and np.array_equal(g[0][1], src) and np.array_equal(g[0][2], dst)
    f = datadir.load_features(str(tmp_path))
    assert len(f) == 1 and np.array_equal(f[0], feat)
    assert datadir.load_labels(str(tmp_path)) == info
    rows = open(os.path.join(str(tmp_path), 'train.csv')).read().splitlines()
    assert rows[0] == ',name,label' and rows[1].split(',')[1].startswith('0_')
    os.remove(os.path.join(str(tmp_path), 'graph_csr.npz'))
    with open(os.path.join(str(tmp_path), 'graph_dgl.pkl'), 'wb') as fh:
        pickle.dump([], fh)
    with pytest.raises(RuntimeError, match='DGL'):
        datadir.load_graphs(str(tmp_path))


@pytest.mark.gpu
def test_train_driver_end_to_end(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import train as drv
    _dataset(tmp_path)
    args = drv.parse(['--data_dir', str(tmp_path) + '/', '--task_setup', 'Disjoint', '--epoch', '2', '--n_way', '3', '--k_spt', '2',
                      '--k_qry', '6', '--task_num', '4', '--update_step', '3', '--update_step_test', '4', '--update_lr', '0.05',
                      '--meta_lr', '0.01', '--hidden_dim', '32', '--batchsz', '40', '--h', '2', '--eval_tasks', '10',
                      '--train_result_report_steps', '5'])
    res = drv.main(args)
    assert res['test_acc'] > 0.6, res            # 3-way chance is 0.33; features are class-separable
