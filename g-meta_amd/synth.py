"""Synthetic datasets of the shapes named in BASELINE.json / SURVEY.md section 8(d) (no real dataset ships
with the reference; all generators are seeded with 222 to echo train.py:33-35)."""
import argparse

import numpy as np


def pa_edges(n, m, rng):
    """Preferential-attachment undirected edge list (u < v, deduplicated), fully vectorised: edge k of
    node v = m + k//m copies a uniformly random endpoint among all earlier edge endpoints."""
    K = m * (n - m)
    src = m + np.arange(K, dtype=np.int64) // m
    first = src == m
    hi = np.maximum(2 * m * (src - m), 1)
    ptr = (rng.random(K) * hi).astype(np.int64)            # position in the endpoint array rep[2k]=src[k], rep[2k+1]=tgt[k]
    tgt = np.where(first, np.arange(K) % m, -1)
    cur = ptr.copy()
    todo = np.nonzero(~first)[0]
    while len(todo):
        c = cur[todo]
        even = (c & 1) == 0
        k = c >> 1
        done_even = todo[even]
        tgt[done_even] = src[k[even]]
        odd_idx = todo[~even]; ko = k[~even]
        known = tgt[ko] >= 0
        tgt[odd_idx[known]] = tgt[ko[known]]
        rest = odd_idx[~known]
        cur[rest] = ptr[ko[~known]]
        todo = rest
    e = np.stack([np.minimum(src, tgt), np.maximum(src, tgt)], 1)
    e = e[e[:, 0] != e[:, 1]]
    return np.unique(e, axis=0)


def node_dataset(n, m, F0, n_classes, seed=222, both_directions=True):
    rng = np.random.default_rng(seed)
    e = pa_edges(n, m, rng)
    if both_directions:
        src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    else:
        src, dst = e[:, 0], e[:, 1]
    feat = rng.standard_normal((n, F0), dtype=np.float32)
    labels = rng.integers(0, n_classes, size=n)
    names = ['0_%d' % v for v in range(n)]
    info = dict(zip(names, labels.tolist()))
    return {'graphs': [(n, src, dst)], 'feats': [feat], 'names': names, 'labels': [str(l) for l in labels], 'info': info}


CONFIGS = {
    # cfg 0: synthetic plumbing case
    'syn0': dict(n=2000, m=3, F0=32, classes=10, hidden=64, h=1, n_way=2, k_spt=1, k_qry=5, task_num=4, update_step=5,
                 update_step_test=10, update_lr=0.01, meta_lr=1e-3, sample_nodes=1000),
    # cfg 1/2: arxiv-ogbn shape (SYN-ARXIV): 169,343 nodes, F0=128, 40 classes, h=2, H=256, 3-way 3-shot 24-qry, T=32, K=10
    'arxiv': dict(n=169343, m=7, F0=128, classes=40, hidden=256, h=2, n_way=3, k_spt=3, k_qry=24, task_num=32, update_step=10,
                  update_step_test=20, update_lr=0.01, meta_lr=1e-3, sample_nodes=1000),
}


def make_args(cfg, **over):
    c = dict(CONFIGS[cfg]); c.update(over)
    return argparse.Namespace(update_lr=c['update_lr'], meta_lr=c['meta_lr'], n_way=c['n_way'], k_spt=c['k_spt'], k_qry=c['k_qry'],
                              task_num=c['task_num'], update_step=c['update_step'], update_step_test=c['update_step_test'],
                              method='G-Meta', sample_nodes=c['sample_nodes'], link_pred_mode='False', task_setup='Disjoint',
                              h=c['h'], hidden_dim=c['hidden'], hoist_z1=c.get('hoist_z1', 0), serialize=c.get('serialize', 0), sparse_bwd=c.get('sparse_bwd', 0), cone=c.get('cone', 0)), c


def make_config(F0, hidden, h, n_out, link=False):
    """train.py:67-75."""
    config = [('GraphConv', [F0, hidden])] + [('GraphConv', [hidden, hidden])] * (h - 1) + [('Linear', [hidden, n_out])]
    if link:
        config.append(('LinkPred', [True]))
    return config
