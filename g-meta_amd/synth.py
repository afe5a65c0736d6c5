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


def multi_node_dataset(n_graphs, n, m, F0, n_classes, seed=222, label_sets=1):
    """Shared multi-graph node classification (Tissue-PPI shape, SURVEY 8(d) SYN-TISSUE): `n_graphs` undirected PA graphs stored
    in both directions, features ~N(0,1), `label_sets` independent labelings (the reference keeps one label.pkl per task{n}/
    directory, train.py:49-51); names 'g_v' (sdp.py:355-357)."""
    rng = np.random.default_rng(seed)
    graphs, feats, names = [], [], []
    for g in range(n_graphs):
        e = pa_edges(n, m, rng)
        graphs.append((n, np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])))
        feats.append(rng.standard_normal((n, F0), dtype=np.float32))
        names += ['%d_%d' % (g, v) for v in range(n)]
    sets = []
    for _ in range(label_sets):
        lab = rng.integers(0, n_classes, size=n_graphs * n)
        sets.append({'labels': [str(l) for l in lab], 'info': dict(zip(names, lab.tolist()))})
    return {'graphs': graphs, 'feats': feats, 'names': names, 'labels': sets[0]['labels'], 'info': sets[0]['info'], 'label_sets': sets,
            'tables': {'train': (names, sets[0]['labels'])}}


def link_dataset(n_graphs, n, m, F0, seed=222, spt_frac=0.3):
    """Shared multi-graph link prediction (FirstMM-DB shape, SURVEY 8(d) SYN-FIRSTMM) laid out like data_process/link_process.py:
    positives stored once as u -> v with u < v (link_process.py:32-34,45-47), an equal number of negative pairs injected as edges
    (link_process.py:83-85, "following SEAL"), 30 % of each kind are support names and 70 % query names (link_process.py:13,37-41,
    69-74); names 'g_i_j' (sdp.py:358-362), label 1 / 0."""
    rng = np.random.default_rng(seed)
    graphs, feats, info = [], [], {}
    tabs = {'train': ([], []), 'train_spt': ([], []), 'train_qry': ([], [])}
    for g in range(n_graphs):
        e = pa_edges(n, m, rng)
        have = set(map(tuple, e.tolist()))
        neg = rng.integers(0, n, size=(3 * len(e), 2))
        neg = neg[neg[:, 0] != neg[:, 1]]
        keep, seen = [], set()
        for a, b in neg.tolist():
            if (a, b) in have or (b, a) in have or (a, b) in seen or (b, a) in seen:
                continue
            seen.add((a, b)); keep.append((a, b))
            if len(keep) == len(e):
                break
        neg = np.array(keep, np.int64).reshape(-1, 2)
        graphs.append((n, np.concatenate([e[:, 0], neg[:, 0]]), np.concatenate([e[:, 1], neg[:, 1]])))
        feats.append(rng.standard_normal((n, F0), dtype=np.float32))
        for arr, lab in ((e, 1), (neg, 0)):
            spt = np.zeros(len(arr), bool)
            spt[rng.choice(len(arr), int(len(arr) * spt_frac), replace=False)] = True
            for (a, b), s in zip(arr.tolist(), spt.tolist()):
                nm = '%d_%d_%d' % (g, a, b)
                info[nm] = lab
                for key in ('train', 'train_spt' if s else 'train_qry'):
                    tabs[key][0].append(nm); tabs[key][1].append(str(lab))
    return {'graphs': graphs, 'feats': feats, 'info': info, 'tables': tabs}


CONFIGS = {
    # cfg 0: synthetic plumbing case
    'syn0': dict(n=2000, m=3, F0=32, classes=10, hidden=64, h=1, n_way=2, k_spt=1, k_qry=5, task_num=4, update_step=5,
                 update_step_test=10, update_lr=0.01, meta_lr=1e-3, sample_nodes=1000),
    # cfg 1/2: arxiv-ogbn shape (SYN-ARXIV): 169,343 nodes, F0=128, 40 classes, h=2, H=256, 3-way 3-shot 24-qry, T=32, K=10
    'arxiv': dict(n=169343, m=7, F0=128, classes=40, hidden=256, h=2, n_way=3, k_spt=3, k_qry=24, task_num=32, update_step=10,
                  update_step_test=20, update_lr=0.01, meta_lr=1e-3, sample_nodes=1000),
    # cfg 3: Tissue-PPI shape (SYN-TISSUE): Shared, 24 graphs x 2,100 nodes, avg in-degree ~50, F0=50, 2 classes, H=128, 3-shot 10-qry,
    # T=4, K=10 / K_test=10 (test.ipynb:98-103), 10 label sets
    'tissue': dict(kind='multi', n_graphs=24, n=2100, m=25, F0=50, classes=2, hidden=128, h=2, n_way=2, k_spt=3, k_qry=10, task_num=4,
                   update_step=10, update_step_test=10, update_lr=0.01, meta_lr=5e-3, sample_nodes=1000, task_setup='Shared', label_sets=10,
                   eval_tasks=10),
    # cfg 4: FirstMM-DB shape (SYN-FIRSTMM): Shared link prediction, 41 directed graphs x 1,400 nodes, ~2.8k positive + as many injected
    # negative edges each, F0=5, 16-shot 32-qry, T=8, K=10 / K_test=20 (test.ipynb:270-279), head [2, 2H]
    'firstmm': dict(kind='link', n_graphs=41, n=1400, m=2, F0=5, classes=2, hidden=128, h=2, n_way=2, k_spt=16, k_qry=32, task_num=8,
                    update_step=10, update_step_test=20, update_lr=0.01, meta_lr=5e-4, sample_nodes=1000, task_setup='Shared', link=True,
                    eval_tasks=10),
}
WORKLOADS = {'syn0': 'BASELINE configs[0]: synthetic plumbing case', 'arxiv': 'BASELINE configs[1]: arxiv-ogbn shape',
             'tissue': 'BASELINE configs[3]: Tissue-PPI shape', 'firstmm': 'BASELINE configs[4]: FirstMM-DB shape'}


def make_args(cfg, **over):
    c = dict(CONFIGS[cfg]); c.update(over)
    return argparse.Namespace(update_lr=c['update_lr'], meta_lr=c['meta_lr'], n_way=c['n_way'], k_spt=c['k_spt'], k_qry=c['k_qry'],
                              task_num=c['task_num'], update_step=c['update_step'], update_step_test=c['update_step_test'],
                              method='G-Meta', sample_nodes=c['sample_nodes'], link_pred_mode='True' if c.get('link') else 'False',
                              task_setup=c.get('task_setup', 'Disjoint'),
                              h=c['h'], hidden_dim=c['hidden'], hoist_z1=c.get('hoist_z1', 0), serialize=c.get('serialize', 0), sparse_bwd=c.get('sparse_bwd', 0), cone=c.get('cone', 0)), c


def make_config(F0, hidden, h, n_out, link=False):
    """train.py:67-75."""
    config = [('GraphConv', [F0, hidden])] + [('GraphConv', [hidden, hidden])] * (h - 1) + [('Linear', [hidden, n_out])]
    if link:
        config.append(('LinkPred', [True]))
    return config


def make_dataset(c, seed=222):
    """Synthetic data of a CONFIGS entry: {'graphs', 'feats', 'info', 'tables'} (tables as Subgraphs(tables=...) takes them)."""
    kind = c.get('kind', 'single')
    if kind == 'single':
        d = node_dataset(c['n'], c['m'], c['F0'], c['classes'], seed)
        d['tables'] = {'train': (d['names'], d['labels'])}
        return d
    if kind == 'multi':
        return multi_node_dataset(c['n_graphs'], c['n'], c['m'], c['F0'], c['classes'], seed, c.get('label_sets', 1))
    return link_dataset(c['n_graphs'], c['n'], c['m'], c['F0'], seed)


def n_out(c):
    """Output width of the linear head: n_way in Disjoint, #classes in Shared (train.py:58-61)."""
    return c['n_way'] if c.get('task_setup', 'Disjoint') == 'Disjoint' else c['classes']
