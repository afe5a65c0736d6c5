#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the reference-order sampling replay.
Runs in the BUILD CONTAINER only (imports the reference's subgraph_data_processing.py unmodified against oracle/dgl_shim).

The reference's task sampler and subgraph sampler draw from the GLOBAL numpy / python RNGs (sdp.py:150-292, 312-314,
337-339) and iterate CPython sets (sdp.py:303,306,311), so which nodes survive `np.random.choice` depends on the whole
call history.  Each fixture records, for seeds (222, 222): the CSV tables, the task name lists the reference drew, and
for every task in visiting order the node list of every subgraph (reference order) plus the relabelled targets.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_replay_golden.py
"""
import sys
sys.dont_write_bytecode = True
import os, json, random, tempfile, warnings  # noqa: E401
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings('ignore', category=SyntaxWarning)
import make_golden as mg                      # noqa: E402  (sets sys.path for the shim + reference, provides the graph helpers)
sdp = mg.sdp

OUT = mg.OUT


def dump(name, graphs, info, csvs, args, T, passes=2):
    torch.manual_seed(222); np.random.seed(222); random.seed(222)          # train.py:33-35 (+random)
    root = tempfile.mkdtemp(prefix='gmeta_replay_') + '/'
    for fn, (nm, lb) in csvs.items():
        mg.write_csv(root + fn, nm, lb)
    G = [mg.make_graph(*g) for g in graphs]
    db = sdp.Subgraphs(root, 'train', info, n_way=args.n_way, k_shot=args.k_spt, k_query=args.k_qry, batchsz=T, args=args, adjs=G, h=args.h)
    out = {'case': name, 'T': T, 'n_graphs': len(graphs), 'args': json.dumps(vars(args)), 'csv_files': json.dumps(sorted(csvs))}
    for k, (n, s, d) in enumerate(graphs):
        out['g%d_n' % k] = n; out['g%d_src' % k] = np.asarray(s, np.int32); out['g%d_dst' % k] = np.asarray(d, np.int32)
    for fn, (nm, lb) in csvs.items():
        out['csv_%s_names' % fn] = np.array(nm); out['csv_%s_labels' % fn] = np.array(lb)
    out['info_names'] = np.array(list(info)); out['info_labels'] = np.array([info[k] for k in info], np.int64)
    out['spt_names'] = np.array([[item for sub in db.support_x_batch[t] for item in sub] for t in range(T)])
    out['qry_names'] = np.array([[item for sub in db.query_x_batch[t] for item in sub] for t in range(T)])
    nodes, sampled, ys, yq = [], [], [], []
    for p in range(passes):                    # the second pass hits the memo (sdp.py:296-297): same subgraphs, no RNG draw
        for t in range(T):
            tup = db[t]
            ys.append(tup[1].numpy()); yq.append(tup[3].numpy())
            for lst in (tup[6], tup[7]):
                for h_c in lst:
                    a = np.asarray(h_c, np.int64)
                    nodes.append(a)
                    sampled.append(int(len(a) > 1 and bool(np.all(np.diff(a) > 0)) and len(a) in (args.sample_nodes, args.sample_nodes + 1, args.sample_nodes + 2)))
    out['nodes_flat'] = np.concatenate(nodes).astype(np.int32)
    out['nodes_off'] = np.cumsum([0] + [len(a) for a in nodes]).astype(np.int64)
    out['y_spt'] = np.stack(ys).astype(np.int32); out['y_qry'] = np.stack(yq).astype(np.int32)
    out['passes'] = passes
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    sizes = np.diff(out['nodes_off'])
    print('%-20s T=%d subgraphs=%d sizes min/median/max %d/%d/%d  at-or-above sample_nodes: %d  %.1f KB' % (
        name, T, len(nodes), sizes.min(), np.median(sizes), sizes.max(), int((sizes >= args.sample_nodes).sum()),
        os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024))


def node_replay(name, n, m, n_cls, h, sample_nodes, T, seed):
    rng = np.random.default_rng(seed)
    e = mg.pa_edges(n, m, rng)
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    lab = rng.integers(0, n_cls, size=n)
    names = ['0_%d' % v for v in range(n)]
    info = {nm: int(l) for nm, l in zip(names, lab)}
    args = mg.ns(h=h, sample_nodes=sample_nodes, n_way=3, k_spt=2, k_qry=4, task_num=T)
    dump(name, [(n, src, dst)], info, {'train.csv': (names, [str(l) for l in lab])}, args, T)


def link_replay(name, sample_nodes, T, seed):
    rng = np.random.default_rng(seed)
    graphs, info = [], {}
    csv_all, csv_spt, csv_qry = ([], []), ([], []), ([], [])
    for g in range(2):
        n = 120 + 20 * g
        e = mg.pa_edges(n, 3, rng)                    # positives, u<v only
        have = set(map(tuple, e)); neg = set()
        while len(neg) < len(e):
            a, b = rng.integers(0, n, size=2)
            if a != b and (min(a, b), max(a, b)) not in have:
                neg.add((int(a), int(b)))
        neg = np.array(sorted(neg), np.int64)
        graphs.append((n, np.concatenate([e[:, 0], neg[:, 0]]), np.concatenate([e[:, 1], neg[:, 1]])))
        for arr, lab in ((e, 1), (neg, 0)):
            perm = rng.permutation(len(arr)); half = len(arr) // 2
            for k, idx in enumerate(perm):
                nm = '%d_%d_%d' % (g, arr[idx, 0], arr[idx, 1])
                info[nm] = lab
                csv_all[0].append(nm); csv_all[1].append(str(lab))
                tgt = csv_spt if k < half else csv_qry
                tgt[0].append(nm); tgt[1].append(str(lab))
    args = mg.ns(task_setup='Shared', link_pred_mode='True', n_way=2, k_spt=4, k_qry=6, task_num=T, sample_nodes=sample_nodes)
    dump(name, graphs, info, {'train.csv': csv_all, 'train_spt.csv': csv_spt, 'train_qry.csv': csv_qry}, args, T)


def shared_short_class(name, T, seed):
    """Shared setup with one class smaller than k_shot + k_query in one graph: the reference's top-up branch (sdp.py:218-238).
    Only the task lists are recorded (ragged -> JSON): such a task cannot pass proto_loss_qry (meta.py:65) in the reference either."""
    rng = np.random.default_rng(seed)
    graphs, names, labels, info = [], [], [], {}
    for g, (n, short) in enumerate(((90, 0), (110, 5))):            # graph 1: class 2 has only 5 members (k_shot 3 + k_query 4 = 7)
        e = mg.pa_edges(n, 3, rng)
        graphs.append((n, np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])))
        lab = rng.integers(0, 2, size=n)
        if short:
            lab[rng.permutation(n)[:short]] = 2
        else:
            lab[rng.permutation(n)[:30]] = 2
        for v in range(n):
            nm = '%d_%d' % (g, v); names.append(nm); labels.append(str(int(lab[v]))); info[nm] = int(lab[v])
    args = mg.ns(task_setup='Shared', n_way=3, k_spt=3, k_qry=4, task_num=T, h=1)
    torch.manual_seed(222); np.random.seed(222); random.seed(222)
    root = tempfile.mkdtemp(prefix='gmeta_replay_') + '/'
    mg.write_csv(root + 'train.csv', names, labels)
    G = [mg.make_graph(*g) for g in graphs]
    db = sdp.Subgraphs(root, 'train', info, n_way=args.n_way, k_shot=args.k_spt, k_query=args.k_qry, batchsz=T, args=args, adjs=G, h=args.h)
    spt = [[[str(x) for x in sub] for sub in db.support_x_batch[t]] for t in range(T)]
    qry = [[[str(x) for x in sub] for sub in db.query_x_batch[t]] for t in range(T)]
    n_short = sum(1 for t in qry for sub in t if len(sub) != args.k_qry)
    assert n_short > 0, 'no task drew the short class'
    out = {'case': name, 'T': T, 'n_graphs': len(graphs), 'args': json.dumps(vars(args)), 'csv_files': json.dumps(['train.csv']),
           'csv_train.csv_names': np.array(names), 'csv_train.csv_labels': np.array(labels),
           'info_names': np.array(list(info)), 'info_labels': np.array([info[k] for k in info], np.int64),
           'spt_json': json.dumps(spt), 'qry_json': json.dumps(qry), 'rng_after': np.random.get_state()[1].copy()}
    for k, (n, s, d) in enumerate(graphs):
        out['g%d_n' % k] = n; out['g%d_src' % k] = np.asarray(s, np.int32); out['g%d_dst' % k] = np.asarray(d, np.int32)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('%-20s T=%d tasks, %d query lists topped up (k_query + 1 entries)  %.1f KB' % (name, T, n_short, os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024))


if __name__ == '__main__':
    node_replay('r0_replay_h2', 500, 4, 6, 2, 40, 5, 11)
    node_replay('r1_replay_h3', 300, 3, 5, 3, 60, 3, 12)
    node_replay('r2_replay_h1', 400, 6, 5, 1, 12, 4, 13)
    link_replay('r3_replay_link', 20, 4, 14)
    shared_short_class('r4_shared_short_class', 8, 15)
