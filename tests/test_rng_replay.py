"""with the reference's seeds (train.py:33-35 + random) the host mirror draws the SAME tasks as the
reference's Subgraphs (global numpy / python RNG parity of create_batch_*), and in sample_mode='reference' the SAME node
sets for neighbourhoods above sample_nodes (CPython set order + legacy np.random.choice + the per-name memo).  Fixtures:
tests/golden/r*_replay_*.npz, written by oracle/make_replay_golden.py from the reference's own module."""
import argparse
import json
import os
import random

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['r0_replay_h2', 'r1_replay_h3', 'r2_replay_h1', 'r3_replay_link']


class _HostStore:
    """Stands in for GraphStore where no GPU is available: only the host CSR that the replay walks."""
    def __init__(self, graphs):
        from gmeta_amd.graphstore import edges_to_in_csr
        self.host_csr = [edges_to_in_csr(n, s, d) for n, s, d in graphs]


def _load(case):
    z = np.load(os.path.join(GOLD, case + '.npz'), allow_pickle=False)
    args = argparse.Namespace(**json.loads(str(z['args'])))
    graphs = [(int(z['g%d_n' % k]), z['g%d_src' % k], z['g%d_dst' % k]) for k in range(int(z['n_graphs']))]
    tables = {fn[:-4]: ([str(x) for x in z['csv_%s_names' % fn]], [str(x) for x in z['csv_%s_labels' % fn]]) for fn in json.loads(str(z['csv_files']))}
    info = {str(k): int(v) for k, v in zip(z['info_names'], z['info_labels'])}
    return z, args, graphs, tables, info


def _db(z, args, tables, info, store, monkeypatch=None):
    import gmeta_amd.subgraphs as sg
    if monkeypatch is not None:
        monkeypatch.setattr(sg, 'GraphStore', _HostStore)
    torch.manual_seed(222); np.random.seed(222); random.seed(222)
    return sg.Subgraphs(None, 'train', info, n_way=args.n_way, k_shot=args.k_spt, k_query=args.k_qry, batchsz=int(z['T']), args=args,
                        adjs=store, h=args.h, tables=tables, verbose=False, sample_mode='reference')


@pytest.mark.parametrize('case', CASES)
def test_task_lists_and_sampled_sets_follow_the_reference_rng(case, monkeypatch):
    """CPU: same task names as the reference drew; walking the tasks in the reference's order, every neighbourhood the
    fixture shows as sampled comes out of the host replay bit-identical, and the relabelled targets match."""
    z, args, graphs, tables, info = _load(case)
    db = _db(z, args, tables, info, _HostStore(graphs), monkeypatch)
    T = int(z['T'])
    assert [db._task_names(t)[0] for t in range(T)] == [[str(x) for x in row] for row in z['spt_names']]
    assert [db._task_names(t)[1] for t in range(T)] == [[str(x) for x in row] for row in z['qry_names']]
    off, flat = z['nodes_off'], z['nodes_flat']
    k = 0
    n_sampled = 0
    for p in range(int(z['passes'])):
        for t in range(T):
            seeds_s, seeds_q, lab_s, lab_q = db._task_arrays(t)
            names_s, names_q = db._task_names(t)
            for name, (g, i, j) in zip(names_s + names_q, np.concatenate([seeds_s, seeds_q]).tolist()):
                ref = flat[off[k]:off[k + 1]]; k += 1
                # oversize <=> the reference sampled: sorted, sample_nodes .. sample_nodes + 2 nodes (sdp.py:312-314,337-339);
                # decide it the way the device path does, from the full neighbourhood size
                full = _full_size(db, g, i, j)
                if full > args.sample_nodes:
                    assert np.array_equal(db._reference_nodes(name, g, i, j), ref), (case, name)
                    n_sampled += 1
                else:
                    assert len(ref) == full
            ys, yq = db._labels(lab_s, lab_q)                    # Disjoint: random.shuffle of the classes (sdp.py:389-397)
            assert np.array_equal(ys.numpy(), z['y_spt'][p * T + t]) and np.array_equal(yq.numpy(), z['y_qry'][p * T + t])
    assert n_sampled > 0 and k == len(off) - 1


def _full_size(db, g, i, j):
    """|N_h(i)| (or the link-pred union) from the host CSR, as a set."""
    def inn(v):
        return db._in(g, v)
    if db.link_pred_mode:
        f = inn(i); a = set(x for u in f for x in inn(u)) | set(f) | {i}
        b = set(inn(j)) | {j}
        return len(a | b)
    cur, seen = {i}, {i}
    for _ in range(db.h):
        cur = set(x for u in cur for x in inn(u))
        seen |= cur
    return len(seen)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_reference_mode_batches_hold_the_reference_node_sets(case):
    """GPU: Subgraphs(sample_mode='reference').__getitem__ in task order returns, subgraph by subgraph, the reference's
    node sets -- exactly (sorted) where it sampled, as a set elsewhere (CPython set order is not kept on the device) --
    and a second pass over the tasks replays the memo without touching the RNG (sdp.py:296-297)."""
    import gmeta_amd
    z, args, graphs, tables, info = _load(case)
    feats = [np.zeros((n, 4), np.float32) for n, _, _ in graphs]
    store = gmeta_amd.GraphStore(graphs, feats)
    db = _db(z, args, tables, info, store)
    off, flat = z['nodes_off'], z['nodes_flat']
    T, k = int(z['T']), 0
    for p in range(int(z['passes'])):
        state = np.random.get_state()[1].copy() if p == 1 else None
        for t in range(T):
            tup = db[t]
            for lst in (tup[6], tup[7]):
                for ids in lst:
                    ref = flat[off[k]:off[k + 1]]; k += 1
                    got = np.asarray(ids)
                    if len(ref) > 1 and np.all(np.diff(ref) > 0):
                        assert np.array_equal(got, ref)
                    else:
                        assert np.array_equal(got, np.sort(ref))
            assert np.array_equal(tup[1].numpy(), z['y_spt'][p * T + t]) and np.array_equal(tup[3].numpy(), z['y_qry'][p * T + t])
        if p == 1:
            assert np.array_equal(np.random.get_state()[1], state)          # memoised: no further draws
    # the batched path visits tasks in the same order and reuses the memo
    b = db.get_batch(list(range(T)))
    assert len(b[0]) == T and sum(len(x) for x in b[6]) + sum(len(x) for x in b[7]) == (len(off) - 1) // int(z['passes'])


def test_shared_short_class_top_up_follows_the_reference(monkeypatch):
    """sdp.py:218-238: a Shared-setup class smaller than k_shot + k_query is topped up from random classes of the same graph (the
    reference's bare `except:` branch).  With the reference's seeds the host mirror draws the same (ragged) task lists, consumes the
    global RNG identically, and the topped-up task carries k_query + 1 query entries exactly like the reference's."""
    import json as _json
    z, args, graphs, tables, info = _load('r4_shared_short_class')
    db = _db(z, args, tables, info, _HostStore(graphs), monkeypatch)
    spt, qry = _json.loads(str(z['spt_json'])), _json.loads(str(z['qry_json']))
    assert [[list(map(str, sub)) for sub in t] for t in db.support_x_batch] == spt
    assert [[list(map(str, sub)) for sub in t] for t in db.query_x_batch] == qry
    assert any(len(sub) == args.k_qry + 1 for t in qry for sub in t)
    assert np.array_equal(np.random.get_state()[1], z['rng_after'])
