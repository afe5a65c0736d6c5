"""GPU-side helpers shared by the -m gpu tests and __graft_entry__.smoke(): drive the HIP path
(through the Python host mirror, i.e. through the C ABI) on a golden fixture."""
import argparse

import numpy as np
import torch

import gmeta_amd
from gmeta_amd import _lib
from gmeta_amd.subgraphs import SubgraphBatch


def make_store(fx):
    return gmeta_amd.GraphStore(fx.edges, fx.feats)


def fixture_batches(fx, store, replay):
    """Support / query mega-batches (one set per task) for a fixture."""
    out = {}
    for tag in ('spt', 'qry'):
        seeds = fx.z[tag + '_seeds']
        T, S = seeds.shape[:2]
        flat = seeds.reshape(-1, 3)
        off = np.arange(T + 1) * S
        if replay:
            lists = [fx.ref_nodes(tag, t, s) for t in range(T) for s in range(S)]
            out[tag] = SubgraphBatch.from_nodes(store, flat, off, lists, fx.link)
        else:
            out[tag] = SubgraphBatch.extract(store, flat, off, fx.args['h'], fx.args['sample_nodes'], 222, fx.link)
    return out['spt'], out['qry']


def fixture_meta(fx):
    args = argparse.Namespace(**fx.args)
    m = gmeta_amd.Meta(args, fx.config).to('cuda')
    with torch.no_grad():
        for p, v in zip(m.net.parameters(), fx.vars0):
            p.copy_(torch.from_numpy(v))
    return m


def hip_meta_step(fx, replay=True, hoist=0, sparse_bwd=0, cone=0, fused_adam_kernel=True):
    """Meta.forward on the fixture's meta-batch.  Returns accs, the meta-gradient that reached Adam,
    the updated weights and the node lists."""
    store = make_store(fx)
    S, Q = fixture_batches(fx, store, replay)
    m = fixture_meta(fx)
    m.hoist_z1 = hoist
    m.sparse_bwd = sparse_bwd
    m.cone = cone
    m.fused_adam_kernel = bool(fused_adam_kernel)
    ys = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_spt']]
    yq = [torch.from_numpy(y.astype(np.int64)) for y in fx.z['y_qry']]
    xs, xq = S.views(), Q.views()
    accs = m(xs, ys, xq, yq, None, None, None, None, None, None, fx.feats)
    # the meta-gradient that reached Adam = p.grad after the step (views of the flat buffer gm_meta_finish[_adam] wrote: the mean over the tasks)
    grad = torch.cat([p.grad.reshape(-1) for p in m.net.parameters()]).cpu().numpy().copy()
    res = {'accs': accs, 'grad': grad, 'vars1': [p.detach().cpu().numpy() for p in m.net.parameters()],
           'spt_parent': S.parent().astype(np.int64), 'qry_parent': Q.parent().astype(np.int64), 'stats': m.last_stats,
           'S': S, 'Q': Q, 'meta': m, 'store': store}
    return res
