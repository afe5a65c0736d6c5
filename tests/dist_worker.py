"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU.  Exercises the N>1 path of
gmeta_amd.Meta.forward (task shard -> local meta-step buffer -> ONE all-reduce -> global-T division -> NaN guard ->
Adam) with the local HIP step replaced by the oracle as a test double (there is no GPU here; the product path itself
never falls back to it)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import gmeta_oracle as orc          # noqa: E402
from golden_util import Fixture      # noqa: E402
import gmeta_amd                     # noqa: E402


def main():
    rank, world, port, case, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    mode = sys.argv[6] if len(sys.argv) > 6 else 'train'      # 'train' | 'short' (one task only: rank 1 gets an EMPTY shard) | 'eval' (sharded finetunning)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fx = Fixture(case)
    graphs = fx.graphs()
    args = argparse.Namespace(**fx.args)
    m = gmeta_amd.Meta(args, fx.config)
    with torch.no_grad():
        for p, v in zip(m.net.parameters(), fx.vars0):
            p.copy_(torch.from_numpy(v))
    # contiguous task shard of this rank (uneven on purpose when T % world != 0)
    n_tasks = 1 if mode == 'short' else fx.T
    bounds = np.linspace(0, n_tasks, world + 1).round().astype(int)
    mine = list(range(bounds[rank], bounds[rank + 1]))

    def oracle_run(self, x_spt, y_spt, x_qry, y_qry, K, need_grad):
        theta = [p.detach().numpy().copy() for p in self.net.parameters()]
        P = sum(t.size for t in theta); K1 = K + 1; T = len(x_spt)
        out = np.zeros(P + 2 * K1 + 1 + T * K1 + 1, np.float32)         # T == 0 (empty shard): all zeros, like Meta._run; last float = violation word (0)
        for t in range(T):
            with np.errstate(all='ignore'):
                lq, aq, mg = orc.task_inner_loop(x_spt[t], x_qry[t], x_spt[t].features(fx.feats), x_qry[t].features(fx.feats), np.asarray(y_spt[t]),
                                                 np.asarray(y_qry[t]), theta, fx.config, self.k_spt, self.update_lr, K, need_grad)
            if mg is not None:
                out[:P] += np.concatenate([g.reshape(-1) for g in mg])
            out[P:P + K1] += lq; out[P + K1:P + 2 * K1] += aq
            out[P + 2 * K1 + 1 + t * K1:P + 2 * K1 + 1 + (t + 1) * K1] = aq
        out[P + 2 * K1] = T
        return torch.from_numpy(out), P, T
    gmeta_amd.Meta._run = oracle_run
    spt = [orc.extract_batch(graphs, fx.z['spt_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('spt', t)) for t in mine]
    qry = [orc.extract_batch(graphs, fx.z['qry_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('qry', t)) for t in mine]
    if mode == 'eval':
        # train.py:evaluate under torch.distributed: every rank holds only ITS slice (None elsewhere), results all-gathered
        pad = lambda part: [None] * int(bounds[rank]) + list(part) + [None] * (fx.T - int(bounds[rank + 1]))      # noqa: E731
        accs = m.finetunning_batch(pad(spt), pad([fx.z['y_spt'][t] for t in mine]), pad(qry), pad([fx.z['y_qry'][t] for t in mine]), shard=True)
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), accs=accs)
        dist.barrier(); dist.destroy_process_group()
        return
    accs = m(spt, [fx.z['y_spt'][t] for t in mine], qry, [fx.z['y_qry'][t] for t in mine], None, None, None, None, None, None, fx.feats)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), accs=accs, loss_q=m.last_stats['loss_q'], task_num=m.last_stats['task_num'],
             **{'v%d' % k: p.detach().numpy() for k, p in enumerate(m.net.parameters())})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
