"""Worker for tests/test_hip_rccl.py: one rank of a world_size-N RCCL job, one GPU per rank.  The REAL product path:
HIP extraction -> gm_meta_step on the rank's task shard -> ONE all_reduce over RCCL -> gm_meta_finish -> fused Adam with the
device-side NaN guard; then the sharded evaluation (finetunning_batch(shard=True): all_gather over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from golden_util import Fixture      # noqa: E402
import gmeta_amd                     # noqa: E402,F401
from hip_util import fixture_batches, fixture_meta, make_store      # noqa: E402


def main():
    rank, world, port, case, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
    if os.environ.get('GMETA_TEST_ONE_GPU') == '1':
        # every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one device): the sharded HIP path -- shard, [grad | stats]
        # buffer, global-T mean, device-side NaN guard, sharded evaluation -- on real hardware where only one GPU is available
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    fx = Fixture(case)
    store = make_store(fx)
    S, Q = fixture_batches(fx, store, replay=True)
    m = fixture_meta(fx)
    m.force_allreduce = True          # world_size 1 still goes through the RCCL all-reduce
    bounds = np.linspace(0, fx.T, world + 1).round().astype(int)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    xs, xq = S.views(), Q.views()
    # a strict subset of a multi-set batch cannot be concatenated: rebuild this rank's shard as its own batches
    from gmeta_amd.subgraphs import SubgraphBatch

    def shard(tag):
        seeds = fx.z[tag + '_seeds'][lo:hi]
        n, s = seeds.shape[:2]
        lists = [fx.ref_nodes(tag, t, k) for t in range(lo, hi) for k in range(s)]
        return SubgraphBatch.from_nodes(store, seeds.reshape(-1, 3), np.arange(n + 1) * s, lists, fx.link).views()
    xs, xq = shard('spt'), shard('qry')
    ys = [torch.from_numpy(fx.z['y_spt'][t].astype(np.int64)) for t in range(lo, hi)]
    yq = [torch.from_numpy(fx.z['y_qry'][t].astype(np.int64)) for t in range(lo, hi)]
    accs = m(xs, ys, xq, yq, None, None, None, None, None, None, fx.feats)
    theta1 = [p.detach().cpu().numpy() for p in m.net.parameters()]
    pad = lambda part: [None] * lo + list(part) + [None] * (fx.T - hi)      # noqa: E731
    ft = m.finetunning_batch(pad(xs), pad(ys), pad(xq), pad(yq), shard=True)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), accs=accs, loss_q=m.last_stats['loss_q'], task_num=m.last_stats['task_num'], ft=ft,
             **{'v%d' % k: v for k, v in enumerate(theta1)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
