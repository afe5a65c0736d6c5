"""Probe for bench.py's cpu_baseline on the GPU box's host: one arxiv-shape task, numpy-omp vs torch-cpu at several
thread counts (a 256-thread torch pool on sub-millisecond ops is pathological; the bench picks its thread count from this)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gmeta_amd                      # noqa: E402
from gmeta_amd import synth          # noqa: E402
import gmeta_oracle as orc           # noqa: E402
import torch_cpu_baseline as tcb     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
np.random.seed(222); torch.manual_seed(222)
args, cfg = synth.make_args(name, task_num=2)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=2, args=args, adjs=store,
                         h=cfg['h'], tables=data['tables'], verbose=False)
batch = db.get_batch([0, 1])
config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg), link=bool(cfg.get('link')))
graphs = [orc.Graph(*g) for g in data['graphs']]
S, Q = batch[0][0].view_of, batch[2][0].view_of
out = []
for B, seeds in ((S, db._task_arrays(0)[0]), (Q, db._task_arrays(0)[1])):
    so, par, off = B.set_sub_off, B.parent(), B.sub_off
    out.append(orc.Batch(graphs, [tuple(int(v) for v in s) for s in seeds], [par[off[k]:off[k + 1]] for k in range(so[0], so[1])]))
bs, bq = out
xs, xq = bs.features(data['feats']), bq.features(data['feats'])
ys, yq = np.asarray(batch[1][0]), np.asarray(batch[3][0])
rng = np.random.default_rng(0)
theta = []
for nm, p in config:
    if nm == 'GraphConv':
        theta += [(rng.standard_normal(p) * 0.05).astype(np.float32), np.zeros(p[1], np.float32)]
    elif nm == 'Linear':
        theta += [(rng.standard_normal((p[1], p[0] * (2 if cfg.get('link') else 1))) * 0.1).astype(np.float32), np.zeros(p[1], np.float32)]
K = 3
print('cores', os.cpu_count(), 'torch default threads', torch.get_num_threads(), flush=True)
for rep in range(2):
    t0 = time.perf_counter(); orc.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, config, cfg['k_spt'], cfg['update_lr'], K, True)
    print('numpy-omp (before torch threads touched) K=3: %.2f s' % (time.perf_counter() - t0), flush=True)
for nt in (8, 32, 64, 128, os.cpu_count()):
    torch.set_num_threads(nt)
    t0 = time.perf_counter(); tcb.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, cfg['h'], cfg['k_spt'], cfg['update_lr'], K, True)
    t1 = time.perf_counter(); tcb.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, cfg['h'], cfg['k_spt'], cfg['update_lr'], K, True)
    print('torch-cpu threads=%d K=3: %.2f s (first %.2f)' % (nt, time.perf_counter() - t1, t1 - t0), flush=True)
    t0 = time.perf_counter(); orc.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, config, cfg['k_spt'], cfg['update_lr'], K, True)
    print('   numpy-omp afterwards: %.2f s' % (time.perf_counter() - t0), flush=True)
