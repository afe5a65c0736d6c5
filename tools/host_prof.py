#!/usr/bin/env python3
"""cProfile of Meta.forward over pre-extracted meta-batches (the bench's timed loop): where the host time of a step goes."""
import cProfile, pstats, os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth
T = int(os.environ.get('T', '4'))
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
maml = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
for f in [x for x in os.environ.get('SOAK_FLAGS', '').split(',') if x]:
    setattr(maml, f, 1)
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * 2, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
bs = [db.get_batch(list(range(s * T, (s + 1) * T))) for s in range(2)]
for k in range(4):
    maml(*bs[k % 2], None)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for k in range(N):
    maml(*bs[k % 2], None)
torch.cuda.synchronize()
print('T=%d: %.3f ms per step without the profiler' % (T, (time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for k in range(N):
    maml(*bs[k % 2], None)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
