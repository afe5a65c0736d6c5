#!/usr/bin/env python3
"""Host-side profile (cProfile) of fresh-extraction meta-steps at the arxiv shape.  SOAK_FLAGS=cone,hoist_z1 selects schedules."""
import cProfile, pstats, os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth
T, N = 32, 24
args, cfg = synth.make_args('arxiv')
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
maml = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
for f in [x for x in os.environ.get('SOAK_FLAGS', '').split(',') if x]:
    setattr(maml, f, 1)
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * N, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
for s in range(4):
    maml(*db.get_batch(list(range(s * T, (s + 1) * T))), None)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
PF = int(os.environ.get('SOAK_PREFETCH', '0'))
for b in db.batches([list(range(s * T, (s + 1) * T)) for s in range(4, N)], prefetch=PF, cone_layers=cfg['h'] if getattr(maml, 'cone', 0) else 0):
    maml(*b, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print('%.2f ms per step (extraction + meta-step), %d steps, prefetch=%d' % (dt / (N - 4) * 1e3, N - 4, PF))
pstats.Stats(pr).sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 28)
