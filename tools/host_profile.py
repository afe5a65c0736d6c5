#!/usr/bin/env python3
"""Where the HOST spends a meta-step (cProfile over N steps of Meta.forward on pre-extracted batches): the FirstMM shape issues ~180 launches per 1.4-ms step,
i.e. the step is bound by the enqueueing thread.     python tools/host_profile.py [config] [steps]"""
import cProfile
import os
import pstats
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else 'firstmm'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args, cfg = synth.make_args(name)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg), link=bool(cfg.get('link')))
m = gmeta_amd.Meta(args, config).to('cuda')
T = cfg['task_num']
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=2 * T, args=args, adjs=store, h=cfg['h'],
                         tables=data['tables'], verbose=False)
bs = [db.get_batch(list(range(T))), db.get_batch(list(range(T, 2 * T)))]
for i in range(20):
    m(*bs[i & 1], data['feats'])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    m(*bs[i & 1], data['feats'])
torch.cuda.synchronize()
print('%s: %.3f ms per step unprofiled' % (name, (time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    m(*bs[i & 1], data['feats'])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(18)
