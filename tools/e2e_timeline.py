#!/usr/bin/env python3
"""Event timeline of the receptive-field schedule's end-to-end loop (who waits for whom): tools/e2e_timeline.py [steps]"""
import os, sys, time, random, threading, collections, concurrent.futures
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth, _lib
T = 32
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
maml = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
maml.cone = 1; maml.hoist_z1 = 1
NB = 8
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * NB, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
ev = []
def mark(name):
    ev.append((time.perf_counter(), threading.current_thread().name[:14], name))
def span(name, fn):
    def w(*a, **k):
        mark(name + ' {')
        try:
            return fn(*a, **k)
        finally:
            mark(name + ' }')
    return w
db._prepare = span('prepare', db._prepare)
db._build = span('build', db._build)
lib = _lib.lib()
class LibProxy:
    def __getattr__(self, k):
        return getattr(lib, k)
proxy = LibProxy(); proxy.gm_batch_prepare_cone_pair = span('cone', lib.gm_batch_prepare_cone_pair)
_lib.lib = lambda: proxy
torch.cuda.Stream.synchronize = span('sync', torch.cuda.Stream.synchronize)
concurrent.futures.ThreadPoolExecutor.submit = span('submit', concurrent.futures.ThreadPoolExecutor.submit)
concurrent.futures.Future.result = span('result', concurrent.futures.Future.result)
idx = [list(range(k * T, (k + 1) * T)) for k in range(NB)]
n_e = int(sys.argv[1]) if len(sys.argv) > 1 else 30
it = iter(db.batches([idx[k % NB] for k in range(n_e + 3)], prefetch=2, cone_layers=cfg['h'], workers=1))
for _ in range(3):
    maml(*next(it), data['feats'])
torch.cuda.synchronize(); ev.clear()
for _ in range(n_e):
    mark('next {'); b = next(it); mark('next }')
    mark('forward {'); maml(*b, data['feats']); mark('forward }')
torch.cuda.synchronize()
t0 = ev[0][0]
cut = [e for e in ev if e[0] - t0 > (ev[-1][0] - t0) * 0.6][:70]
for t, th, name in cut:
    col = 0 if th.startswith('Main') else 1 if 'prepare' in th else 2
    print('%9.3f ms  %s%-14s %s' % ((t - t0) * 1e3, ' ' * (28 * col), th, name))
