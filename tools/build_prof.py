#!/usr/bin/env python3
"""Kernel time of ONE meta-batch build (extraction + finalisation [+ receptive-field tables]) at the arxiv shape, for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats -d out -o x -- python tools/build_prof.py [cone_layers]"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import _lib, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = 32
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); random.seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * 6, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)


def build(k):
    b = db.get_batch(list(range(k * T, (k + 1) * T)))
    if L:
        _lib.check(_lib.lib().gm_batch_prepare_cone_pair(b[0][0].view_of.handle, b[2][0].view_of.handle, L, _lib.stream_ptr()), 'cone')
    return b


build(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(1, 5):
    build(k)
torch.cuda.synchronize()
print('build%s: %.2f ms per meta-batch (host wall, GPU otherwise idle)' % (' + cone tables' if L else '', (time.perf_counter() - t0) / 4 * 1e3))

if os.environ.get('PHASES'):
    # host-wall phases of one build on an idle GPU (no profiler): where the builder thread's time goes
    acc = {}
    def tick(name, t):
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t)
    n = 20
    for k in range(n):
        idx = list(range((k % 5) * T, (k % 5 + 1) * T))
        t = time.perf_counter(); prep = db._prepare(idx); tick('prepare (python: task draw, seeds, labels)', t)
        t = time.perf_counter(); b = db._build(prep); tick('build (gm_extract_pair: launch + finalise)', t)
        if L and os.environ.get('PHASES') == '2':
            for nm, x in (('cone support', b[0][0]), ('cone query', b[2][0])):
                t = time.perf_counter()
                _lib.check(_lib.lib().gm_batch_prepare_cone(x.view_of.handle, L, _lib.stream_ptr()), 'cone'); tick(nm, t)
        elif L:
            t = time.perf_counter()
            _lib.check(_lib.lib().gm_batch_prepare_cone_pair(b[0][0].view_of.handle, b[2][0].view_of.handle, L, _lib.stream_ptr()), 'cone'); tick('cone tables, both batches (gm_batch_prepare_cone_pair)', t)
        t = time.perf_counter(); torch.cuda.synchronize(); tick('final sync', t)
        roots = [b[0][0].view_of, b[2][0].view_of]
        t = time.perf_counter(); del b; tick('drop the 10-tuple (views, lists, tensors)', t)
        t = time.perf_counter(); del roots; tick('gm_batch_destroy x 2', t)
    for k, v in acc.items():
        print('  %-48s %.3f ms' % (k, v / n * 1e3))
    print('  total %.3f ms' % (sum(acc.values()) / n * 1e3))
