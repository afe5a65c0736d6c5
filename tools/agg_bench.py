#!/usr/bin/env python3
"""Micro-benchmark of the aggregate kernel on the arxiv-shape query batch (one rank's share of a meta-batch).
    GM_AGG_VARIANT=k python tools/agg_bench.py [width] [tasks]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import _lib, synth

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); import random; random.seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
batch = db.get_batch(list(range(T)))
Q = batch[2][0].view_of
deg = np.diff(Q.csr()[0])
print('rows %d edges %d  deg mean %.2f p50 %d p90 %d p99 %d max %d zero %.1f%%  subs %d' % (Q.rows, Q.edges, deg.mean(), np.percentile(deg, 50),
      np.percentile(deg, 90), np.percentile(deg, 99), deg.max(), 100.0 * (deg == 0).mean(), Q.subs))
lib = _lib.lib()
x = torch.randn(Q.rows, width, device='cuda'); out = torch.empty_like(x)
norm = torch.empty(Q.rows, device='cuda'); 
p = C.c_void_p(); lib.gm_batch_device_ptr(Q.handle, _lib.F_NORM, C.byref(p))
bytes_ = lib.gm_aggregate_bytes(Q.handle, width)
gather = int(os.environ.get('AGG_GATHER', '0'))       # 1: layer-1 mode, rows gathered from the store's feature table (width must be F0)
for transposed in (0, 1):
    if gather and transposed:
        continue
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            _lib.check(lib.gm_aggregate(Q.handle, transposed, gather, None if gather else _lib.ptr(x), width, p, None, _lib.ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('variant %s width %d transposed %d: %.3f ms  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)' % (os.environ.get('GM_AGG_VARIANT', '0'), width, transposed,
          dt * 1e3, bytes_ / dt / 1e9, 100 * bytes_ / dt / 8e12))
# calibration: plain copy of the same bytes
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    out.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print('torch copy_: %.3f ms  %.0f GB/s (read+write)' % (dt * 1e3, 2 * x.numel() * 4 / dt / 1e9))
