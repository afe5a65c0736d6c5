#!/usr/bin/env python3
"""A/B of the aggregate kernels on real batches of the arxiv shape: window kernel (k_agg_win, hub rows in its schedule) vs the LDS-DMA stream kernel
(agg_stream.hip).  For the support and the query batch of a meta-batch, widths 256 / 128 and the layer-1 gather: output compared bitwise, time per
launch, algorithmic GB/s (SURVEY 8(d) B_agg).
    GM_AGG_STREAM=1 [GM_AGG_STREAM_WGS=k] [GM_AGG_STREAM_DEPTH=8|12|16] python tools/agg_stream_bench.py [tasks]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault('GM_AGG_STREAM', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import _lib, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); import random; random.seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=T, args=args, adjs=store, h=cfg['h'],
                         tables=data['tables'], verbose=False)
batch = db.get_batch(list(range(T)))
torch.cuda.synchronize(); print('batch built', flush=True)
lib = _lib.lib()


def run(B, width, transposed, gather, stream, x, out, norm_p, n=20):
    _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', stream), 'set_tuning')
    call = lambda: _lib.check(lib.gm_aggregate(B.handle, transposed, gather, None if gather else _lib.ptr(x), width, norm_p, None, _lib.ptr(out), _lib.stream_ptr()))
    call(); torch.cuda.synchronize(); call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for tag, B in (('support', batch[0][0].view_of), ('query', batch[2][0].view_of)):
    deg = np.diff(B.csr()[0])
    print('%s: rows %d edges %d  deg mean %.2f p99 %d max %d  hub rows (>32) %d' % (tag, B.rows, B.edges, deg.mean(), np.percentile(deg, 99), deg.max(), int((deg > 32).sum())), flush=True)
    p = C.c_void_p(); lib.gm_batch_device_ptr(B.handle, _lib.F_NORM, C.byref(p))
    for width, gather in ((256, 0), (128, 0), (128, 1)):
        x = torch.randn(B.rows, width, device='cuda')
        for transposed in ((0, 1) if not gather else (0,)):
            o_old = torch.full((B.rows, width), 7.0, device='cuda'); o_new = torch.full((B.rows, width), 9.0, device='cuda')
            t_old = run(B, width, transposed, gather, 0, x, o_old, p)
            t_new = run(B, width, transposed, gather, 1, x, o_new, p)
            by = lib.gm_aggregate_bytes(B.handle, width)
            same = bool(torch.equal(o_old, o_new))
            bad = int((o_old != o_new).any(dim=1).sum()) if not same else 0
            print('  width %3d gather %d transposed %d: window %.1f us %.0f GB/s (%.3f) | stream %.1f us %.0f GB/s (%.3f) | bitwise %s%s'
                  % (width, gather, transposed, t_old * 1e6, by / t_old / 1e9, by / t_old / 8e12, t_new * 1e6, by / t_new / 1e9, by / t_new / 8e12, same,
                     '' if same else ' (%d rows differ, max %.3g)' % (bad, float((o_old - o_new).abs().max()))), flush=True)
