#!/usr/bin/env python3
"""Do an aggregate launch and the persistent split GEMM of the other stream CO-RESIDE on the CUs?  The GEMM (k_gemm_split_p: 16 waves x 120 VGPRs,
101 KiB of LDS per CU) runs over the arxiv-shape query batch on one stream; while it runs, aggregate launches over the support batch are queued on a
second stream.  Reported: each kernel alone, then together -- the GEMM's duration with the aggregates beside it, how many aggregate launches completed
inside the GEMM's span, and the wall time of the pair against the sum of the parts.
    GM_AGG_STREAM=1 [GM_AGG_STREAM_WGS=k] python tools/coreside_probe.py [tasks]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault('GM_AGG_STREAM', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401  (probe build of the library: gm_debug_stamp / gm_stream_debug / gm_head_loss_debug)
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
S, Q = batch[0][0].view_of, batch[2][0].view_of
lib = _lib.lib()
torch.cuda.synchronize()

W = (torch.randn(256, 256, device='cuda') * 0.05).contiguous()
xq = torch.randn(Q.rows, 256, device='cuda'); oq = torch.empty(Q.rows, 256, device='cuda')
xs = torch.randn(S.rows, 256, device='cuda'); os_ = torch.empty(S.rows, 256, device='cuda')
pn = C.c_void_p(); lib.gm_batch_device_ptr(S.handle, _lib.F_NORM, C.byref(pn))
s_gemm = torch.cuda.Stream()
s_agg = torch.cuda.Stream(priority=-1)          # (higher priority: the support chain's stream in gm_meta_step)
N_AGG = 1 if os.environ.get('PROBE_ONE') else 12
lib.gm_stream_debug(1, None, 8192)


def gemm():
    _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(xq), 256, _lib.ptr(W), 0, 256, _lib.ptr(oq), 1, C.c_void_p(s_gemm.cuda_stream)), 'dense_update')


def agg():
    _lib.check(lib.gm_aggregate(S.handle, 0, 0, _lib.ptr(xs), 256, pn, None, _lib.ptr(os_), C.c_void_p(s_agg.cuda_stream)), 'aggregate')


stamps = torch.zeros(64, dtype=torch.int64, device='cuda')


def stamp(k, st):
    _lib.check(lib.gm_debug_stamp(C.c_void_p(stamps.data_ptr() + 8 * k), C.c_void_p(st.cuda_stream)), 'stamp')


for mode, name in ((0, 'window kernel (k_agg_win)'), (1, 'stream kernel (k_agg_stream)')):
    _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', mode), 'set_tuning')
    for _ in range(3):
        gemm(); agg()
    torch.cuda.synchronize()
    # alone
    stamp(0, s_gemm); gemm(); stamp(1, s_gemm); torch.cuda.synchronize()
    stamp(2, s_agg)
    for _ in range(N_AGG):
        agg()
    stamp(3, s_agg); torch.cuda.synchronize()
    h = stamps.cpu().numpy()
    t_gemm, t_agg = (h[1] - h[0]) / 100.0, (h[3] - h[2]) / 100.0 / N_AGG
    # together: the GEMM first, then the aggregates on the other stream while it runs; device-clock stamps between the launches of both streams
    w0 = torch.cuda.Event(); w0.record(s_gemm); s_agg.wait_event(w0)
    stamp(4, s_gemm); gemm(); stamp(5, s_gemm)
    stamp(6, s_agg)
    for k in range(N_AGG):
        agg(); stamp(7 + k, s_agg)
    torch.cuda.synchronize()
    h = stamps.cpu().numpy().astype(np.int64)
    if mode == 1:
        wg = np.zeros(8192, np.uint64)
        lib.gm_stream_debug(1, wg.ctypes.data_as(C.c_void_p), 8192)
        wg = wg.astype(np.int64).reshape(-1, 2); wg = wg[wg[:, 0] > 0]
        print('    last stream launch, %d workgroups: start %.0f .. %.0f us, end %.0f .. %.0f us after the GEMM stream\'s start stamp; median duration %.1f us'
              % (len(wg), (wg[:, 0].min() - h[4]) / 100.0, (wg[:, 0].max() - h[4]) / 100.0, (wg[:, 1].min() - h[4]) / 100.0, (wg[:, 1].max() - h[4]) / 100.0,
                 float(np.median(wg[:, 1] - wg[:, 0])) / 100.0))
    g0, g1, a0 = h[4], h[5], h[6]
    ends = [(h[7 + k] - g0) / 100.0 for k in range(N_AGG)]
    inside = sum(1 for k in range(N_AGG) if h[7 + k] <= g1)
    print('%s: GEMM alone %.0f us, aggregate alone %.1f us per launch' % (name, t_gemm, t_agg))
    print('    together (device clock, us after the GEMM stream\'s start stamp): GEMM ends at %.0f; aggregate stream starts at %.0f, its launches end at %s; %d of %d inside the GEMM span; pair %.0f us vs sum of parts %.0f us'
          % ((g1 - g0) / 100.0, (a0 - g0) / 100.0, ' '.join('%.0f' % e for e in ends), inside, N_AGG, max((g1 - g0) / 100.0, ends[-1]), t_gemm + t_agg * N_AGG), flush=True)
