#!/usr/bin/env python3
"""Is a stream-kernel aggregate launch bitwise the same when another stream keeps the chip busy?  Support-batch launch (arxiv shape, T tasks) on one stream,
repeated; on a second stream either nothing, stream-kernel launches over the query batch, or the persistent split GEMM.  Mismatching rows are classified
(hub row / plain row).     python tools/agg_stream_race.py [tasks] [reps]"""
import ctypes as C
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import _lib, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=T, args=args, adjs=store, h=cfg['h'],
                         tables=data['tables'], verbose=False)
batch = db.get_batch(list(range(T)))
S, Q = batch[0][0].view_of, batch[2][0].view_of
lib = _lib.lib()
print('support batch %d rows, query batch %d rows' % (S.rows, Q.rows))
W = (torch.randn(256, 256, device='cuda') * 0.05).contiguous()
xq = torch.randn(Q.rows, 256, device='cuda'); oq = torch.empty(Q.rows, 256, device='cuda'); og = torch.empty(Q.rows, 256, device='cuda')
xs = torch.randn(S.rows, 256, device='cuda')
pn_s = C.c_void_p(); lib.gm_batch_device_ptr(S.handle, _lib.F_NORM, C.byref(pn_s))
pn_q = C.c_void_p(); lib.gm_batch_device_ptr(Q.handle, _lib.F_NORM, C.byref(pn_q))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
deg = np.diff(np.asarray(S.csr()[0]))


def agg_s(out, width=256):
    _lib.check(lib.gm_aggregate(S.handle, 0, 0, _lib.ptr(xs), width, pn_s, None, _lib.ptr(out), C.c_void_p(sb.cuda_stream)), 'aggregate S')


def other(kind):
    if kind == 'stream-agg':
        _lib.check(lib.gm_aggregate(Q.handle, 0, 0, _lib.ptr(xq), 256, pn_q, None, _lib.ptr(oq), C.c_void_p(sa.cuda_stream)), 'aggregate Q')
    elif kind == 'gemm':
        _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(xq), 256, _lib.ptr(W), 0, 256, _lib.ptr(og), 1, C.c_void_p(sa.cuda_stream)), 'dense_update')


for mode, mname in ((1, 'stream kernel'), (0, 'window kernel')):
    _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', mode), 'set_tuning')
    ref = torch.empty(S.rows, 256, device='cuda'); agg_s(ref); torch.cuda.synchronize()
    for kind in (('gemm',) if os.environ.get('RACE_VERBOSE') else ('nothing', 'stream-agg', 'gemm')):
        bad_runs, hub_rows, plain_rows, worst = 0, 0, 0, 0.0
        for rep in range(REPS):
            out = torch.full((S.rows, 256), 7.0, device='cuda')
            torch.cuda.synchronize()
            for _ in range(3):
                other(kind)
            agg_s(out)
            other(kind)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad_runs += 1
                rows = torch.nonzero((out != ref).any(1)).flatten().cpu().numpy()
                hub_rows += int((deg[rows] > 32).sum()); plain_rows += int((deg[rows] <= 32).sum())
                worst = max(worst, float((out - ref).abs().max()))
                if os.environ.get('RACE_VERBOSE'):
                    ptr, idx = (np.asarray(a) for a in S.csr()[:2])
                    nrm = np.maximum(deg, 1).astype(np.float32) ** np.float32(-0.5)
                    xh = xs.cpu().numpy(); d_all = (out - ref).cpu().numpy()
                    part = int(os.environ.get('GM_AGG_HUB_PART', 128))
                    for r in rows[:4]:
                        d = d_all[r]; e0, e1 = ptr[r], ptr[r + 1]
                        P = max(1, (e1 - e0) // part)
                        msg = []
                        for k in range(P):
                            lo = e0 + k * part; hi = e1 if k == P - 1 else lo + part
                            pk = (xh[idx[lo:hi]] * nrm[idx[lo:hi], None]).sum(0)
                            for sign, nm in ((-1, 'missing'), (1, 'doubled')):
                                if np.abs(d - sign * pk).max() < 1e-3 * (np.abs(pk).max() + 1e-6):
                                    msg.append('part %d of %d %s' % (k, P, nm))
                        cols = np.nonzero(d)[0]
                        o_h = out[int(r)].cpu().numpy(); r_h = ref[int(r)].cpu().numpy()
                        what = []
                        if np.all(o_h[cols] == 0): what.append('output is ZERO there')
                        if np.all(o_h[cols] == 7.0): what.append('output still holds the fill value')
                        for R in (12, 8, 24):           # a stale ring slot: the gather of edge e read what edge e - R left there
                            for e in range(e0 + R, e1):
                                cand = nrm[idx[e]] * (xh[idx[e - R], cols] - xh[idx[e], cols])
                                if np.abs(d[cols] - cand).max() < 1e-3:
                                    what.append('edge %d of the row consumed the slot content of edge %d (ring depth %d)' % (e - e0, e - R - e0, R))
                            for e in range(e0, e1 - R):      # the slot overwritten too early: the read of edge e saw pieces of edge e + R's row
                                cand = nrm[idx[e]] * (xh[idx[e + R], cols] - xh[idx[e], cols])
                                if np.abs(d[cols] - cand).max() < 1e-3:
                                    what.append('edge %d of the row read pieces of the row gathered for edge %d (ring depth %d)' % (e - e0, e + R - e0, R))
                        for e in range(e0, e1):
                            if np.abs(d[cols] + nrm[idx[e]] * xh[idx[e], cols]).max() < 1e-3:
                                what.append('edge %d of the row contributed zero there' % (e - e0))
                        print('      row %d deg %d: columns %s differ, max %.3g; %s; %s' % (r, deg[r], sorted(set(int(c) // 16 for c in cols)), np.abs(d).max(),
                                                                                       ', '.join(msg) or 'no whole part explains it', '; '.join(what) or 'unexplained'), flush=True)
        print('%s on the support batch, other stream: %-10s  %d of %d launches differ from the solo result; rows: %d hub, %d plain; largest difference %.3g'
              % (mname, kind, bad_runs, REPS, hub_rows, plain_rows, worst), flush=True)
    if mode == 1 and os.environ.get('RACE_VERBOSE'):
        break
_lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 1), 'set_tuning')
