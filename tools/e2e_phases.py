#!/usr/bin/env python3
"""Where the time of the receptive-field schedule's end-to-end loop goes (arxiv, task_num 32, cone + hoist_z1): per builder thread the host wall of each
phase of a meta-batch build WHILE the meta-steps run, and the consumer's wait / step time.  tools/e2e_phases.py [workers ...]"""
import os, sys, time, random, threading, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth, _lib
T = int(os.environ.get('T', '32'))
args, cfg = synth.make_args(os.environ.get('CONFIG', 'arxiv'), task_num=T)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
maml = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
FLAGS = [f for f in os.environ.get('FLAGS', 'cone,hoist_z1').split(',') if f]
for f in FLAGS:
    setattr(maml, f, 1)
NB = 8
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * NB, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
acc = collections.defaultdict(float); cnt = collections.defaultdict(int); lock = threading.Lock()
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t
            with lock:
                acc[name] += dt; cnt[name] += 1
    return w
db._prepare = timed('prepare', db._prepare)
_b0 = timed('build', db._build)
marks = {'end': None}
def _build_marked(prep, **kw):
    t = time.perf_counter()
    if marks['end'] is not None:
        with lock:
            acc['builder: gap between the end of one job and the start of the next build'] += t - marks['end']; cnt['builder: gap between the end of one job and the start of the next build'] += 1
    return _b0(prep, **kw)
db._build = _build_marked
lib = _lib.lib()
real_cone = lib.gm_batch_prepare_cone
class LibProxy:
    def __getattr__(self, k):
        return getattr(lib, k)
proxy = LibProxy(); proxy.gm_batch_prepare_cone = timed('cone (one batch)', real_cone)
proxy.gm_batch_prepare_cone_pair = timed('cone (both batches)', lib.gm_batch_prepare_cone_pair)
_lib_lib = _lib.lib
_lib.lib = lambda: proxy
real_sync = torch.cuda.Stream.synchronize
_s0 = timed('stream sync', real_sync)
def _sync_marked(self):
    r = _s0(self)
    marks['end'] = time.perf_counter()
    return r
torch.cuda.Stream.synchronize = _sync_marked
idx = [list(range(k * T, (k + 1) * T)) for k in range(NB)]
n_e = int(os.environ.get('STEPS', '100'))
for wk in [int(x) for x in sys.argv[1:]] or [1, 2]:
    it = iter(db.batches([idx[k % NB] for k in range(n_e + wk + 2)], prefetch=int(os.environ.get('PREFETCH', wk + 1)), cone_layers=cfg['h'] if 'cone' in FLAGS else 0, workers=wk))
    for _ in range(wk + 2):
        maml(*next(it), data['feats'])
    torch.cuda.synchronize(); acc.clear(); cnt.clear()
    wait = step = 0.0
    te = time.perf_counter()
    for _ in range(n_e):
        t = time.perf_counter(); b = next(it); wait += time.perf_counter() - t
        t = time.perf_counter(); maml(*b, data['feats']); step += time.perf_counter() - t
    torch.cuda.synchronize()
    ms = (time.perf_counter() - te) / n_e * 1e3
    print('workers %d: %.3f ms per step (%.0f tasks/s); consumer: wait for a batch %.3f ms, Meta.forward %.3f ms' % (wk, ms, T / ms * 1e3, wait / n_e * 1e3, step / n_e * 1e3))
    for k in sorted(acc):
        print('   %-18s %.3f ms per call x %.2f calls per step' % (k, acc[k] / max(cnt[k], 1) * 1e3, cnt[k] / n_e))
    del it
