#!/usr/bin/env python3
"""Soak test at the arxiv shape: N meta-steps, each with a FRESH extraction (get_batch) + Meta.forward, then a batched
finetunning; checks finite outputs and that device memory does not grow."""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if os.environ.get('SOAK_SWITCH'):
    sys.setswitchinterval(float(os.environ['SOAK_SWITCH']))      # GIL hand-over period (default 5 ms)
T = 32
args, cfg = synth.make_args('arxiv')
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
maml = gmeta_amd.Meta(args, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')
for f in [x for x in os.environ.get('SOAK_FLAGS', '').split(',') if x]:      # e.g. SOAK_FLAGS=cone,hoist_z1
    setattr(maml, f, 1)
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * N, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
free0 = None
t_ext = t_step = 0.0
t0 = time.perf_counter()
PF = int(os.environ.get('SOAK_PREFETCH', '0'))      # >0: extraction of the next meta-batch overlaps the current meta-step (Subgraphs.batches)
it = iter(db.batches([list(range(s * T, (s + 1) * T)) for s in range(N)], prefetch=PF, cone_layers=cfg['h'] if getattr(maml, 'cone', 0) else 0))
for s in range(N):
    ta = time.perf_counter()
    b = next(it)
    tb = time.perf_counter()
    accs = maml(*b, None)
    tc = time.perf_counter(); t_ext += tb - ta; t_step += tc - tb
    assert np.isfinite(accs).all(), accs
    if s == max(4, N // 2):      # the stream-ordered pool keeps freed blocks (no release to the OS): compare plateau to plateau
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; res0 = torch.cuda.memory_reserved()
    if s % 10 == 0:
        print('step %3d acc0 %.3f accK %.3f loss %.4f free %.1f GB' % (s, accs[0], accs[-1], maml.last_stats['loss_q'], torch.cuda.mem_get_info()[0] / 2**30), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]; res1 = torch.cuda.memory_reserved()
print('%d steps incl. extraction: %.1f ms/step = %.0f tasks/s (host time: get_batch %.1f ms, Meta.forward %.1f ms) ; free memory drift %.1f MB'
      % (N, dt / N * 1e3, T * N / dt, t_ext / N * 1e3, t_step / N * 1e3, (free0 - free1) / 2**20))
ev = db.get_batch(list(range(8)))
fa = maml.finetunning_batch(ev[0], ev[1], ev[2], ev[3])
print('finetunning_batch over 8 tasks, K_test=%d: mean accs first/last %.3f %.3f' % (maml.update_step_test, fa[:, 0].mean(), fa[:, -1].mean()))
# torch's caching allocator re-grows the meta-step workspace when a larger batch arrives (reserved memory is its, not a leak):
# what must stay flat is everything else -- the library's stream-ordered pool and the batches
print('torch reserved grew by %.1f MB over the same span' % ((res1 - res0) / 2**20))
assert abs((free0 - free1) - (res1 - res0)) < 512 * 2**20, 'device memory grows outside torch\'s allocator'
