#!/usr/bin/env python3
"""Run-to-run reproducibility of a meta-step (same batch, same initial weights): N steps from identical state, bitwise comparison of the accuracies,
query losses and meta-gradient with the first run; under a list of GM_* settings.
    python tools/repro_stress.py [config] [task_num] [runs] [KNOB=v,KNOB=v ...]"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import _lib, synth

name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
settings = sys.argv[4:] or ['']
args, cfg = synth.make_args(name, task_num=T)
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg), link=bool(cfg.get('link')))
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=T, args=args, adjs=store, h=cfg['h'],
                         tables=data['tables'], verbose=False)
b = db.get_batch(list(range(T)))
lib = _lib.lib()


def step(serialize, upd=0):
    torch.manual_seed(7)
    if upd:
        args.update_step = upd
    m = gmeta_amd.Meta(args, config).to('cuda')
    m.serialize = serialize
    accs = m(*b, None)
    global last_params
    last_params = list(m.net.parameters())
    g = torch.cat([p.grad.reshape(-1) for p in m.net.parameters()])
    return np.asarray(accs), np.asarray(m.last_stats['losses_q']), g.clone()


for st in settings:
    serialize, alt, upd = 0, 0, 0
    for kv in filter(None, st.split(',')):
        k, v = kv.split('=')
        if k == 'SERIALIZE':
            serialize = int(v)
        elif k == 'GEMM_MODE':
            lib.gm_set_gemm_mode(int(v))
        elif k == 'ALT':                    # a step in the other GEMM arithmetic before every run (tests/test_hip_fullsize.py's pattern)
            alt = int(v)
        elif k == 'UPDATE_STEP':
            upd = int(v)
        else:
            _lib.check(lib.gm_set_tuning(k.encode(), int(v)), 'set_tuning ' + k)
    a0, l0, g0 = step(serialize, upd)
    bad, worst = 0, 0.0
    for _ in range(N):
        if alt:
            prev = lib.gm_get_gemm_mode(); lib.gm_set_gemm_mode(1 - prev); step(serialize, upd); lib.gm_set_gemm_mode(prev)
        a, l, g = step(serialize, upd)
        if not (np.array_equal(a, a0) and np.array_equal(l, l0) and torch.equal(g, g0)):
            bad += 1
            if os.environ.get('STRESS_VERBOSE'):
                sizes = [p.numel() for p in last_params]
                off, parts = 0, []
                for i, nsz in enumerate(sizes):
                    d = float((g[off:off + nsz] - g0[off:off + nsz]).abs().max()); off += nsz
                    parts.append('%d:%.1e' % (i, d))
                print('   run differs: accs %s losses %s | max |dg| per parameter %s' % (np.array_equal(a, a0), np.abs(l - l0).max(), ' '.join(parts)), flush=True)
            worst = max(worst, float((g - g0).abs().max()) / float(g0.abs().max()))
    print('%-50s %d of %d runs differ from the first (largest meta-gradient difference %.2e of its scale)' % (st or '(defaults)', bad, N, worst), flush=True)
    for kv in filter(None, st.split(',')):
        k, v = kv.split('=')
        if k not in ('SERIALIZE', 'GEMM_MODE', 'ALT', 'UPDATE_STEP'):
            dflt = {'GM_AGG_STREAM': 1, 'GM_HEAD_STAGE': 1, 'GM_HEAD_THREADS': 1024}.get(k)
            if dflt is not None:
                lib.gm_set_tuning(k.encode(), dflt)
