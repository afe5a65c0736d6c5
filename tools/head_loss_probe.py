#!/usr/bin/env python3
"""Phase timeline of k_head_loss (one 1024-thread workgroup per task): device-clock stamps of block 0 of the last launch of a meta-step.
    python tools/head_loss_probe.py [config] [task_num]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401  (probe build of the library: gm_debug_stamp / gm_stream_debug / gm_head_loss_debug)
import gmeta_amd
from gmeta_amd import _lib, synth

name = sys.argv[1] if len(sys.argv) > 1 else 'firstmm'
over = {'task_num': int(sys.argv[2])} if len(sys.argv) > 2 else {}
args, cfg = synth.make_args(name, **over)
np.random.seed(222); import random; random.seed(222); torch.manual_seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
link = bool(cfg.get('link'))
config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg), link=link)
m = gmeta_amd.Meta(args, config).to('cuda')
T = cfg['task_num']
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=T, args=args, adjs=store, h=cfg['h'],
                         tables=data['tables'], verbose=False)
b = db.get_batch(list(range(T)))
lib = _lib.lib()
for _ in range(3):
    m(*b, data['feats'])
lib.gm_head_loss_debug(1, None)
m.serialize = 1
for rep in range(3):
    m(*b, data['feats']); torch.cuda.synchronize()
    st = np.zeros(64, np.uint64)
    lib.gm_head_loss_debug(1, st.ctypes.data_as(C.c_void_p))
    s = st.astype(np.int64)
    print('%s last k_head_loss launch (the differentiated query loss), block 0, us since its first instruction: loads 1 %.2f | centre rows %.2f | logits %.2f | loss %.2f | head weight gradients %.2f | dQ rows %.2f   (%d shader cycles: %.0f MHz)'
          % (name, *((s[k] - s[0]) / 100.0 for k in (1, 2, 3, 4, 5, 6)), s[7], s[7] / max(1e-9, (s[6] - s[0]) / 100.0)))
    if rep == 2:
        print('    logits phase per wave, start / end (us): ' + ' '.join('%.2f/%.2f' % ((s[8 + w] - s[0]) / 100.0, (s[24 + w] - s[0]) / 100.0) for w in range(16)))
