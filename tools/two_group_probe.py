#!/usr/bin/env python3
"""Probe: one 32-task gm_meta_step against G concurrent gm_meta_steps over task groups (own streams and workspaces) at the arxiv shape.
GM_GEMM_SPLIT_GRID caps the persistent GEMM's grid so that the groups' kernels can share the chip."""
import copy, os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth
T = 32
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
args, cfg = synth.make_args('arxiv')
np.random.seed(222); random.seed(222); torch.manual_seed(222)
data = synth.make_dataset(cfg)
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg))
metas = [gmeta_amd.Meta(args, config).to('cuda') for _ in range(G)]
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'], batchsz=T, args=args, adjs=store,
                         h=cfg['h'], tables=data['tables'], verbose=False)
whole = db.get_batch(list(range(T)))
bounds = np.linspace(0, T, G + 1).astype(int)
groups = [db.get_batch(list(range(bounds[g], bounds[g + 1]))) for g in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]
K = args.update_step


def one():
    metas[0]._run(whole[0], whole[1], whole[2], whole[3], K, 1)


def grouped():
    cur = torch.cuda.current_stream()
    for g in range(G):
        streams[g].wait_stream(cur)
        with torch.cuda.stream(streams[g]):
            b = groups[g]
            metas[g]._run(b[0], b[1], b[2], b[3], K, 1)
    for g in range(G):
        cur.wait_stream(streams[g])


for name, fn in (('one %d-task step' % T, one), ('%d concurrent groups' % G, grouped)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-24s %.3f ms per step' % (name, (time.perf_counter() - t0) / n * 1e3))
