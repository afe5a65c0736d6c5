import cProfile, pstats, sys, os, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmeta_amd
from gmeta_amd import synth
T = 32
args, cfg = synth.make_args('arxiv', task_num=T)
np.random.seed(222); random.seed(222)
data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T * 4, args=args, adjs=store, h=2,
                         tables={'train': (data['names'], data['labels'])}, verbose=False)
db.get_batch(list(range(T)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(1, 4):
    x = db.get_batch(list(range(b * T, (b + 1) * T)))
torch.cuda.synchronize()
print('get_batch: %.2f ms per meta-batch' % ((time.perf_counter() - t0) / 3 * 1e3))
pr = cProfile.Profile(); pr.enable()
x = db.get_batch(list(range(T)))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
