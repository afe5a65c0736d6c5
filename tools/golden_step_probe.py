"""One golden fixture through the HIP meta-step under every schedule flag, repeatedly: largest deviation of the meta-gradient from the reference value and where
(used to find the ill-conditioned entry of g1_sampled_h2: model.hip, head_fwd_sub).     python tools/golden_step_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('tests', '', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, d))
os.chdir(os.path.join(ROOT, 'tests'))
import hip_util as hu
from test_hip_parity import Fixture
fx = Fixture('g1_sampled_h2')
ref_g = np.concatenate([g.reshape(-1) for g in fx.grad])
sizes = [g.size for g in fx.grad]
print('param sizes', sizes)
for rep in range(4):
    for (h, s, c) in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1)]:
        res = hu.hip_meta_step(fx, replay=True, hoist=h, sparse_bwd=s, cone=c)
        d = np.abs(res['grad'] - ref_g)
        i = int(d.argmax())
        print(rep, (h, s, c), 'max diff %.3e at %d: ours %.6f ref %.6f' % (d.max(), i, res['grad'][i], ref_g[i]), 'losses', np.asarray(res['stats']['losses_q'])[-2:], flush=True)
