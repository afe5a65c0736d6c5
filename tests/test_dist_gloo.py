"""CPU: world_size-2 gloo run of the sharded meta-step: every rank gets a contiguous task shard, ONE
all-reduce carries [grad | losses_q | corrects | task count], the division uses the GLOBAL task count, the NaN guard
is evaluated on the reduced loss, and every rank applies the identical Adam step.  Compared with the reference's
golden single-process outputs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from golden_util import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return str(p)


def _run(case, world, tmp_path):
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), str(r), str(world), port, case, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]


@pytest.mark.parametrize('case', ['g2_shared', 'g0_disjoint_h1'])
def test_two_ranks_equal_single_process_reference(case, tmp_path):
    fx = Fixture(case)
    res = _run(case, 2, tmp_path)                      # g2_shared: T=3 -> shards of 2 and 1 tasks (uneven)
    for r in res:
        assert float(r['task_num']) == fx.T            # global task count came through the all-reduce
        np.testing.assert_allclose(r['accs'], fx.z['accs'], atol=1e-6)
        np.testing.assert_allclose(float(r['loss_q']), fx.z['loss_q'][:, -1].mean(), atol=1e-4)
        for k, (v1, g) in enumerate(zip(fx.vars1, fx.grad)):
            m = np.abs(g) > 1e-5
            np.testing.assert_allclose(r['v%d' % k][m], v1[m], atol=1e-4, rtol=0)
    for k in range(len(fx.vars1)):                     # replicas stay bit-identical
        assert np.array_equal(res[0]['v%d' % k], res[1]['v%d' % k])


def test_nan_guard_is_taken_on_the_reduced_loss(tmp_path):
    fx = Fixture('g6_nan_skip')
    res = _run('g6_nan_skip', 2, tmp_path)
    for r in res:
        assert np.isnan(float(r['loss_q']))
        for k, v0 in enumerate(fx.vars0):
            assert np.array_equal(r['v%d' % k], v0)    # optimiser step skipped on every rank (meta.py:163-164)
