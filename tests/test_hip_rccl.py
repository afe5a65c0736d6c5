"""GPU (-m gpu, needs >= 2 GPUs; skipped on a 1-GPU box): the sharded meta-step over REAL RCCL, so that the first multi-GPU
bench run is not also the first NCCL run: 2 ranks, uneven task shards, one all-reduce of
[grad | losses_q | corrects | count], global-T mean, device-side NaN guard, identical Adam step on every rank; sharded
evaluation with an all_gather.  Compared with the reference's golden single-process outputs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from golden_util import Fixture

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL refuses two ranks on one device)')


def _run(case, world, tmp_path, one_gpu=False):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = str(s.getsockname()[1]); s.close()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if one_gpu:
        env['GMETA_TEST_ONE_GPU'] = '1'
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'rccl_worker.py'), str(r), str(world), port, case, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]


def _check_two_ranks(case, res):
    fx = Fixture(case)
    for r in res:
        assert float(r['task_num']) == fx.T
        np.testing.assert_allclose(r['accs'], fx.z['accs'], atol=1e-6)
        np.testing.assert_allclose(float(r['loss_q']), fx.z['loss_q'][:, -1].mean(), atol=1e-4)
        for k, (v1, g) in enumerate(zip(fx.vars1, fx.grad)):
            m = np.abs(g) > 1e-5
            np.testing.assert_allclose(r['v%d' % k][m], v1[m], atol=1e-4, rtol=0)
        assert r['ft'].shape == (fx.T, fx.K_test + 1)
    for k in range(len(fx.vars1)):
        assert np.array_equal(res[0]['v%d' % k], res[1]['v%d' % k])       # replicas stay bit-identical
    assert np.array_equal(res[0]['ft'], res[1]['ft'])


def _check_nan(res):
    fx = Fixture('g6_nan_skip')
    for r in res:
        assert np.isnan(float(r['loss_q']))
        for k, v0 in enumerate(fx.vars0):
            assert np.array_equal(r['v%d' % k], v0)                       # fused Adam skipped its step on found_inf (meta.py:163-164)


@needs2
@pytest.mark.parametrize('case', ['g2_shared', 'g0_disjoint_h1'])
def test_two_rccl_ranks_equal_single_process_reference(case, tmp_path):
    _check_two_ranks(case, _run(case, 2, tmp_path))


@needs2
def test_nan_guard_over_rccl(tmp_path):
    _check_nan(_run('g6_nan_skip', 2, tmp_path))


@pytest.mark.parametrize('case', ['g2_shared', 'g0_disjoint_h1', 'g3_linkpred', 'g5_in_gt_out'])
def test_two_ranks_sharing_one_gpu_equal_single_process_reference(case, tmp_path):
    """The same two-rank checks where only ONE GPU exists: both ranks run their task shard through the HIP path on cuda:0 and exchange
    the [grad | losses_q | corrects | count] block over gloo (uneven shards for g2_shared: T = 3; g3_linkpred = the task-sharded
    link-prediction setup of BASELINE configs[4]: pair centres, head [2, 2H], one task per rank; g5 = multiply-first layers)."""
    _check_two_ranks(case, _run(case, 2, tmp_path, one_gpu=True))


def test_nan_guard_with_two_ranks_sharing_one_gpu(tmp_path):
    _check_nan(_run('g6_nan_skip', 2, tmp_path, one_gpu=True))


def test_single_rank_rccl_group_runs_the_allreduce_path(tmp_path):
    """One rank, real RCCL communicator (works on a 1-GPU box): the all-reduce / all_gather code path of Meta.forward and
    finetunning_batch(shard=True) executes over NCCL and reproduces the golden outputs."""
    fx = Fixture('g2_shared')
    res = _run('g2_shared', 1, tmp_path)
    np.testing.assert_allclose(res[0]['accs'], fx.z['accs'], atol=1e-6)
    for k, (v1, g) in enumerate(zip(fx.vars1, fx.grad)):
        m = np.abs(g) > 1e-5
        np.testing.assert_allclose(res[0]['v%d' % k][m], v1[m], atol=1e-4, rtol=0)
