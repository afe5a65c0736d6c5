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


def _run(case, world, tmp_path, mode='train'):
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='2')
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), str(r), str(world), port, case, str(tmp_path), mode],
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


def _oracle_tasks(fx, tasks):
    import gmeta_oracle as orc
    graphs = fx.graphs()
    spt = [orc.extract_batch(graphs, fx.z['spt_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('spt', t)) for t in tasks]
    qry = [orc.extract_batch(graphs, fx.z['qry_seeds'][t], fx.args['h'], fx.args['sample_nodes'], 222, fx.link, replay_nodes=fx.replay_lists('qry', t)) for t in tasks]
    return graphs, spt, qry


def test_empty_shard_contributes_zeros(tmp_path):
    """A short trailing meta-batch (train.py keeps it, like DataLoader(drop_last=False)) can leave a rank without tasks: it joins
    the all-reduce with zeros and every rank ends with the single-task update."""
    import gmeta_oracle as orc
    fx = Fixture('g2_shared')
    res = _run('g2_shared', 2, tmp_path, mode='short')
    graphs, spt, qry = _oracle_tasks(fx, [0])
    accs, grad, theta1, _ = orc.meta_step(graphs, fx.feats, spt, qry, [fx.z['y_spt'][0]], [fx.z['y_qry'][0]], fx.vars0, fx.config, fx.args['k_spt'],
                                          fx.args['update_lr'], fx.args['meta_lr'], fx.K, adam_state={})
    for r in res:
        assert float(r['task_num']) == 1.0
        np.testing.assert_allclose(r['accs'], accs, atol=1e-6)
        for k, (v1, g) in enumerate(zip(theta1, grad)):
            m = np.abs(g) > 1e-5
            np.testing.assert_allclose(r['v%d' % k][m], v1[m], atol=1e-4, rtol=0)
    for k in range(len(theta1)):
        assert np.array_equal(res[0]['v%d' % k], res[1]['v%d' % k])


def test_sharded_evaluation_gathers_every_task(tmp_path):
    """finetunning tasks are split over the ranks (uneven: 3 tasks on 2 ranks) and the per-task accuracies
    all-gathered in task order -- every rank returns the full [T, K_test+1] array == the single-process loop (train.py:118-121)."""
    import gmeta_oracle as orc
    fx = Fixture('g2_shared')
    res = _run('g2_shared', 2, tmp_path, mode='eval')
    graphs, spt, qry = _oracle_tasks(fx, range(fx.T))
    want = np.stack([orc.finetune(graphs, fx.feats, spt[t], qry[t], fx.z['y_spt'][t], fx.z['y_qry'][t], fx.vars0, fx.config, fx.args['k_spt'],
                                  fx.args['update_lr'], fx.K_test) for t in range(fx.T)])
    for r in res:
        assert r['accs'].shape == (fx.T, fx.K_test + 1)
        np.testing.assert_allclose(r['accs'], want, atol=1e-6)
