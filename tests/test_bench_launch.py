"""CPU: `python bench.py --gpus N` started PLAINLY (no torch.distributed.run, no WORLD_SIZE in the environment) must start its own N ranks
-- the first contact with a multi-GPU node may well use that command shape (the N = 1 command is plain).  GMETA_BENCH_LAUNCH_PROBE=1 stops
every rank after the rendezvous and the task sharding (gloo), before anything touches a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                           'TORCHELASTIC_RUN_ID', 'GROUP_RANK', 'LOCAL_WORLD_SIZE')}
    env.update(GMETA_BENCH_LAUNCH_PROBE='1', PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
    return env


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert lines, out[-3000:]
    return json.loads(lines[-1])


def test_plain_invocation_spawns_its_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no_cpu_baseline'],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    j = _last_json(out)
    assert j['n_gpus'] == 2 and j['ranks_reporting'] == [0, 1] and j['local_ranks'] == [0, 1]
    assert j['tasks_per_rank'] == [16, 16] and sum(j['tasks_per_rank']) == j['task_num'] == 32
    assert sum(1 for l in out.strip().splitlines() if l.startswith('{')) == 1          # ONE line, from rank 0


def test_uneven_shards_and_the_torchrun_shape_still_work():
    # the driver's own command for N > 1, three ranks over 32 tasks: shards of 11 / 10 / 11
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--steps', '1', '--warmup', '0'],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    j = _last_json(out)
    assert j['n_gpus'] == 3 and sorted(j['tasks_per_rank']) == [10, 11, 11] and sum(j['tasks_per_rank']) == 32


def test_world_size_mismatch_is_an_error_with_a_usable_message():
    env = _clean_env(); env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode != 0 and 'starts its own ranks' in p.stdout.decode()
