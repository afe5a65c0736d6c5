#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg) -- one worker process of the TASK-PARALLEL CPU baseline.

The reference loops the tasks of a meta-batch serially (meta.py:118), but they are independent until the mean over tasks
(meta.py:161), so the honest "host cores of the same box" figure runs them side by side: bench.py starts W of these workers,
each with its own slice of the meta-batch's tasks and `threads` BLAS / OpenMP threads, waits until all have built their
inputs (untimed, like the pre-extracted subgraphs on the GPU side), releases them together and takes the wall time until the
last one is done.  Every worker runs oracle/gmeta_oracle.py:task_inner_loop -- the same restatement the parity tests use.

    python oracle/cpu_task_worker.py <inputs.npz> <comma-separated task ids>      (thread count via OMP_NUM_THREADS etc.)
protocol: prints READY, waits for a line on stdin, runs, prints one JSON line {"tasks": [...], "seconds": [...]}"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gmeta_oracle as orc  # noqa: E402


def main():
    z = np.load(sys.argv[1], allow_pickle=False)
    tasks = [int(t) for t in sys.argv[2].split(',') if t != '']
    hp = json.loads(str(z['hp']))
    config = [(n, list(p)) for n, p in json.loads(str(z['config']))]
    graphs = [orc.Graph(int(z['g%d_n' % g]), z['g%d_src' % g], z['g%d_dst' % g]) for g in range(int(z['n_graphs']))]
    feats = [z['feat%d' % g] for g in range(int(z['n_graphs']))]
    theta = [z['theta%d' % k] for k in range(int(z['n_theta']))]
    inputs = []
    for t in tasks:
        bb = []
        for side in ('s', 'q'):
            seeds = [tuple(int(v) for v in s) for s in z['t%d_%s_seeds' % (t, side)]]
            flat, off = z['t%d_%s_nodes' % (t, side)], z['t%d_%s_off' % (t, side)]
            bb.append(orc.Batch(graphs, seeds, [flat[off[k]:off[k + 1]] for k in range(len(seeds))]))
        inputs.append((bb[0], bb[1], bb[0].features(feats), bb[1].features(feats), z['t%d_ys' % t], z['t%d_yq' % t]))
    if inputs:          # one short warm-up (library initialisation, page faults), K = 2
        bs, bq, xs, xq, ys, yq = inputs[0]
        orc.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, config, hp['k_spt'], hp['update_lr'], 2, True)
    print('READY', flush=True)
    sys.stdin.readline()
    secs = []
    for bs, bq, xs, xq, ys, yq in inputs:
        t0 = time.perf_counter()
        orc.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, config, hp['k_spt'], hp['update_lr'], hp['K'], True)
        secs.append(time.perf_counter() - t0)
    print(json.dumps({'tasks': tasks, 'seconds': secs}), flush=True)


if __name__ == '__main__':
    main()
