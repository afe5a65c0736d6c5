#!/usr/bin/env python3
"""bench.py -- meta-tasks/sec of the G-Meta inner-loop hot path on MI355X.

One "step" = one Meta.forward (ProtoMAML meta-step: K inner SGD steps on the support subgraphs, K+1 query
evaluations, first-order meta-gradient, Adam) over a meta-batch of task_num tasks whose h-hop subgraphs are
already extracted and resident in HBM.  Default workload = BASELINE.json configs[1]: arxiv-ogbn shape
(synthetic graph of 169,343 nodes, F0=128, h=2, hidden 256, 3-way 3-shot 24-query, task_num=32, K=10,
sample_nodes=1000); --config selects the other BASELINE configs (syn0 = configs[0], tissue = configs[3],
firstmm = configs[4]).  With --gpus N the tasks of a meta-batch are sharded over N ranks (strong scaling;
uneven shards when N does not divide task_num) and the meta-gradient is summed by one RCCL all-reduce per
step.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          (starts its own 8 ranks: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3        # dense fp32 matrix peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_quota():
    """CPUs this process may actually use: the cgroup quota when there is one (a container can see 256 CPUs and own 16), else the affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            return min(float(n), float(q) / float(per)), True
    except (OSError, ValueError):
        pass
    return float(n), False


def extraction_bytes(store, db, idx, batch, h, link):
    """ALGORITHMIC bytes of extracting one meta-batch (with the store's real index widths: int64 row bounds, int32 ids):
      expand   per subgraph, every frontier node v of the h-hop expansion once: its two row bounds + its in-neighbour list = 16 + 4 deg_in(v)
               (node classification: the centre and, for h >= 2, its distinct 1-hop in-neighbours, for h = 3 also the 2-hop ones;
               link prediction: i side 2 hops, j side 1 hop -- sdp.py:327-333)
      induce   per selected node v (batch row): row bounds + adjacency list in BOTH orientations (the by-source CSR the backward aggregate
               needs is induced from the parent's out-edge lists) = 32 + 4 (deg_in(v) + deg_out(v)), read once
      write    the batched CSR in both orientations, parents, feature rows, norms, centres
    Node sets are whatever the GPU extraction produced (bit-exact vs the oracle's, tests/test_hip_parity.py); the sampling itself works on
    the LDS bitmap and moves no HBM bytes."""
    S, Q = batch[0][0].view_of or batch[0][0], batch[2][0].view_of or batch[2][0]
    deg_in = [np.diff(p).astype(np.int64) for p, _ in store.host_csr]
    deg_out = [np.bincount(ix.astype(np.int64), minlength=len(p) - 1).astype(np.int64) for p, ix in store.host_csr]
    expand = 0
    for t in idx:
        a = db._task_arrays(t)
        for seeds in (a[0], a[1]):
            for g, i, j in seeds:
                ptr, ix = store.host_csr[int(g)]; di = deg_in[int(g)]

                def hop(front):
                    return np.unique(np.concatenate([ix[ptr[v]:ptr[v + 1]] for v in front])) if len(front) else np.zeros(0, np.int64)
                front = [np.array([int(i)], np.int64)]
                hops = 2 if link else h
                seen = front[0]
                for _ in range(hops - 1):
                    nxt = np.setdiff1d(hop(front[-1]), seen); seen = np.union1d(seen, nxt); front.append(nxt)
                exp_nodes = np.concatenate(front)
                if link:
                    exp_nodes = np.concatenate([exp_nodes, np.array([int(j)], np.int64)])
                expand += 16 * len(exp_nodes) + 4 * int(di[exp_nodes].sum())
    induce = write = 0
    for B in (S, Q):
        par = np.asarray(B.parent(), np.int64); gid = np.repeat(np.asarray(B.graph_ids(), np.int64), np.diff(B.sub_off))
        for g in np.unique(gid):
            pv = par[gid == g]
            induce += 32 * len(pv) + 4 * int(deg_in[int(g)][pv].sum() + deg_out[int(g)][pv].sum())
        write += 2 * 4 * (B.rows + 1) + 2 * 4 * B.edges + 3 * 4 * B.rows + 4 * B.subs * (2 if link else 1)
    return {'expand': int(expand), 'induce': int(induce), 'write': int(write)}


def launch_classes(launches):
    """The aggregate launches of one serialised step grouped by their algorithmic bytes: count, bytes, time and rate per class."""
    out = []
    for name, lo_b, hi_b in (('< 64 MB', 0, 64e6), ('64-512 MB', 64e6, 512e6), ('0.5-1.5 GB', 512e6, 1.5e9), ('>= 1.5 GB', 1.5e9, 1e18)):
        sel = [(m, w) for m, w in launches if lo_b <= w < hi_b and m > 0]
        if sel:
            ms, by = sum(m for m, _ in sel), sum(w for _, w in sel)
            out.append({'algorithmic_bytes': name, 'launches': len(sel), 'gb': round(by / 1e9, 2), 'ms': round(ms, 3), 'gb_per_s': round(by / (ms * 1e-3) / 1e9, 1),
                        'frac': round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return out


def schedule_roofline(lib, maml, step, drain, n_steps, ms_per_step):
    """Roofline accounting of a FLAGGED schedule (the receptive-field schedules): a few serialised steps with HIP events around every launch,
    the aggregate priced on the rows each level actually touches (gm_profile_read category 0 under gm_hparams_t.cone: destination-level
    row bounds + norms, the edges between the two levels, source-level rows read once, destination rows written once), the update GEMMs
    and weight gradients on their level rows (A read once + C written once / A and G read once), and the split of the step's kernel time."""
    def rd(cat):
        ms, n, w = C.c_double(), C.c_int64(), C.c_int64()
        lib.gm_profile_read(cat, C.byref(ms), C.byref(n), C.byref(w))
        return ms.value, n.value, w.value
    lib.gm_profile_enable(1)
    ser0 = maml.serialize
    maml.serialize = 1
    step(0); drain()
    tot = {c: [0.0, 0, 0] for c in (0, 1, 2, 4, 5, 6, 7, 13, 14, 15)}
    for k in range(n_steps):
        step(k); drain()
        for c in tot:
            ms, n, w = rd(c)
            tot[c][0] += ms; tot[c][1] += n; tot[c][2] += w
    maml.serialize = ser0
    lib.gm_profile_enable(0)
    per = lambda c, i: tot[c][i] / n_steps
    agg_ms, agg_n, agg_by = per(0, 0), per(0, 1), per(0, 2)
    g_ms = per(1, 0) + per(4, 0) + per(6, 0); g_fl = per(1, 2) + per(4, 2) + per(6, 2); g_n = per(1, 1) + per(4, 1) + per(6, 1)
    w_ms = per(2, 0) + per(5, 0) + per(7, 0); w_fl = per(2, 2) + per(5, 2) + per(7, 2); w_n = per(2, 1) + per(5, 1) + per(7, 1)
    g_by, w_by, h_ms, h_n = per(13, 2), per(14, 2), per(15, 0), per(15, 1)
    ach = agg_by / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else None
    t_hbm = (agg_by + g_by + w_by) / (HBM_PEAK_GBS * 1e9) * 1e3
    t_mfma = (g_fl + w_fl) / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3
    return {
        'roofline': {'bound': 'hbm', 'kernel': 'k_agg on the receptive-field levels (level-to-level compact CSRs)', 'achieved': round(ach, 1) if ach else None,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None, 'launches_per_step': int(agg_n),
                     'avg_launch_us': round(agg_ms / max(agg_n, 1) * 1e3, 2), 'algorithmic_bytes_per_step': int(agg_by),
                     'priced_on': 'the rows each level touches: 4 (n_dst + 1) + 4 e + 4 n_dst + 4 F (n_src + n_dst) per launch (SURVEY 8(d) B_agg on level_rows / level_edges)',
                     'measured': 'HIP events around every launch of %d serialised steps' % n_steps},
        'update': {'gemm': {'launches_per_step': int(g_n), 'ms_per_step': round(g_ms, 4), 'gflop_per_step': round(g_fl / 1e9, 2), 'tflops': round(g_fl / (g_ms * 1e-3) / 1e12, 2) if g_ms > 0 else None,
                            'a_plus_c_bytes_per_step': int(g_by), 'gb_per_s': round(g_by / (g_ms * 1e-3) / 1e9, 1) if g_ms > 0 else None,
                            'frac_of_hbm_peak': round(g_by / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if g_ms > 0 else None},
                   'wgrad': {'launches_per_step': int(w_n), 'ms_per_step': round(w_ms, 4), 'gflop_per_step': round(w_fl / 1e9, 2), 'tflops': round(w_fl / (w_ms * 1e-3) / 1e12, 2) if w_ms > 0 else None,
                             'a_plus_g_bytes_per_step': int(w_by), 'gb_per_s': round(w_by / (w_ms * 1e-3) / 1e9, 1) if w_ms > 0 else None,
                             'frac_of_hbm_peak': round(w_by / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if w_ms > 0 else None}},
        'kernel_time_ms_per_step': {'aggregate': round(agg_ms, 4), 'gemm': round(g_ms, 4), 'wgrad_incl_reduction': round(w_ms, 4), 'head_loss': round(h_ms, 4),
                                    'head_loss_launches': int(h_n), 'sum_of_timed_launches': round(agg_ms + g_ms + w_ms + h_ms, 4),
                                    'timed_launches_per_step': int(agg_n + g_n + w_n + h_n)},
        'step_bound': {'ms_at_hbm_peak': round(t_hbm, 4), 'ms_at_f32_mfma_peak': round(t_mfma, 4), 'frac_of_serial_bound': round((t_hbm + t_mfma) / ms_per_step, 4),
                       'note': 'every launch of this schedule is a few microseconds of work behind a dependent-launch boundary: the schedule is bound by its launch chain, not by a pipe'},
    }


def box_calibration(lib, _lib, qry_batch, seconds=0.6):
    """Two fixed micro-workloads, ~0.6 s each, run right after the timed region on the same (hot) chip: what THIS box delivers, so that lines taken
    on different leases of the pool can be read against each other (box to box the headline moved +-3 % with identical code in rounds 3-5).
      hbm_copy     1 GiB -> 1 GiB device copy (torch's copy kernel), bytes read + written per second
      split_gemm   the library's persistent three-piece split GEMM alone: [rows, 256] @ [256, 256] over the query batch's row tiles (gm_dense_update,
                   mode 1), fp32-equivalent TFLOP/s (x6 = bf16 MFMA TFLOP/s issued); only when the batch is large enough to fill the chip"""
    import torch
    out = {'device': torch.cuda.get_device_name(), 'what': box_calibration.__doc__.split('\n')[0].strip()}
    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device='cuda').normal_(); dst = torch.empty_like(src)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 0
    ev0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            dst.copy_(src)
        reps += 20
        torch.cuda.synchronize()
    ev1.record(); torch.cuda.synchronize()
    out['hbm_copy'] = {'GBps': round(2.0 * n * 4 * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1), 'bytes_per_copy': n * 4, 'copies': reps}
    del src, dst
    rows = int(qry_batch.rows)
    if rows >= 262144:
        K = N = 256
        x = torch.empty(rows, K, dtype=torch.float32, device='cuda').normal_(); w = torch.empty(K, N, dtype=torch.float32, device='cuda').normal_() * 0.05
        o = torch.empty(rows, N, dtype=torch.float32, device='cuda')
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: _lib.check(lib.gm_dense_update(qry_batch.handle, _lib.ptr(x), K, _lib.ptr(w), 0, N, _lib.ptr(o), 1, st), 'dense_update')
        for _ in range(3):
            call()
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 0
        ev0.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                call()
            reps += 20
            torch.cuda.synchronize()
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        tf = 2.0 * rows * K * N / (ms * 1e-3) / 1e12
        out['split_gemm'] = {'fp32_equivalent_tflops': round(tf, 1), 'bf16_mfma_tflops': round(6 * tf, 1), 'ms_per_call': round(ms, 4), 'rows': rows, 'K': K, 'N': N,
                             'calls': reps, 'note': 'weight split + plain (non-fused) launch that stores C; back to back = hot chip, as inside a meta-step'}
    return out


def shard_bounds(T, world):
    """Contiguous task ranges of a meta-batch per rank (sizes differ by at most one)."""
    return np.linspace(0, T, world + 1).round().astype(int)


def cpu_baseline(db, data, cfg, config, batch, budget_s=45.0):
    """The CPU restatement of the same workload, timed on the host cores (kind "port"; the reference's own Python
    needs DGL 0.4.3, which cannot be installed here).  Two variants of the SAME per-task inner loop
    (K support steps fwd+bwd, K+1 query evaluations, the first-order meta-gradient):
      numpy-omp  oracle/gmeta_oracle.py: numpy/BLAS matmuls + the OpenMP C aggregate (forward and transposed)
      torch-cpu  oracle/torch_cpu_baseline.py: torch CPU ops + autograd, index_add_ for update_all -- the closest
                 analogue of the reference's DGL-CPU path (learner.py:38-47)
    on whole tasks of the first GPU meta-batch (same subgraphs: node sets replayed from the GPU extraction, which is
    bit-exact vs the oracle's; extraction is excluded on both sides).  The thread count of each variant is tuned first on
    a short probe (the same task at K=2): on a 256-core host the full-width pools are pathological for these op sizes
    (measured: torch-cpu 49 s/task at 256 threads vs 0.6 s at 8).  Then up to 2 warm-up + >= 5 timed tasks, median."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch
    import threadpoolctl
    import gmeta_oracle as orc
    import torch_cpu_baseline as tcb
    host_cores = os.cpu_count()
    graphs = [orc.Graph(*g) for g in data['graphs']]
    S, Q = batch[0][0].view_of or batch[0][0], batch[2][0].view_of or batch[2][0]
    link = bool(cfg.get('link'))
    rng = np.random.default_rng(222)
    theta = []
    for name, p in config:
        if name == 'GraphConv':
            theta += [(rng.standard_normal(p) * np.sqrt(2.0 / sum(p))).astype(np.float32), np.zeros(p[1], np.float32)]
        elif name == 'Linear':
            theta += [(rng.standard_normal((p[1], p[0] * (2 if link else 1))) * 0.1).astype(np.float32), np.zeros(p[1], np.float32)]
    n_gcn = cfg['h']
    K, k_spt, lr = cfg['update_step'], cfg['k_spt'], cfg['update_lr']
    T = len(batch[0])
    cache = {}

    def task(t):
        if t not in cache:
            out = []
            for B, seeds in ((S, db._task_arrays(t)[0]), (Q, db._task_arrays(t)[1])):
                so, par, off = B.set_sub_off, B.parent(), B.sub_off
                lists = [par[off[k]:off[k + 1]] for k in range(so[t], so[t + 1])]
                out.append(orc.Batch(graphs, [tuple(int(v) for v in s) for s in seeds], lists))
            cache[t] = (out[0], out[1], out[0].features(data['feats']), out[1].features(data['feats']), np.asarray(batch[1][t]), np.asarray(batch[3][t]))
        return cache[t]

    def run(name, t, k_steps):
        bs, bq, xs, xq, ys, yq = task(t % T)
        t0 = time.perf_counter()
        if name == 'numpy-omp':
            orc.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, config, k_spt, lr, k_steps, True)
        else:
            tcb.task_inner_loop(bs, bq, xs, xq, ys, yq, theta, n_gcn, k_spt, lr, k_steps, True)
        return time.perf_counter() - t0

    variants = {}
    for name in ('numpy-omp', 'torch-cpu'):
        # thread-count probe: ascending, stop once a count is clearly worse than the best so far
        best_nt, best_dt, probe = None, None, {}
        for nt in [c for c in (8, 16, 32, 64, 128, 256) if c <= host_cores]:
            torch.set_num_threads(nt)
            with threadpoolctl.threadpool_limits(limits=nt):
                run(name, 0, 2)
                dt = run(name, 0, 2)
            probe[nt] = round(dt, 3)
            if best_dt is None or dt < best_dt:
                best_nt, best_dt = nt, dt
            elif dt > 1.3 * best_dt:
                break
        torch.set_num_threads(best_nt)
        times, t_used, t, n_warm = [], 0.0, 0, 2
        with threadpoolctl.threadpool_limits(limits=best_nt):
            while len(times) < 5 or (t_used < budget_s * 0.4 and len(times) < 9):
                dt = run(name, t, K)
                t += 1; t_used += dt
                if n_warm > 0 and dt * 8 < budget_s:      # warm-ups only when 2 + 5 tasks fit the budget
                    n_warm -= 1
                    continue
                n_warm = 0
                times.append(dt)
                if t_used > budget_s and len(times) >= 2:
                    break
        med = float(np.median(times))
        variants[name] = {'value': round(1.0 / med, 4), 'median_s_per_task': round(med, 3), 'tasks_timed': len(times), 'threads': best_nt,
                          'min_s': round(min(times), 3), 'max_s': round(max(times), 3), 'thread_probe_s_at_K2': probe}
    best = max(variants, key=lambda k: variants[k]['value'])
    # ---- the same restatement with the tasks of the meta-batch run SIDE BY SIDE (they are independent, meta.py:118-161): W worker
    # processes x the tuned thread count, all T tasks of the meta-batch, wall time from a common start to the last worker's end
    par = None
    quota, has_quota = cpu_quota()
    try:
        # task-level parallelism beats thread-level parallelism on these op sizes: one worker per task, 1 or 2 threads each (both timed on the
        # whole meta-batch, the better one reported); with a cgroup CPU quota the workers share `quota` cores
        runs = [cpu_baseline_task_parallel(db, data, cfg, config, batch, theta, nt, max(host_cores, len(batch[0]) * nt)) for nt in (1, 2)]
        par = max(runs, key=lambda r: r['value'])
        par['tried'] = [{'workers': r['workers'], 'threads_per_worker': r['threads_per_worker'], 'value': r['value']} for r in runs]
        par['cpus_available'] = quota
        par['cpus_available_source'] = 'cgroup cpu.max quota' if has_quota else 'sched_getaffinity'
    except Exception as e:       # a reported extra, never the product path
        par = {'value': None, 'error': repr(e)}
    return {'value': variants[best]['value'], 'unit': 'meta-tasks/s', 'cores': variants[best]['threads'], 'host_cores': host_cores,
            'cpus_available': quota, 'kind': 'port',
            'variant': best, 'cpu_model': cpu_model(), 'variants': variants, 'task_parallel': par,
            'sample': 'whole tasks of the first meta-batch of the same config (K=%d inner steps incl. the meta-gradient), one task at a '
                      'time like the reference loop (meta.py:118), subgraphs pre-extracted on both sides; per variant: thread count tuned on '
                      'a K=2 probe of one task (full-width pools are slower on these op sizes), then up to 2 warm-up tasks and the median of '
                      '>= 5 timed tasks (fewer only if the time budget runs out); `cores` = threads the best variant used' % K}


def cpu_baseline_task_parallel(db, data, cfg, config, batch, theta, threads, host_cores, max_workers=32):
    """All tasks of the first meta-batch on the host cores at once: oracle/cpu_task_worker.py processes (fresh interpreters: no fork of
    a process that holds a HIP context), `threads` BLAS/OpenMP threads each, as many workers as fit the cores (at most one per task)."""
    import subprocess
    import tempfile
    S, Q = batch[0][0].view_of or batch[0][0], batch[2][0].view_of or batch[2][0]
    T = len(batch[0])
    workers = max(1, min(T, max_workers, host_cores // max(threads, 1)))
    arrs = {'hp': json.dumps({'k_spt': cfg['k_spt'], 'update_lr': cfg['update_lr'], 'K': cfg['update_step']}),
            'config': json.dumps([[n, list(p)] for n, p in config]), 'n_graphs': len(data['graphs']), 'n_theta': len(theta)}
    for g, (n, src, dst) in enumerate(data['graphs']):
        arrs['g%d_n' % g] = n; arrs['g%d_src' % g] = np.asarray(src); arrs['g%d_dst' % g] = np.asarray(dst); arrs['feat%d' % g] = data['feats'][g]
    for k, v in enumerate(theta):
        arrs['theta%d' % k] = v
    for t in range(T):
        for side, B, seeds in (('s', S, db._task_arrays(t)[0]), ('q', Q, db._task_arrays(t)[1])):
            so, par, off = B.set_sub_off, B.parent(), B.sub_off
            lists = [par[off[k]:off[k + 1]] for k in range(so[t], so[t + 1])]
            arrs['t%d_%s_seeds' % (t, side)] = np.asarray(seeds, np.int64)
            arrs['t%d_%s_nodes' % (t, side)] = np.concatenate(lists).astype(np.int64)
            arrs['t%d_%s_off' % (t, side)] = np.cumsum([0] + [len(x) for x in lists]).astype(np.int64)
        arrs['t%d_ys' % t] = np.asarray(batch[1][t]); arrs['t%d_yq' % t] = np.asarray(batch[3][t])
    tmp = tempfile.mkdtemp(prefix='gmeta_cpu_')
    path = os.path.join(tmp, 'inputs.npz')
    np.savez(path, **arrs)
    env = dict(os.environ)
    for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        env[k] = str(threads)
    env['CUDA_VISIBLE_DEVICES'] = ''; env['HIP_VISIBLE_DEVICES'] = ''
    procs = []
    try:
        for w in range(workers):
            ids = ','.join(str(t) for t in range(w, T, workers))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'oracle', 'cpu_task_worker.py'), path, ids], stdin=subprocess.PIPE,
                                          stdout=subprocess.PIPE, text=True, env=env))
        for p in procs:                                    # inputs built, warm-up done
            line = p.stdout.readline()
            if line.strip() != 'READY':
                raise RuntimeError('cpu worker failed to start: %r' % line)
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write('go\n'); p.stdin.flush()
        per_task = []
        for p in procs:
            per_task += json.loads(p.stdout.readline())['seconds']
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close(); p.wait(timeout=30)
            except Exception:
                p.kill()
        try:
            os.remove(path); os.rmdir(tmp)
        except OSError:
            pass
    return {'value': round(T / wall, 3), 'unit': 'meta-tasks/s', 'wall_s': round(wall, 3), 'tasks': T, 'workers': workers, 'threads_per_worker': threads,
            'cores': workers * threads, 'host_cores': host_cores, 'median_s_per_task': round(float(np.median(per_task)), 3),
            'what': 'numpy-omp variant, all %d tasks of the meta-batch run side by side by %d worker processes x %d threads (the tasks are independent, '
                    'meta.py:118-161; the reference itself loops them serially), wall time from a common start to the last worker' % (T, workers, threads)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def relaunch_command(n, argv):
    """argv of the launcher a plain `python bench.py --gpus N ...` turns into (the driver's own command line for N > 1, with a free port)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
            '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)


def launch_probe(a, rank, world, local):
    """GMETA_BENCH_LAUNCH_PROBE=1 (tests/test_bench_launch.py): everything bench.py does BEFORE it touches a GPU -- the rendezvous, the task
    sharding, the one-line report from rank 0 -- over gloo, so that the launch path of `python bench.py --gpus N` is covered on a CPU box."""
    import torch
    import torch.distributed as dist
    from gmeta_amd import synth
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    over = {'task_num': a.task_num} if a.task_num else {}
    _, cfg = synth.make_args(a.config, **over)
    bounds = shard_bounds(cfg['task_num'], world)
    mine = torch.tensor([rank, local, int(bounds[rank + 1] - bounds[rank])], dtype=torch.int64)
    got = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(got, mine)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'probe': 'launch', 'n_gpus': world, 'task_num': int(cfg['task_num']), 'ranks_reporting': [int(g[0]) for g in got],
                          'local_ranks': [int(g[1]) for g in got], 'tasks_per_rank': [int(g[2]) for g in got]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='arxiv', choices=['arxiv', 'syn0', 'tissue', 'firstmm'])
    ap.add_argument('--task_num', type=int, default=None)
    ap.add_argument('--hoist_z1', type=int, default=0)
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_eval', action='store_true', help='skip the evaluation (finetunning) leg of the configs that have one')
    ap.add_argument('--n_batches', type=int, default=2, help='distinct pre-extracted meta-batches cycled through')
    ap.add_argument('--sparse_bwd', type=int, default=0, help='1: time the flagged exact row-sparse backward schedule instead of the default dense one')
    ap.add_argument('--cone', type=int, default=0, help='1: time the flagged receptive-field schedule (layer l only on the rows that reach a centre)')
    ap.add_argument('--serialize', type=int, default=0, help='1: run the timed region on one stream (what the rocprofv3 per-kernel summaries use)')
    ap.add_argument('--extra_steps', type=int, default=3, help='steps per flagged exact schedule reported under "extra" (N=1 only; 0 = skip)')
    ap.add_argument('--e2e_steps', type=int, default=10, help='N=1 only: extra steps with a FRESH extraction per step (prefetched on a second thread), reported under "end_to_end"; 0 = skip')
    ap.add_argument('--defer', type=int, default=0, help='1: Meta.forward_deferred, accuracies of step k read after step k+1 is queued')
    ap.add_argument('--roofline_steps', type=int, default=2, help='extra serialised steps after the timed region for the per-kernel roofline')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on this node, rendezvous on
        # 127.0.0.1 at a free port; rank 0 of the relaunched job prints the one JSON line (exec: this process IS the launcher from here on)
        os.execv(sys.executable, relaunch_command(a.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with torch.distributed.run --nproc-per-node %d, or plainly '
                         '(python bench.py --gpus %d starts its own ranks)' % (a.gpus, world, a.gpus, a.gpus))
    if os.environ.get('GMETA_BENCH_LAUNCH_PROBE') == '1':
        return launch_probe(a, rank, world, local)

    import gmeta_amd
    from gmeta_amd import _lib, synth

    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get('GMETA_FORCE_DIST') == '1'      # the latter: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    over = {'hoist_z1': a.hoist_z1, 'serialize': a.serialize, 'sparse_bwd': a.sparse_bwd, 'cone': a.cone}
    if a.task_num:
        over['task_num'] = a.task_num
    args, cfg = synth.make_args(a.config, **over)
    T = cfg['task_num']
    if world > T:
        raise SystemExit('task_num=%d cannot be sharded over %d ranks (every rank needs at least one task)' % (T, world))
    bounds = shard_bounds(T, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    # ---- identical synthetic data + task lists on every rank (seed 222), each rank keeps its task shard
    np.random.seed(222); import random; random.seed(222); torch.manual_seed(222)
    t0 = time.perf_counter()
    data = synth.make_dataset(cfg)
    link = bool(cfg.get('link'))
    store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
    config = synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], synth.n_out(cfg), link=link)
    maml = gmeta_amd.Meta(args, config).to('cuda')
    maml.force_allreduce = os.environ.get('GMETA_FORCE_DIST') == '1' and os.environ.get('GMETA_SKIP_ALLREDUCE') != '1'
    n_eval = int(cfg.get('eval_tasks', 0)) if (world == 1 and not a.no_eval) else 0
    db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=cfg['n_way'], k_shot=cfg['k_spt'], k_query=cfg['k_qry'],
                             batchsz=T * (a.n_batches + (a.e2e_steps + 2 if world == 1 else 0)) + n_eval, args=args, adjs=store, h=cfg['h'],
                             tables=data['tables'], verbose=False)
    batches, ext_ms = [], []
    db.get_batch(list(range(lo, hi)))      # warm-up extraction (first call pays one-off setup)
    for b in range(a.n_batches):
        idx = list(range(b * T + lo, b * T + hi))
        torch.cuda.synchronize(); te = time.perf_counter()
        batches.append(db.get_batch(idx))
        torch.cuda.synchronize(); ext_ms.append((time.perf_counter() - te) * 1e3)
    setup_s = time.perf_counter() - t0
    rows = sum(x.rows for x in (batches[0][0][0].view_of, batches[0][2][0].view_of))
    edges = sum(x.edges for x in (batches[0][0][0].view_of, batches[0][2][0].view_of))

    pending = []

    def step(k):
        if a.defer:
            # deferred read-back (Meta.forward_deferred): the accuracies of step k are read while step k+1 is already queued,
            # like a training loop that only prints them every few steps (train.py:110); all work of every step still runs
            # inside the timed region (the last handle is drained before the clock stops)
            pending.append(maml.forward_deferred(*batches[k % a.n_batches][:4]))
            return pending.pop(0).accs() if len(pending) > 1 else None
        return maml(*batches[k % a.n_batches], data['feats'])

    def drain():
        out = None
        while pending:
            out = pending.pop(0).accs()
        return out

    lib = _lib.lib()

    def prof_read(cat=0):       # 0 aggregate (bytes), 1 grouped GEMM (flops), 2 weight gradient (flops), 3 aggregate, compulsory HBM bytes
        ms, n, by = C.c_double(), C.c_int64(), C.c_int64()
        lib.gm_profile_read(cat, C.byref(ms), C.byref(n), C.byref(by))
        return ms.value, n.value, by.value

    for k in range(a.warmup):
        step(k)
    drain()
    # Per-launch HIP events (roofline / MFMA utilisation) cost a few microseconds per launch, which is noise at the arxiv shape but
    # 30-50 % on the launch-latency-bound configs: the timed region runs WITHOUT them unless it is itself the roofline sample
    # (--serialize 1); otherwise they are switched on for the extra steps right after it.
    lib.gm_profile_enable(1 if a.serialize else 0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ov_ms, ov_n, ov_bytes = 0.0, 0, 0
    MM_CATS = (1, 2, 4, 5, 6, 7)      # exact-fp32 GEMM / weight gradient, split-bf16 (three pieces) ditto, split-fp16 (two pieces) ditto (gm_profile_read categories)
    ov_mm = {c: [0.0, 0, 0] for c in MM_CATS}
    strict_bytes = 0
    ov_gemm_bytes = 0
    ov_bound = 0
    for k in range(a.steps):
        accs = step(k)                 # returns after the one device->host read of losses/accs
        if a.defer or not a.serialize:
            continue                   # per-launch events are read in the extra steps below
        ms, n, by = prof_read()
        ov_ms += ms; ov_n += n; ov_bytes += by
        strict_bytes += prof_read(3)[2]
        ov_gemm_bytes += prof_read(11)[2]
        ov_bound += prof_read(12)[2]
        for cat in MM_CATS:
            ms, n, fl = prof_read(cat)
            ov_mm[cat][0] += ms; ov_mm[cat][1] += n; ov_mm[cat][2] += fl
    if a.defer:
        accs = drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device='cuda', dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    # ---- per-kernel roofline of the aggregate: in the timed region the support chain and the query evaluations share
    # the GPU on two streams, which stretches every kernel; the kernel-alone duration is measured on extra steps with
    # serialize=1 (same inputs, same launches, one stream) -- the rocprofv3 summary under profiles/ uses the same mode.
    agg_ms, agg_n, agg_bytes = ov_ms, ov_n, ov_bytes
    gemm_bytes = ov_gemm_bytes
    agg_launches = []
    bound_extra = ov_bound
    mm = ov_mm if a.serialize else {c: [0.0, 0, 0] for c in MM_CATS}    # [ms, launches, flops] of the GEMM / weight-gradient launches (serialised steps)
    ser_steps = a.steps if a.serialize else 0
    one_stream_ms = None
    deferred_ms = None
    if not a.serialize and not a.defer and a.roofline_steps > 0:
        # the same steps with the accuracies of step k read after step k + 1 is queued (Meta.forward_deferred -- what train.py does between its report
        # steps): the host prologue of a step then runs while the GPU still works on the previous one.  `value` keeps the reference's calling
        # convention (Meta.forward returns the accuracies of every step).
        nd = max(4, min(a.steps, 50))
        pend = [maml.forward_deferred(*batches[0][:4])]
        pend.pop(0).accs()
        torch.cuda.synchronize(); te = time.perf_counter()
        for k in range(nd):
            pend.append(maml.forward_deferred(*batches[k % a.n_batches][:4]))
            if len(pend) > 1:
                pend.pop(0).accs()
        while pend:
            pend.pop(0).accs()
        torch.cuda.synchronize()
        deferred_ms = (time.perf_counter() - te) / nd * 1e3
    if not a.serialize and a.roofline_steps > 0:            # every rank takes part: Meta.forward all-reduces when N > 1
        # the same step on ONE stream (no events): what the two queues buy -- kernels of the support chain and of the query evaluations side by side,
        # the stream aggregate's workgroups inside the CUs a persistent GEMM workgroup of the other queue occupies
        maml.serialize = 1
        step(0); drain()
        torch.cuda.synchronize(); te = time.perf_counter()
        for k in range(a.roofline_steps):
            step(k)
        drain(); torch.cuda.synchronize()
        one_stream_ms = (time.perf_counter() - te) / a.roofline_steps * 1e3
        maml.serialize = 0
        lib.gm_profile_enable(1)
        step(0); drain()                                    # one two-stream step with events: the aggregate's rate while it shares the GPU
        ov_ms, ov_n, ov_bytes = prof_read()
        maml.serialize = 1
        step(0); drain()
        agg_ms, agg_n, agg_bytes, strict_bytes, gemm_bytes, bound_extra = 0.0, 0, 0, 0, 0, 0
        for k in range(a.roofline_steps):
            step(k); drain()
            ms, n, by = prof_read()
            agg_ms += ms; agg_n += n; agg_bytes += by
            strict_bytes += prof_read(3)[2]
            for cat in MM_CATS:
                ms, n, fl = prof_read(cat)
                mm[cat][0] += ms; mm[cat][1] += n; mm[cat][2] += fl
            gemm_bytes += prof_read(11)[2]
            bound_extra += prof_read(12)[2]
            if k == a.roofline_steps - 1:           # per-launch view of the last serialised step: which launches carry the mix
                cap = 512
                ms_a, wk_a = (C.c_double * cap)(), (C.c_int64 * cap)()
                n_l = lib.gm_profile_read_launches(0, ms_a, wk_a, cap)
                agg_launches = [(ms_a[i], wk_a[i]) for i in range(max(n_l, 0))]
        maml.serialize = 0
        ser_steps = a.roofline_steps
    lib.gm_profile_enable(0)
    box = None
    if rank == 0 and not os.environ.get('GMETA_NO_BOX'):
        try:
            box = box_calibration(lib, _lib, batches[0][2][0].view_of)
        except Exception as e:         # a calibration aid, never the product path
            box = {'error': repr(e)}
    # ---- secondary numbers: the flagged schedules that produce identical results without the structural zeros /
    # loop-invariant recomputation (never the headline `value`)
    extra = {}
    if world == 1 and a.extra_steps > 0 and not (a.sparse_bwd or a.hoist_z1 or a.serialize or a.cone):
        _skip = set(os.environ.get('GMETA_BENCH_SKIP', '').split(','))
        for name, kw in (('sparse_bwd', dict(sparse_bwd=1)), ('sparse_bwd+hoist_z1', dict(sparse_bwd=1, hoist_z1=1)),
                         ('cone', dict(sparse_bwd=0, hoist_z1=0, cone=1)), ('cone+hoist_z1', dict(cone=1, hoist_z1=1))):
            if 'flagged' in _skip:
                break
            for k_, v_ in kw.items():
                setattr(maml, k_, v_)
            step(0); drain()
            torch.cuda.synchronize(); te = time.perf_counter()
            for k in range(a.extra_steps):
                step(k)
            drain()
            torch.cuda.synchronize()
            ms_e = (time.perf_counter() - te) / a.extra_steps * 1e3
            extra[name] = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1)}
            if kw.get('cone') and a.roofline_steps > 0:
                extra[name].update(schedule_roofline(lib, maml, step, drain, a.roofline_steps, ms_e))
            if name == 'cone+hoist_z1' and a.e2e_steps > 0:
                # the same schedule with a FRESH extraction (+ receptive-field tables) per step, built while the previous meta-steps run
                # (Subgraphs.batches): one host thread for the host halves, N builder threads (best of 1 / 2 / 4 reported; train.py --num_workers)
                base_l = [list(range((a.n_batches + k) * T, (a.n_batches + k + 1) * T)) for k in range(a.e2e_steps + 2)]
                n_e = max(2 * a.e2e_steps, 60)
                by_wk = {}
                for wk in tuple(int(x) for x in os.environ.get('GMETA_BENCH_BUILDERS', '1,2,4').split(',')):
                    it = iter(db.batches([base_l[k % len(base_l)] for k in range(n_e + len(base_l))], prefetch=wk + 1, cone_layers=cfg['h'], workers=wk))
                    for _ in range(len(base_l)):          # one pass over the distinct meta-batches first (their per-task host tables are memoised from then on)
                        maml(*next(it), data['feats'])
                    torch.cuda.synchronize(); te = time.perf_counter()
                    for _ in range(n_e):
                        maml(*next(it), data['feats'])
                    torch.cuda.synchronize()
                    ms_x = (time.perf_counter() - te) / n_e * 1e3
                    by_wk[wk] = {'ms_per_step': round(ms_x, 3), 'meta_tasks_per_s': round(T / (ms_x * 1e-3), 1), 'steps': n_e, 'builder_threads': wk}
                    del it
                best = min(by_wk, key=lambda k: by_wk[k]['ms_per_step'])
                extra[name]['end_to_end'] = dict(by_wk[best], by_builder_threads={str(k): v['ms_per_step'] for k, v in by_wk.items()},
                                                 what='Subgraphs.batches(prefetch=workers+1, workers=N): extraction + sampling + induced batches + receptive-field tables of '
                                                      'every meta-batch built on the GPU by N builder threads (own streams) while the previous meta-steps run; Meta.forward per '
                                                      'step.  Round 6: one builder keeps up with this 2.2 ms meta-step (a build is ~2.3 ms of host wall; dropped batches go to a slab '
                                                      'cache instead of hipFreeAsync, which cost 0.7-1.7 ms per meta-batch in the training thread); more builders contend for the GPU')
        maml.sparse_bwd = 0; maml.hoist_z1 = 0; maml.cone = 0
        if lib.gm_get_gemm_mode() == 1 and 'exact' not in _skip:      # the same dense schedule with every GEMM on the exact-fp32 MFMA kernels (include/gmeta_hip.h, gm_set_gemm_mode)
            lib.gm_set_gemm_mode(0)
            step(0); drain()
            torch.cuda.synchronize(); te = time.perf_counter()
            for k in range(a.extra_steps):
                step(k)
            drain()
            torch.cuda.synchronize()
            ms_e = (time.perf_counter() - te) / a.extra_steps * 1e3
            extra['exact_f32_mfma_gemm'] = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1),
                                            'what': 'default schedule with GM_GEMM_MODE=f32: every update GEMM on v_mfma_f32_32x32x2_f32'}
            lib.gm_set_gemm_mode(1)
            if lib.gm_get_split_pieces() == 3:      # ... and the OPT-IN fast mode: two fp16 pieces per operand (22 significand bits -- narrower than fp32; never the headline)
                lib.gm_set_split_pieces(2)
                step(0); drain()
                torch.cuda.synchronize(); te = time.perf_counter()
                for k in range(a.extra_steps):
                    step(k)
                drain()
                torch.cuda.synchronize()
                ms_e = (time.perf_counter() - te) / a.extra_steps * 1e3
                extra['split_fp16_two_pieces'] = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1),
                                                  'dtype': 'f32 storage, 2xfp16 operands (22-bit)',
                                                  'what': 'default schedule with GM_SPLIT_PIECES=2 (opt-in): the split launches inside gm_meta_step take two fp16 pieces per operand, '
                                                          'three products -- operands NARROWER than the reference\'s fp32 (learner.py:36,47), reported for context only'}
                lib.gm_set_split_pieces(-1)
        if lib.gm_get_gemm_mode() == 1 and 'store' not in _skip:
            # the same schedule with the forward-only evaluations storing EVERY row of the last layer's activation (only the centre rows are ever read)
            _lib.check(lib.gm_set_tuning(b'GM_CENTRE_STORE', 0), 'set_tuning')
            step(0); drain()
            torch.cuda.synchronize(); te = time.perf_counter()
            for k in range(a.extra_steps):
                step(k)
            drain()
            torch.cuda.synchronize()
            ms_e = (time.perf_counter() - te) / a.extra_steps * 1e3
            extra['store_every_row'] = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1),
                                        'what': 'default schedule with GM_CENTRE_STORE=0: the last GraphConv also stores the rows of its activation that nobody reads '
                                                '(bitwise the same step; the default computes every row and stores the centre rows the head gathers, learner.py h[to_fetch])'}
            _lib.check(lib.gm_set_tuning(b'GM_CENTRE_STORE', 2), 'set_tuning')
        if lib.gm_get_fuse_agg() == 1 and lib.gm_get_gemm_mode() == 1 and 'unfused' not in _skip:
            # the same schedule with every pass writing Z (fused aggregate + GEMM off): step time, and the aggregate's roofline over an
            # all-full-launch sample -- the figure of the earlier rounds (the large query launches are aggregate launches again)
            lib.gm_set_fuse_agg(0)
            step(0); drain()
            torch.cuda.synchronize(); te = time.perf_counter()
            for k in range(a.extra_steps):
                step(k)
            drain()
            torch.cuda.synchronize()
            ms_e = (time.perf_counter() - te) / a.extra_steps * 1e3
            extra['unfused_aggregate'] = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1),
                                          'what': 'default schedule with GM_FUSE_AGG=0: every forward pass writes Z_l and runs the plain GEMM'}
            if a.roofline_steps > 0:
                lib.gm_profile_enable(1)
                maml.serialize = 1
                step(0); drain()
                u_ms, u_n, u_by = 0.0, 0, 0
                for k in range(a.roofline_steps):
                    step(k); drain()
                    ms, n, by = prof_read()
                    u_ms += ms; u_n += n; u_by += by
                maml.serialize = 0
                lib.gm_profile_enable(0)
                if u_ms > 0:
                    u_ach = u_by / (u_ms * 1e-3) / 1e9
                    extra['unfused_aggregate']['roofline'] = {'kernel': 'k_agg, full launches only', 'achieved': round(u_ach, 1), 'unit': 'GB/s',
                                                              'frac': round(u_ach / HBM_PEAK_GBS, 4), 'launches_measured': u_n,
                                                              'avg_launch_ms': round(u_ms / max(u_n, 1), 4),
                                                              'algorithmic_bytes_per_launch': u_by // max(u_n, 1)}
            lib.gm_set_fuse_agg(-1)
        lv = {}
        for side, x in (('spt', batches[0][0][0].view_of), ('qry', batches[0][2][0].view_of)):      # what the cone schedule touches
            ok = C.c_int32(); nr = (C.c_int64 * (cfg['h'] + 1))(); ne = (C.c_int64 * (cfg['h'] + 1))()
            _lib.check(lib.gm_batch_cone_dims(x.handle, cfg['h'], C.byref(ok), nr, ne), 'cone_dims')
            lv[side] = {'batch_rows': int(x.rows), 'batch_edges': int(x.edges), 'level_rows': list(nr), 'level_edges': list(ne)}
        extra['cone']['receptive_field'] = lv
    # ---- evaluation leg (BASELINE configs[3]: "x10 eval tasks"): Meta.finetunning over eval_tasks tasks in one batched call
    if n_eval > 0 and not (a.sparse_bwd or a.hoist_z1 or a.serialize or a.cone):
        base = T * (a.n_batches + a.e2e_steps + 2)
        ev = db.get_batch(list(range(base, base + n_eval)))
        maml.finetunning_batch(ev[0], ev[1], ev[2], ev[3])
        torch.cuda.synchronize(); te = time.perf_counter()
        reps = 3
        for _ in range(reps):
            acc_e = maml.finetunning_batch(ev[0], ev[1], ev[2], ev[3])
        torch.cuda.synchronize()
        ms_e = (time.perf_counter() - te) / reps * 1e3
        extra['finetunning'] = {'tasks': n_eval, 'update_step_test': cfg['update_step_test'], 'ms_per_call': round(ms_e, 3),
                                'tasks_per_s': round(n_eval / (ms_e * 1e-3), 1), 'mean_final_acc': round(float(acc_e[:, -1].mean()), 4),
                                'what': 'Meta.finetunning (meta.py:175-234) for all eval tasks in one batched call (the reference loops them, train.py:118-121)'}

    # ---- end to end ("report also with extraction included"): every step extracts its own meta-batch
    e2e = None
    if world == 1 and a.e2e_steps > 0:
        # e2e_steps + 2 distinct meta-batches, walked n_e times in a row like the epochs of train.py (the task tables are fixed at construction,
        # sdp.py:150-292; their per-task host arrays are memoised after the first visit, as the reference memoises its subgraphs)
        lists = [list(range((a.n_batches + k) * T, (a.n_batches + k + 1) * T)) for k in range(a.e2e_steps + 2)]
        n_e = max(3 * a.e2e_steps, 30)
        if os.environ.get('GMETA_BENCH_GC_FREEZE'):
            import gc
            gc.collect(); gc.freeze()
        it = iter(db.batches([lists[k % len(lists)] for k in range(len(lists) + n_e)], prefetch=1, cone_layers=cfg['h'] if a.cone else 0))
        for _ in range(len(lists)):
            maml(*next(it), data['feats'])
        torch.cuda.synchronize(); te = time.perf_counter()
        for _ in range(n_e):
            maml(*next(it), data['feats'])
        torch.cuda.synchronize()
        ms_e = (time.perf_counter() - te) / n_e * 1e3
        del it
        # the pre-extracted step again, right here: by now the chip has run several seconds of phases back to back (the timed region started on a
        # cooler one; profiles/r03_gemm_hot_vs_cool.txt), so the cost of building beside the step is ms_per_step - this, not ms_per_step - `ms_per_step` of the line
        step(0); drain()
        torch.cuda.synchronize(); te = time.perf_counter()
        for k in range(n_e):
            step(k)
        drain(); torch.cuda.synchronize()
        ms_same = (time.perf_counter() - te) / n_e * 1e3
        e2e = {'ms_per_step': round(ms_e, 3), 'meta_tasks_per_s': round(T / (ms_e * 1e-3), 1), 'steps': n_e, 'pre_extracted_ms_per_step_measured_right_after': round(ms_same, 3),
               'what': 'Subgraphs.batches (h-hop extraction + sampling + induced batches of every meta-batch built on the GPU one step ahead, on a builder '
                       'thread / stream of its own) + Meta.forward per step; same schedule as `value`; the meta-batches of the second and later epochs '
                       '(per-task host tables memoised)'}
    # ---- extraction kernels against the HBM roofline (SURVEY 8(d) "k-hop expansion, induce: HBM bandwidth"): one more extraction of the first
    # meta-batch with HIP events around k_nodes / k_fill / the finalisation (gm_profile_read categories 8-10), bytes from the actual batch
    extraction = None
    if world == 1 and not (a.serialize or a.cone or a.sparse_bwd or a.hoist_z1):
        idx0 = list(range(lo, hi))
        torch.cuda.synchronize()
        reps = 3
        t_ex = time.perf_counter()
        for _ in range(reps):
            bx = db.get_batch(idx0)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t_ex) / reps * 1e3
        # the kernels' HIP events (default build: both batches in one gm_extract_pair call on this thread), then the wall time of two separate gm_extract calls
        lib.gm_profile_enable(1)
        try:
            for _ in range(reps):
                bx = db.get_batch(idx0)
            torch.cuda.synchronize()
            ex = [prof_read(c) for c in (8, 9, 10)]
        finally:
            lib.gm_profile_enable(0)
        os.environ['GMETA_EXTRACT_MODE'] = 'serial'
        try:
            t_ex = time.perf_counter()
            for _ in range(reps):
                bx = db.get_batch(idx0)
            torch.cuda.synchronize()
            wall1_ms = (time.perf_counter() - t_ex) / reps * 1e3
        finally:
            del os.environ['GMETA_EXTRACT_MODE']
        by = extraction_bytes(store, db, idx0, bx, cfg['h'], link)
        k_ms = (ex[0][0] + ex[1][0]) / reps
        tot = by['expand'] + by['induce'] + by['write']
        extraction = {'bound': 'hbm', 'kernels': 'k_nodes (h-hop expansion + sampling + node lists + induced degrees) + k_fill (batched CSR, both orientations); one launch of each over the support AND the query subgraphs (gm_extract_pair)',
                      'algorithmic_bytes_per_meta_batch': tot, 'bytes': by, 'k_nodes_ms': round(ex[0][0] / reps, 4), 'k_fill_ms': round(ex[1][0] / reps, 4),
                      'finalize_span_ms': round(ex[2][0] / reps, 4), 'launches_per_meta_batch': ex[0][1] // reps,
                      'subgraphs': int(sum(x.subs for x in (bx[0][0].view_of, bx[2][0].view_of))),
                      'achieved': round(tot / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': round(tot / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
                      'host_wall_ms_per_meta_batch': round(wall_ms, 3), 'host_wall_ms_two_calls': round(wall1_ms, 3),
                      'symmetric_parent_fast_path': bool(store.symmetric()),
                      'measured': 'HIP events around the kernels on the extraction stream, %d extractions of the first meta-batch after the timed region' % reps,
                      'note': 'latency-bound integer work: one workgroup per subgraph, adjacency lists walked through dependent loads; the induced '
                              'subgraphs are built in two phases (count, then fill) so the lists are walked twice against the one pass priced here'}
        del bx
    # ---- the one collective of a sharded meta-step, timed on its own (same buffer size, same stream): what the step's value already contains
    allreduce = None
    rank_report = None
    if use_dist:
        # what every rank of the communicator reports about itself: [rank, device, tasks of its shard, rows of its shard] over RCCL
        mine = torch.tensor([rank, torch.cuda.current_device(), hi - lo, int(rows)], dtype=torch.int64, device='cuda')
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rank_report = {'ranks_reporting': [int(g[0]) for g in got], 'devices': [int(g[1]) for g in got], 'tasks_per_rank': [int(g[2]) for g in got],
                       'rows_per_rank': [int(g[3]) for g in got], 'backend': dist.get_backend()}
        P_ = sum(p.numel() for p in maml.net.parameters())
        buf = torch.zeros(P_ + 2 * (cfg['update_step'] + 1) + 1, dtype=torch.float32, device='cuda')
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        ev0.record()
        for _ in range(reps):
            dist.all_reduce(buf)
        ev1.record(); torch.cuda.synchronize()
        allreduce = {'us_per_call': round(ev0.elapsed_time(ev1) / reps * 1e3, 2), 'bytes': int(buf.numel() * 4), 'ranks': world,
                     'what': 'torch.distributed.all_reduce(SUM) of [grad | losses_q | corrects | count] over RCCL, %d back-to-back calls between two events' % reps}
    if use_dist:                           # tear the communicator down before printing: nothing follows the JSON line
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ach = agg_bytes / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else None
        strict = strict_bytes / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else None
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE): NOT measured in this run; read from the
        # committed summary of a separate rocprofv3 --pmc pass over this command (see profiles/), and only quoted for the workload it was taken on
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, 'profiles', 'agg_traffic.json')
        if os.path.exists(tp) and a.config == 'arxiv' and not a.task_num and world == 1:
            try:
                tj = json.load(open(tp))
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_source = 'profiles/agg_traffic.json (%s; separate rocprofv3 --pmc pass, not this run)' % tj.get('taken', 'date unknown')
            except Exception:
                traffic = None
        two_piece = lib.gm_get_gemm_mode() == 1 and lib.gm_get_split_pieces() == 2 and not (a.cone or a.sparse_bwd)
        gemm_mode = (('split-fp16: large N=128/256 launches on v_mfma_f32_32x32x16_f16 with every fp32 operand split into two fp16 pieces (22 significand bits) '
                      'under per-task power-of-two scales from magnitude bounds recorded on the device, three products, fp32 accumulation (error vs fp64 below the '
                      'fp32 fmaf chain\'s, tests/test_hip_gemm_numerics.py; DESIGN.md section 4); launches without bounds: split-bf16, three exact pieces / six '
                      'products; other launches exact fp32' if two_piece else
                      'split-bf16: large N=128/256 launches on v_mfma_f32_32x32x16_bf16 with every fp32 operand split exactly into three bf16 pieces, six '
                      'products, fp32 accumulation (error vs fp64 <= the fp32 fmaf chain\'s; DESIGN.md section 4); other launches exact fp32')
                     if lib.gm_get_gemm_mode() == 1 else 'exact fp32 (v_mfma_f32_32x32x2_f32) everywhere')
        fused = lib.gm_get_fuse_agg() == 1 and lib.gm_get_gemm_mode() == 1 and not (a.cone or a.hoist_z1)
        if fused:                           # the library fuses only where at least half of the query batch's rows have one or two sources (model.hip)
            try:
                ip = batches[0][2][0].view_of.csr()[0]
                deg = np.diff(np.asarray(ip, np.int64))
                fused = 2 * int((deg > 2).sum()) <= len(deg)
            except Exception:
                pass
        out = {
            'metric': 'meta-tasks/sec (inner-loop fwd+bwd) at task_num=%d' % T, 'value': round(T / (ms_per_step * 1e-3), 3),
            'unit': 'meta-tasks/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32 storage, 2xfp16 operands (22-bit)' if two_piece else 'f32', 'data': 'synthetic',
            'arithmetic': 'fp32 storage and accumulation throughout; update GEMMs: ' + gemm_mode,
            'config': {'workload': synth.WORKLOADS[a.config] +
                                   ' (synthetic %s, F0=%d), %s%s, h=%d, hidden=%d, %d-way %d-shot %d-qry, task_num=%d, update_step=%d, '
                                   'sample_nodes=%d; subgraphs pre-extracted in HBM'
                                   % ('PA graph N=%d m=%d' % (cfg['n'], cfg['m']) if cfg.get('kind', 'single') == 'single' else
                                      '%d graphs x %d nodes (PA m=%d)' % (cfg['n_graphs'], cfg['n'], cfg['m']), cfg['F0'],
                                      cfg.get('task_setup', 'Disjoint'), ' link prediction' if link else '', cfg['h'], cfg['hidden'], cfg['n_way'],
                                      cfg['k_spt'], cfg['k_qry'], T, cfg['update_step'], cfg['sample_nodes']),
                       'schedule': ('hoist_z1 ' if a.hoist_z1 else '') + ('cone (receptive-field rows only, forward and backward)' if a.cone else
                                    'sparse_bwd (dense forward, exact row-sparse backward)' if a.sparse_bwd else
                                    'full (reference-equivalent: every forward/backward dense over all subgraph rows)'),
                       'fused_aggregate_gemm': bool(fused),
                       'last_layer_stores': 'every row of the last GraphConv is computed (and its relu bits written where a backward pass follows); the activation is stored at the centre rows the head gathers '
                                            '(extra.store_every_row: the same step with GM_CENTRE_STORE=0)',
                       'streams': 1 if a.serialize else 2,
                       'readback': 'deferred by one step (Meta.forward_deferred)' if a.defer else 'every step (Meta.forward returns the accuracies)',
                       'parallelism': 'tasks sharded over %d rank(s) (rank 0: %d of %d), one all-reduce of the meta-gradient per step' % (world, hi - lo, T),
                       'rows_per_rank': int(rows), 'edges_per_rank': int(edges),
                       'extract_ms_per_meta_batch_rank0': round(float(np.min(ext_ms)), 2), 'last_accs': [round(float(x), 4) for x in accs]},
            'roofline': {'bound': 'hbm', 'kernel': 'k_agg_win + k_agg_stream (batched subgraph message passing, all widths: the full forward launches of large sparse batches '
                                                     'take the LDS-DMA stream kernel, every other launch the window kernel)',
                         'achieved': round(ach, 1) if ach else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None, 'traffic': traffic, 'traffic_source': traffic_source,
                         'strict_hbm_achieved': round(strict, 1) if strict else None,
                         'strict_hbm_frac': round(strict / HBM_PEAK_GBS, 4) if strict else None,
                         'strict_hbm_note': 'same launches and durations, layer-1 launches (which gather rows of the L2/MALL-resident feature table) '
                                            'priced at the table size instead of rows x width',
                         'launches_measured': agg_n, 'avg_launch_ms': round(agg_ms / max(agg_n, 1), 4),
                         'algorithmic_bytes_per_launch': agg_bytes // max(agg_n, 1),
                         'measured': 'HIP events on the launch stream over %s' % ('the timed region (serialize=1)' if a.serialize else
                                     '%d serialised steps run right after the timed region' % a.roofline_steps),
                         'achieved_while_overlapped': round(ov_bytes / (ov_ms * 1e-3) / 1e9, 1) if ov_ms > 0 else None,
                         'by_launch_size': launch_classes(agg_launches),
                         'frac_with_round3_pricing': round((agg_bytes + bound_extra) / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if agg_ms > 0 else None,
                         'round3_pricing_note': 'rounds 2-3 priced the sources of a partial launch as min(edges, rows) rows (an upper bound); `frac` counts the DISTINCT source rows',
                         'launch_mix': ('full launches (support chain, the differentiated query pass) AND the partial launches of the forward-only query passes, '
                                        'whose 0..2-source rows are aggregated inside the fused aggregate+GEMM kernel: those launches are priced with B_agg '
                                        'restricted to what they touch (every indptr entry; indices, norms and output rows of the >=3-source rows; their '
                                        'DISTINCT source rows read once -- counted on the device per batch).  GM_FUSE_AGG=0 gives the all-full-launch sample of the earlier rounds.')
                                       if fused else 'full launches only (fused aggregate+GEMM off or not applicable to this schedule)'},
        }
        if mm[1][0] + mm[4][0] + mm[6][0] > 0:
            # Every launch is priced on the matrix pipe it ran on: the exact-fp32 kernels (v_mfma_f32_32x32x2_f32) against the 157.3 TFLOP/s dense
            # fp32 matrix peak, the split-bf16 kernels -- which issue SIX bf16 MFMA flops per flop of the fp32 product -- against 2.5 PFLOP/s bf16.
            def pipe(exact, split, split16):
                ms = mm[exact][0] + mm[split][0] + mm[split16][0]; fl = mm[exact][2] + mm[split][2] + mm[split16][2]
                t_pk = (mm[exact][2] / (MFMA_F32_PEAK_TFLOPS * 1e12) + 6.0 * mm[split][2] / (MFMA_BF16_PEAK_TFLOPS * 1e12) +
                        3.0 * mm[split16][2] / (MFMA_BF16_PEAK_TFLOPS * 1e12)) * 1e3       # ms at the peak of the pipe used (fp16 and bf16 MFMA: the same dense peak)
                d = {'fp32_equivalent_tflops': round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None, 'launches': mm[exact][1] + mm[split][1] + mm[split16][1],
                     'frac': round(t_pk / ms, 4) if ms > 0 else None}
                if mm[split16][0] > 0:
                    tf6 = mm[split16][2] / (mm[split16][0] * 1e-3) / 1e12
                    d['split_fp16'] = {'launches': mm[split16][1], 'fp32_equivalent_tflops': round(tf6, 1), 'fp16_mfma_tflops': round(3 * tf6, 1),
                                       'frac_of_fp16_peak': round(3 * tf6 / MFMA_BF16_PEAK_TFLOPS, 4), 'ms_per_step': round(mm[split16][0] / max(ser_steps, 1), 3),
                                       'note': 'memory-bound at three products per fp32 product: see DESIGN.md section 4 for the bytes these launches move'}
                if mm[split][0] > 0:
                    tfs = mm[split][2] / (mm[split][0] * 1e-3) / 1e12
                    d['split_bf16'] = {'launches': mm[split][1], 'fp32_equivalent_tflops': round(tfs, 1), 'bf16_mfma_tflops': round(6 * tfs, 1),
                                       'frac_of_bf16_peak': round(6 * tfs / MFMA_BF16_PEAK_TFLOPS, 4), 'ms_per_step': round(mm[split][0] / max(ser_steps, 1), 3)}
                if mm[exact][0] > 0:
                    tfe = mm[exact][2] / (mm[exact][0] * 1e-3) / 1e12
                    d['exact_f32'] = {'launches': mm[exact][1], 'tflops': round(tfe, 1), 'frac_of_f32_peak': round(tfe / MFMA_F32_PEAK_TFLOPS, 4),
                                      'ms_per_step': round(mm[exact][0] / max(ser_steps, 1), 3)}
                return d, t_pk
            g_d, g_pk = pipe(1, 4, 6); w_d, w_pk = pipe(2, 5, 7)
            sp_ms = mm[4][0] + mm[6][0]
            if sp_ms > 0 and gemm_bytes > 0:
                # the split update GEMMs against the HBM roofline: with the matrix work on the bf16 / fp16 pipes they stream A and C (the weight planes
                # come from L2): compulsory bytes 4 rows (K + N) per launch / HIP-event time
                gb = gemm_bytes / (sp_ms * 1e-3) / 1e9
                g_d['hbm_roofline'] = {'bound': 'hbm', 'kernel': 'k_gemm_split_p (split update GEMMs, plain and fused aggregate+GEMM launches)', 'achieved': round(gb, 1),
                                       'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gb / HBM_PEAK_GBS, 4), 'launches': mm[4][1] + mm[6][1],
                                       'compulsory_bytes_per_step': gemm_bytes // max(ser_steps, 1), 'ms_per_step': round(sp_ms / max(ser_steps, 1), 3),
                                       'what': 'A operand read once + C written once, 4 rows (K + N) bytes per launch (fused launches: the aggregate rows they form count as their A operand)'}
            g_d['what'] = 'forward X@W and backward dZ = dQ@W^T'; w_d['what'] = 'dW = (norm*Z)^T dQ, db, incl. the partial reduction'
            out['mfma'] = {'note': 'update GEMMs, HIP events around every launch of the same serialised steps as the roofline; flops counted as 2*rows*K*N of the '
                                   'fp32 product; `frac` = time those launches would take at the dense peak of the pipe each one ran on (fp32 MFMA 157.3 TFLOP/s for '
                                   'the exact kernels; bf16 / fp16 MFMA 2.5 PFLOP/s at 6 bf16 (three-piece kernels) or 3 fp16 (two-piece kernels) flops per fp32 flop for the split kernels) / measured time',
                           'peak_tflops': {'f32': MFMA_F32_PEAK_TFLOPS, 'bf16': MFMA_BF16_PEAK_TFLOPS}, 'gemm_mode': gemm_mode, 'gemm': g_d, 'wgrad': w_d}
            if ser_steps > 0:
                # composite bound of the whole step: every update launch at the peak of its own pipe + all aggregate bytes at the HBM peak
                fl = sum(mm[c][2] for c in MM_CATS) / ser_steps; by = agg_bytes / ser_steps
                t_mfma, t_hbm = (g_pk + w_pk) / ser_steps, by / (HBM_PEAK_GBS * 1e9) * 1e3
                t_f32 = fl / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3
                out['step_bound'] = {'update_gflop_per_step': round(fl / 1e9, 1), 'aggregate_gb_per_step': round(by / 1e9, 2),
                                     'ms_at_mfma_peak_of_pipe_used': round(t_mfma, 2), 'ms_at_hbm_peak': round(t_hbm, 2),
                                     'frac_of_serial_bound': round((t_mfma + t_hbm) / ms_per_step, 3),
                                     'frac_of_overlapped_bound': round(max(t_mfma, t_hbm) / ms_per_step, 3),
                                     'labelled_extra_ms_if_all_flops_ran_at_fp32_mfma_peak': round(t_f32, 2),
                                     'note': 'split launches priced at 6 (bf16, three pieces) or 3 (fp16, two pieces) MFMA flops per fp32 flop on the 2.5 PFLOP/s pipe, exact-fp32 launches at 157.3 TFLOP/s; '
                                             'the last field is the bound an all-exact-fp32 implementation would have (context only)'}
        if one_stream_ms:
            out['two_queues'] = {'ms_per_step_two_streams': round(ms_per_step, 3), 'ms_per_step_one_stream': round(one_stream_ms, 3),
                                 'gain': round(1.0 - ms_per_step / one_stream_ms, 4), 'stream_aggregate': bool(lib.gm_get_tuning(b'GM_AGG_STREAM')),
                                 'what': 'the same step with the query evaluations on the support chain\'s stream; the difference is what side-by-side execution buys. '
                                         'A persistent split-GEMM workgroup owns 480 of the 512 VGPRs of each SIMD lane and 101 KiB of LDS: only a kernel of <= 32 VGPRs '
                                         'and <= 59 KiB starts beside it (profiles/r05_coreside_micro.log) -- the stream aggregate (30 VGPRs, 52 KiB) is built to that budget'}
        if deferred_ms:
            out['deferred_readback'] = {'ms_per_step': round(deferred_ms, 3), 'meta_tasks_per_s': round(cfg['task_num'] / deferred_ms * 1e3, 1),
                                        'what': 'Meta.forward_deferred: the accuracies of step k are read after step k + 1 is queued (train.py between report steps); every step does '
                                                'all of its work, the host prologue of a step overlaps the previous step on the GPU.  Not `value`: the reference returns the accuracies every step'}
        if box:
            out['box'] = box
        if allreduce:
            out['allreduce'] = allreduce
        if rank_report:
            out['ranks'] = rank_report
        if extraction:
            out['extraction'] = extraction
        if e2e:
            out['end_to_end'] = e2e
        if extra:
            out['extra'] = {'note': 'flagged exact schedules (same accs/meta-gradient, golden-tested) and the evaluation leg; not the headline value', **extra}
        if not a.no_cpu_baseline and world == 1:           # rank 0 at N=1 only
            try:
                out['cpu_baseline'] = cpu_baseline(db, data, cfg, config, batches[0])
            except Exception as e:   # the baseline is a reported number, never the product path
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        try:                               # RCCL / HIP runtime banners sit in the C stdio buffer: flush them first so that the JSON is the LAST line
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
