"""GPU (-m gpu): the HIP path at BASELINE.json's full sizes (arxiv shape: 169,343-node graph, F0=128, h=2, hidden 256,
3-way 3-shot 24-query, sample_nodes=1000), checked through size-independent properties (the floats of the same
shapes are compared with the oracle in test_hip_fullsize_oracle.py, the headline T=32 / K=10 configuration included):
extraction invariants + CSR transpose consistency (bit-exact integer work), aggregate linearity / degree identity /
adjointness <A x, y> = <x, A^T y>, batched == per-task, stream-mode and run-to-run determinism, hoisted == full schedule."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 8


@pytest.fixture(scope='module')
def world():
    import random
    import gmeta_amd
    from gmeta_amd import synth
    np.random.seed(222); random.seed(222); torch.manual_seed(222)
    args, cfg = synth.make_args('arxiv', task_num=T)
    data = synth.node_dataset(cfg['n'], cfg['m'], cfg['F0'], cfg['classes'])
    store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
    db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T, args=args, adjs=store, h=2,
                             tables={'train': (data['names'], data['labels'])}, verbose=False)
    batch = db.get_batch(list(range(T)))
    return dict(args=args, cfg=cfg, data=data, store=store, db=db, batch=batch)


def test_extraction_invariants_full_size(world):
    """a1/a3 at full size: ascending ids, centre present, size cap, every node within 2 hops (C BFS restatement of
    sdp.py:300-311), CSR rows/cols inside their subgraph, by-source CSR == exact transpose of the by-destination CSR."""
    import gmeta_oracle as orc
    n, src, dst = world['data']['graphs'][0]
    G = orc.Graph(n, src, dst)
    lib = orc._load_c()
    assert lib, 'oracle C kernels not built'
    seen = np.zeros(n, np.uint8); fa = np.zeros(n, np.int32); fb = np.zeros(n, np.int32)
    n_sampled = 0
    for side in (0, 2):
        B = world['batch'][side][0].view_of
        par, sub = B.parent(), B.sub_off
        ip, ix = B.csr(); tp, tx = B.csr(transposed=True)
        cen = B._read(8, B.subs, np.int32)
        seeds = np.concatenate([world['db']._seeds(world['db']._task_names(t)[0 if side == 0 else 1]) for t in range(T)])
        assert ip[0] == 0 and ip[-1] == B.edges and tp[-1] == B.edges and np.all(np.diff(ip) >= 0)
        rows = np.arange(B.rows)
        owner = np.searchsorted(sub, rows, side='right') - 1
        dstrow = np.repeat(rows, np.diff(ip))
        assert np.array_equal(owner[ix], owner[dstrow])                       # edges never leave their subgraph
        # transpose consistency: sort (src,dst) pairs both ways
        a = np.stack([ix, dstrow], 1); b = np.stack([np.repeat(rows, np.diff(tp)), tx], 1)
        assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b)           # by-source lists are (src asc, dst asc)
        for k in range(0, B.subs, 7):                                         # every 7th subgraph against the BFS restatement
            nodes = par[sub[k]:sub[k + 1]]
            i = int(seeds[k, 1])
            assert np.all(np.diff(nodes) > 0) and nodes[cen[k]] == i
            cnt = lib.oracle_khop_mark(C.c_int64(n), G.indptr.ctypes.data_as(C.c_void_p), G.indices.ctypes.data_as(C.c_void_p), C.c_int64(i), 2,
                                       seen.ctypes.data_as(C.c_void_p), fa.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p))
            assert seen[nodes].all()
            if cnt > 1000:
                n_sampled += 1
                assert len(nodes) in (1000, 1001)
                ours = orc.sample_nodes(np.nonzero(seen)[0].astype(np.int32), 1000, 222, 0, i)
                assert np.array_equal(ours, nodes)                             # the keyed sampler, bit-exact at full size
            else:
                assert len(nodes) == cnt and np.array_equal(np.nonzero(seen)[0], nodes)
            # induced edges of this subgraph == every parent edge with both ends inside
            lo, hi = sub[k], sub[k + 1]
            e_hip = ip[hi] - ip[lo]
            inside = np.zeros(n, bool); inside[nodes] = True
            e_ref = sum(int(inside[G.preds(v)].sum()) for v in nodes[:: max(1, len(nodes) // 50)])
            e_hip_s = sum(int(ip[lo + r + 1] - ip[lo + r]) for r in range(0, len(nodes), max(1, len(nodes) // 50)))
            assert e_ref == e_hip_s and e_hip >= 0
    assert n_sampled > 0


@pytest.mark.parametrize('width', [256, 128])
def test_aggregate_properties_full_size(world, width):
    from gmeta_amd import _lib
    lib = _lib.lib()
    Q = world['batch'][2][0].view_of
    n = Q.rows
    g = torch.Generator(device='cuda').manual_seed(width)
    x = torch.randn(n, width, device='cuda', generator=g); y = torch.randn(n, width, device='cuda', generator=g)

    def agg(v, transposed=0, s_in=None, s_out=None):
        out = torch.empty_like(v)
        _lib.check(lib.gm_aggregate(Q.handle, transposed, 0, _lib.ptr(v), width, _lib.ptr(s_in), _lib.ptr(s_out), _lib.ptr(out), _lib.stream_ptr()))
        return out
    ax, ay = agg(x), agg(y)
    # degree identity: A * ones = in-degree (integer-valued sums are exact in fp32 here)
    deg = torch.from_numpy(np.diff(Q.csr()[0]).astype(np.float32)).cuda()
    assert torch.equal(agg(torch.ones(n, width, device='cuda'))[:, 0], deg)
    # linearity
    lin = agg(2.0 * x - 0.5 * y)
    assert torch.allclose(lin, 2.0 * ax - 0.5 * ay, atol=2e-4, rtol=1e-5)
    # adjointness of the two CSR orientations: <A x, y> == <x, A^T y>   (fp64 accumulation of the inner products)
    lhs = (ax.double() * y.double()).sum().item(); rhs = (x.double() * agg(y, 1).double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))
    # scalings commute with the sum:  s_out * A (s_in * x)
    si = torch.rand(n, device='cuda', generator=g) + 0.5; so = torch.rand(n, device='cuda', generator=g) + 0.5
    assert torch.allclose(agg(x, 0, si, so), so[:, None] * agg(si[:, None] * x), atol=2e-4, rtol=1e-5)
    # run-to-run determinism (bitwise)
    assert torch.equal(agg(x), ax)
    # hub rows are split over several blocks whose partial rows meet in a scratch buffer that every launch reuses: alternate the
    # inputs and check the hub rows against an fp64 gather each time (a stale partial row would be off by O(1))
    indptr, indices = Q.csr()[:2]
    deg_i = np.diff(indptr)
    hubs = np.nonzero(deg_i > 256)[0][:64]
    assert len(hubs) > 0, 'no hub row with more than one part in this batch'
    idx = torch.from_numpy(np.asarray(indices)).cuda().long()
    for rep in range(6):
        v = x if rep % 2 == 0 else y
        got = agg(v)
        for r in hubs[:: max(1, len(hubs) // 8)]:
            want = v[idx[int(indptr[r]):int(indptr[r + 1])]].double().sum(0)
            assert torch.allclose(got[r].double(), want, atol=1e-3, rtol=1e-5), (rep, int(r))


def _meta(world, **kw):
    import argparse
    import gmeta_amd
    from gmeta_amd import synth
    a = argparse.Namespace(**vars(world['args']))
    for k, v in kw.items():
        setattr(a, k, v)
    a.update_step = 3
    torch.manual_seed(222)
    cfg = world['cfg']
    return gmeta_amd.Meta(a, synth.make_config(cfg['F0'], cfg['hidden'], cfg['h'], cfg['n_way'])).to('cuda')


def _step(m, batch):
    grads = {}
    orig = m.meta_optim.step
    m.meta_optim.step = lambda *a, **k: grads.setdefault('g', torch.cat([p.grad.reshape(-1) for p in m.net.parameters()]).clone())
    accs = m(*batch, None)
    m.meta_optim.step = orig
    return accs, grads['g']


def test_meta_step_determinism_and_schedule_equivalence(world):
    b = world['batch']
    a0, g0 = _step(_meta(world), b)
    a1, g1 = _step(_meta(world), b)
    assert np.array_equal(a0, a1) and torch.equal(g0, g1)                      # run-to-run bitwise determinism
    a2, g2 = _step(_meta(world, serialize=1), b)
    assert np.array_equal(a0, a2) and torch.equal(g0, g2)                      # two streams == one stream, bitwise
    # Hoisting the layer-1 aggregate changes nothing: bitwise when both schedules form Z_1 with the same kernel.  With the defaults (round 6) the dense
    # schedule's passes form it in the fused GEMM feeders + a partial WINDOW launch, the hoisted Z_1 of the 286 k-row query batch is one full launch of
    # the STREAM kernel, which sums a hub row's parts in another association: equal to rounding there, bitwise with the stream kernel off.
    a3, g3 = _step(_meta(world, hoist_z1=1), b)
    np.testing.assert_allclose(a3, a0, atol=1e-6)
    assert float((g3 - g0).abs().max()) <= 1e-5
    from gmeta_amd import _lib
    lib = _lib.lib()
    _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 0), 'set_tuning')
    try:
        a4, g4 = _step(_meta(world), b)
        a5, g5 = _step(_meta(world, hoist_z1=1), b)
    finally:
        _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 1), 'set_tuning')
    assert np.array_equal(a4, a5) and torch.equal(g4, g5)
    assert np.isfinite(a0).all() and torch.isfinite(g0).all() and float(g0.abs().max()) > 0


@pytest.mark.parametrize('gemm_mode', [0, 1])
def test_cone_schedule_equals_full_schedule(world, gemm_mode):
    """gm_hparams_t.cone: layer l only on the rows (L-l) in-hops upstream of a centre.  Same accuracies, same
    meta-gradient up to summation order; deterministic; composes with the layer-1 hoist bit for bit.
    gemm_mode 0: every GEMM on the exact-fp32 MFMA kernels -> the two schedules differ by summation order only (2e-6).
    gemm_mode 1 (default): the dense schedule's large launches run on the split-bf16 kernel (exact 3-way operand split,
    gemm_split.h) while the compact cone matrices stay on the fp32 kernels -> agreement to fp32 rounding (2e-5; the north-star
    tolerance is 1e-4)."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    prev = lib.gm_get_gemm_mode()
    lib.gm_set_gemm_mode(gemm_mode)
    try:
        _cone_vs_full(world, 2e-6 if gemm_mode == 0 else 2e-5)
    finally:
        lib.gm_set_gemm_mode(prev)


def _cone_vs_full(world, atol):
    b = world['batch']
    a0, g0 = _step(_meta(world), b)
    a1, g1 = _step(_meta(world, cone=1), b)
    a2, g2 = _step(_meta(world, cone=1), b)
    a3, g3 = _step(_meta(world, cone=1, hoist_z1=1), b)
    a4, g4 = _step(_meta(world, cone=1, serialize=1), b)
    assert np.array_equal(a1, a2) and torch.equal(g1, g2)
    assert np.array_equal(a1, a3) and torch.equal(g1, g3)
    assert np.array_equal(a1, a4) and torch.equal(g1, g4)
    np.testing.assert_allclose(a1, a0, atol=1e-6)
    assert torch.allclose(g1, g0, atol=atol, rtol=1e-4), float((g1 - g0).abs().max())


def test_cone_tables_match_numpy_construction(world):
    """The receptive-field tables are integer work: bit-exact against a numpy construction from the batch CSR."""
    import ctypes as C
    import gmeta_amd
    from gmeta_amd import _lib
    lib = _lib.lib()
    q = gmeta_amd.SubgraphBatch.concat(list(world['batch'][2]))
    L = world['cfg']['h']
    _lib.check(lib.gm_batch_prepare_cone(q.handle, L, None), 'prepare_cone')
    ok = C.c_int32(); nrows = (C.c_int64 * (L + 1))(); nedges = (C.c_int64 * (L + 1))()
    _lib.check(lib.gm_batch_cone_dims(q.handle, L, C.byref(ok), nrows, nedges), 'cone_dims')
    assert ok.value == 1

    def rd(level, what, n):
        a = np.empty(n, np.int32)
        _lib.check(lib.gm_batch_cone_read(q.handle, L, level, what, a.ctypes.data, a.nbytes), 'cone_read')
        return a
    (indptr, indices), (indptr_t, indices_t) = q.csr(), q.csr(True)
    sub_off = q.sub_off
    up = (sub_off[:-1] + np.asarray(q.centres_local()).reshape(-1)).astype(np.int64)      # level L: centre rows, centre order
    set_row_off = sub_off[q.set_sub_off]
    row_of_edge_t = np.repeat(np.arange(q.rows), np.diff(indptr_t))
    for l in range(L, 0, -1):
        rows_up = rd(l, 0, nrows[l])
        assert np.array_equal(rows_up, up)
        deg = indptr[up + 1] - indptr[up]
        src = np.concatenate([indices[indptr[r]:indptr[r + 1]] for r in up]) if len(up) else np.zeros(0, np.int64)
        lo = np.unique(src)
        assert nrows[l - 1] == len(lo) and nedges[l] == len(src)
        pos_lo = np.full(q.rows, -1, np.int64); pos_lo[lo] = np.arange(len(lo))
        pos_up = np.full(q.rows, -1, np.int64); pos_up[up] = np.arange(len(up))
        assert np.array_equal(rd(l, 1, len(up) + 1), np.concatenate([[0], np.cumsum(deg)]))
        assert np.array_equal(rd(l, 2, len(src)), pos_lo[src])
        # by-source CSR: out-edges of every lower row that end in the upper level, in the batch's by-source order
        keep = (pos_lo[row_of_edge_t] >= 0) & (pos_up[indices_t] >= 0)
        cnt = np.bincount(pos_lo[row_of_edge_t[keep]], minlength=len(lo))
        assert np.array_equal(rd(l, 3, len(lo) + 1), np.concatenate([[0], np.cumsum(cnt)]))
        assert np.array_equal(rd(l, 4, len(src)), pos_up[indices_t[keep]])
        assert np.array_equal(rd(l - 1, 5, q.sets + 1), np.searchsorted(lo, set_row_off))
        up = lo
    assert np.array_equal(rd(0, 0, nrows[0]), up)


def test_batched_tasks_equal_per_task_runs(world):
    """Tasks are independent inside forward_ProtoMAML (meta.py:118-157): the batched step's meta-gradient is the mean of
    the single-task steps' and its accs their mean."""
    b = world['batch']
    a_all, g_all = _step(_meta(world), b)
    acc_sum, g_sum = 0, 0
    db = world['db']
    for t in range(T):
        one = db.get_batch([t])
        # same relabelled targets as in the batched call (labels are re-drawn per get_batch call in Disjoint mode)
        one = list(one); one[1] = [b[1][t]]; one[3] = [b[3][t]]
        a, g = _step(_meta(world), tuple(one))
        acc_sum = acc_sum + a; g_sum = g_sum + g
    np.testing.assert_allclose(a_all, acc_sum / T, atol=1e-6)
    assert torch.allclose(g_all, g_sum / T, atol=1e-5, rtol=1e-4)


def test_split_bf16_gemm_equals_exact_fp32_gemm(world):
    """The default GEMM arithmetic (fp32 operands split exactly into three bf16 pieces, six MFMA products, fp32 accumulation) against
    the exact-fp32 MFMA kernels on the SAME meta-step at the arxiv shape: identical accuracies, losses and meta-gradient to fp32
    rounding -- two orders of magnitude inside the 1e-4 tolerance of BASELINE.json -- and both bitwise reproducible."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    prev = lib.gm_get_gemm_mode()
    try:
        lib.gm_set_gemm_mode(0)
        m0 = _meta(world); a0, g0 = _step(m0, world['batch']); l0 = m0.last_stats['losses_q']
        lib.gm_set_gemm_mode(1)
        m1 = _meta(world); a1, g1 = _step(m1, world['batch']); l1 = m1.last_stats['losses_q']
        a2, g2 = _step(_meta(world), world['batch'])
    finally:
        lib.gm_set_gemm_mode(prev)
    assert np.array_equal(a1, a2) and torch.equal(g1, g2)
    np.testing.assert_allclose(a1, a0, atol=1e-6)
    np.testing.assert_allclose(l1, l0, atol=2e-6, rtol=2e-6)
    d = float((g1 - g0).abs().max()); scale = float(g0.abs().max())
    assert d <= 2e-5 and d <= 2e-3 * scale, (d, scale)


def test_fused_aggregate_gemm_is_bitwise_the_unfused_pass(world):
    """Forward passes nobody differentiates (the inner steps' query evaluations, finetunning's query passes) form the aggregate of rows
    with one or two sources inside the GEMM's operand feeders (k_gemm_split_p<true>) instead of writing Z through HBM: same fma order,
    so the whole meta-step -- accuracies, every loss, the meta-gradient -- and the finetunning accuracies are BITWISE those of the
    unfused pass (learner.py:41-47 either way) when that pass aggregates with the window kernel.  (The stream kernel, which takes the unfused full
    launches of batches this size by default, sums a hub row's parts with another association: there the two agree to rounding -- last assertion.)"""
    from gmeta_amd import _lib
    lib = _lib.lib()
    b = world['batch']
    try:
        lib.gm_set_fuse_agg(0)
        ms = _meta(world); as_, gs = _step(ms, b); ls = np.asarray(ms.last_stats['losses_q']).copy()      # unfused, stream kernel (the default)
        _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 0), 'set_tuning')
        m0 = _meta(world); a0, g0 = _step(m0, b); l0 = np.asarray(m0.last_stats['losses_q']).copy()
        f0 = np.asarray(m0.finetunning_batch(b[0], b[1], b[2], b[3]))
        lib.gm_set_fuse_agg(1)
        assert lib.gm_get_fuse_agg() == 1
        m1 = _meta(world); a1, g1 = _step(m1, b); l1 = np.asarray(m1.last_stats['losses_q']).copy()
        f1 = np.asarray(m1.finetunning_batch(b[0], b[1], b[2], b[3]))
    finally:
        lib.gm_set_fuse_agg(-1)
        _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 1), 'set_tuning')
    assert np.array_equal(a0, a1) and np.array_equal(l0, l1) and torch.equal(g0, g1)
    assert np.array_equal(f0, f1)
    np.testing.assert_allclose(as_, a1, atol=1e-6)
    np.testing.assert_allclose(ls, l1, atol=2e-6, rtol=0)
    assert float((gs - g1).abs().max()) <= 1e-5


@pytest.mark.parametrize('tasks', [1, 3])
def test_fused_aggregate_gemm_bitwise_on_small_task_counts(world, tasks):
    """Same as above on 1- and 3-task batches (36k / 107k query rows: workgroups of the persistent kernel get one or two tiles, sets end
    in partial tiles) and through the one-stream path: the counted prefetch queues of the fused feeders at their boundary cases."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    b = world['db'].get_batch(list(range(tasks)))
    out = []
    try:
        _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 0), 'set_tuning')       # (bitwise against the window kernel's unfused pass: see the test above)
        for fuse in (0, 1):
            lib.gm_set_fuse_agg(fuse)
            m = _meta(world, serialize=1 if tasks == 1 else 0)
            a, g = _step(m, b)
            out.append((a, g, np.asarray(m.last_stats['losses_q']).copy(), np.asarray(m.finetunning_batch(b[0], b[1], b[2], b[3]))))
    finally:
        lib.gm_set_fuse_agg(-1)
        _lib.check(lib.gm_set_tuning(b'GM_AGG_STREAM', 1), 'set_tuning')
    (a0, g0, l0, f0), (a1, g1, l1, f1) = out
    assert np.array_equal(a0, a1) and np.array_equal(l0, l1) and torch.equal(g0, g1) and np.array_equal(f0, f1)
