"""GPU (-m gpu): numerics of the two update-GEMM kernels (learner.py:36,47 `torch.matmul(feat, weight)`) at the arxiv query-batch
shape (286k rows, per-task weights) against an fp64 product: the split-bf16 kernel (every fp32 operand split exactly into three
bf16 pieces, six MFMA products, fp32 accumulation -- the default for large N = 256 / N = 128 launches) must be as accurate as the exact-fp32
MFMA kernel and as a plain PyTorch fp32 matmul; and the same for the split-fp16 kernel gm_meta_step uses where magnitude bounds are recorded
(two fp16 pieces per operand under power-of-two scales, three products), including operands whose magnitudes span many octaves inside one
bound (the low pieces then sit in fp16's subnormal range: the matrix cores must not flush them)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def qbatch():
    import gmeta_amd
    from gmeta_amd import synth
    np.random.seed(222); random.seed(222); torch.manual_seed(222)
    T = 8
    args, cfg = synth.make_args('arxiv', task_num=T)
    data = synth.make_dataset(cfg)
    store = gmeta_amd.GraphStore(data['graphs'], data['feats'])
    db = gmeta_amd.Subgraphs(None, 'train', data['info'], n_way=3, k_shot=3, k_query=24, batchsz=T, args=args, adjs=store, h=2,
                             tables=data['tables'], verbose=False)
    batch = db.get_batch(list(range(T)))
    return batch[2][0].view_of, store


@pytest.mark.parametrize('K,N', [(256, 256), (128, 256), (128, 128), (64, 128)])
def test_split_bf16_gemm_is_as_accurate_as_fp32(qbatch, K, N):
    from gmeta_amd import _lib
    lib = _lib.lib()
    Q, _ = qbatch
    T = Q.sets
    g = torch.Generator(device='cuda').manual_seed(K)
    # wide dynamic range: rows scaled over three decades, a few exact zeros and denormal-ish values
    x = torch.randn(Q.rows, K, device='cuda', generator=g) * torch.logspace(-2, 1, Q.rows, device='cuda')[torch.randperm(Q.rows, device='cuda', generator=g)][:, None]
    x[::97, ::5] = 0.0
    W = torch.randn(T, K, N, device='cuda', generator=g) * 0.1
    outs = {}
    for mode in (0, 1):
        out = torch.empty(Q.rows, N, device='cuda')
        _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(x), K, _lib.ptr(W), K * N, N, _lib.ptr(out), mode, _lib.stream_ptr()), 'gm_dense_update')
        outs[mode] = out
    torch.cuda.synchronize()
    so = torch.from_numpy(Q.sub_off[Q.set_sub_off].astype(np.int64))           # row range of every set
    rows = torch.cat([torch.arange(int(so[t]), int(so[t + 1]), 37) for t in range(T)]).cuda()
    owner = (torch.searchsorted(so.cuda(), rows, right=True) - 1)
    ref = torch.bmm(x[rows].double().unsqueeze(1), W[owner].double()).squeeze(1)
    torch_f32 = torch.bmm(x[rows].unsqueeze(1), W[owner]).squeeze(1)
    e_f32 = float((outs[0][rows].double() - ref).abs().max())
    e_split = float((outs[1][rows].double() - ref).abs().max())
    e_torch = float((torch_f32.double() - ref).abs().max())
    mag = float(ref.abs().max())
    assert e_f32 <= 1e-5 * mag and e_split <= 1e-5 * mag, (e_f32, e_split, mag)
    assert e_split <= 1.5 * max(e_f32, e_torch) + 1e-7 * mag, (e_split, e_f32, e_torch)
    # element-wise: error normalised by the condition scale sum_k |x_k||w_k| (what a backward error analysis bounds): the split kernel's
    # worst normalised error is of the size of the fp32 kernels' and far below the K * 2^-24 worst-case bound of an fp32 dot product
    scale = torch.bmm(x[rows].double().abs().unsqueeze(1), W[owner].double().abs()).squeeze(1).clamp_min(1e-300)
    n_f32 = float(((outs[0][rows].double() - ref).abs() / scale).max())
    n_split = float(((outs[1][rows].double() - ref).abs() / scale).max())
    n_torch = float(((torch_f32.double() - ref).abs() / scale).max())
    assert n_split <= 2.0 * max(n_f32, n_torch) + 2.0 ** -23, (n_split, n_f32, n_torch)
    assert n_split <= K * 2.0 ** -24, (n_split, K * 2.0 ** -24)
    # run-to-run determinism of the split kernel
    out2 = torch.empty(Q.rows, N, device='cuda')
    _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(x), K, _lib.ptr(W), K * N, N, _lib.ptr(out2), 1, _lib.stream_ptr()), 'gm_dense_update')
    assert torch.equal(out2, outs[1])


@pytest.mark.parametrize('K,N', [(256, 256), (128, 256), (128, 128)])
def test_split_fp16_gemm_is_as_accurate_as_fp32(qbatch, K, N):
    """gm_dense_update mode 2: two fp16 pieces per operand (22 significand bits), three products.  Same bars as the three-piece kernel."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    Q, _ = qbatch
    T = Q.sets
    g = torch.Generator(device='cuda').manual_seed(1000 + K)
    x = torch.randn(Q.rows, K, device='cuda', generator=g) * torch.logspace(-2, 1, Q.rows, device='cuda')[torch.randperm(Q.rows, device='cuda', generator=g)][:, None]
    x[::97, ::5] = 0.0
    W = torch.randn(T, K, N, device='cuda', generator=g) * 0.1 * torch.logspace(-1, 1, T, device='cuda')[:, None, None]      # per-task weight scales: per-task bounds
    outs = {}
    for mode in (0, 2):
        out = torch.empty(Q.rows, N, device='cuda')
        _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(x), K, _lib.ptr(W), K * N, N, _lib.ptr(out), mode, _lib.stream_ptr()), 'gm_dense_update')
        outs[mode] = out
    torch.cuda.synchronize()
    so = torch.from_numpy(Q.sub_off[Q.set_sub_off].astype(np.int64))
    rows = torch.cat([torch.arange(int(so[t]), int(so[t + 1]), 37) for t in range(T)]).cuda()
    owner = (torch.searchsorted(so.cuda(), rows, right=True) - 1)
    ref = torch.bmm(x[rows].double().unsqueeze(1), W[owner].double()).squeeze(1)
    torch_f32 = torch.bmm(x[rows].unsqueeze(1), W[owner]).squeeze(1)
    scale = torch.bmm(x[rows].double().abs().unsqueeze(1), W[owner].double().abs()).squeeze(1).clamp_min(1e-300)
    n_f32 = float(((outs[0][rows].double() - ref).abs() / scale).max())
    n_16 = float(((outs[2][rows].double() - ref).abs() / scale).max())
    n_torch = float(((torch_f32.double() - ref).abs() / scale).max())
    # worst element-wise error over the condition scale sum_k |x_k||w_k|: of the size of the fp32 kernels', far below an fp32 dot product's K * 2^-24 bound
    assert n_16 <= 2.0 * max(n_f32, n_torch) + 2.0 ** -22, (n_16, n_f32, n_torch)
    assert n_16 <= K * 2.0 ** -24, (n_16, K * 2.0 ** -24)
    # rms error relative to each row's output scale: not above the fp32 kernels'
    rs = ref.pow(2).mean(1, keepdim=True).sqrt().clamp_min(1e-300)
    r_16 = float((((outs[2][rows].double() - ref) / rs) ** 2).mean().sqrt())
    r_f32 = float((((outs[0][rows].double() - ref) / rs) ** 2).mean().sqrt())
    r_torch = float((((torch_f32.double() - ref) / rs) ** 2).mean().sqrt())
    assert r_16 <= 1.25 * max(r_f32, r_torch), (r_16, r_f32, r_torch)
    out2 = torch.empty(Q.rows, N, device='cuda')
    _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(x), K, _lib.ptr(W), K * N, N, _lib.ptr(out2), 2, _lib.stream_ptr()), 'gm_dense_update')
    assert torch.equal(out2, outs[2])


def test_split_fp16_keeps_small_rows_under_a_large_bound(qbatch):
    """Rows 2^-1 .. 2^-20 below the bound that scales them (one bound for all of x here): the high piece of a small element is a normal fp16 number
    down to 2^-29 of the bound, its low piece a SUBNORMAL one below 2^-18 -- kept with 2^-24 absolute precision of the scaled value, i.e. a row at
    2^-r of the bound still comes out with ~(39 - r) good bits.  A matrix core that flushed fp16 subnormals would leave such rows at 11 bits."""
    from gmeta_amd import _lib
    lib = _lib.lib()
    Q, _ = qbatch
    K = N = 256
    g = torch.Generator(device='cuda').manual_seed(7)
    octave = (torch.arange(Q.rows, device='cuda') % 21).float()                       # row r scaled by 2^-(r % 21)
    x = torch.randn(Q.rows, K, device='cuda', generator=g).clamp(-4, 4) * torch.exp2(-octave)[:, None]
    W = torch.randn(Q.sets, K, N, device='cuda', generator=g) * 0.1
    out = torch.empty(Q.rows, N, device='cuda')
    _lib.check(lib.gm_dense_update(Q.handle, _lib.ptr(x), K, _lib.ptr(W), K * N, N, _lib.ptr(out), 2, _lib.stream_ptr()), 'gm_dense_update')
    torch.cuda.synchronize()
    so = torch.from_numpy(Q.sub_off[Q.set_sub_off].astype(np.int64))
    rows = torch.cat([torch.arange(int(so[t]), int(so[t + 1]), 11) for t in range(Q.sets)]).cuda()
    owner = (torch.searchsorted(so.cuda(), rows, right=True) - 1)
    ref = torch.bmm(x[rows].double().unsqueeze(1), W[owner].double()).squeeze(1)
    rel = ((out[rows].double() - ref).abs().max(1).values / ref.pow(2).mean(1).sqrt())   # per row, relative to the row's own output scale
    oc = octave[rows]
    for r in range(21):
        worst = float(rel[oc == r].max())
        # 2^-21 from the dropped product terms (x 16: the maximum over 256 columns x ~1,700 rows of a roughly Gaussian error of rms ~2e-7) + the
        # subnormal floor 2^-(39 - r - 3).  Flushed low pieces would show as ~2^-11 = 5e-4 from r ~ 18 on.
        assert worst <= 16 * 2.0 ** -21 + 2.0 ** -(36 - r), (r, worst)
