"""GPU (-m gpu): numerics of the two update-GEMM kernels (learner.py:36,47 `torch.matmul(feat, weight)`) at the arxiv query-batch
shape (286k rows, per-task weights) against an fp64 product: the split-bf16 kernel (every fp32 operand split exactly into three
bf16 pieces, six MFMA products, fp32 accumulation -- the default for large N = 256 / N = 128 launches) must be as accurate as the exact-fp32
MFMA kernel and as a plain PyTorch fp32 matmul."""
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
