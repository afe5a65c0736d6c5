"""Mirror of G-Meta/learner.py: Classifier (functional GCN stack + Linear head) with the same parameter
list, initialisation and forward signature, computed by the HIP kernels behind gm_gcn_forward /
gm_gcn_backward (include/gmeta_hip.h).  `g` is a gmeta_amd.SubgraphBatch instead of a batched DGLGraph."""
import ctypes as C

import torch
import torch.nn as nn
from torch.nn import init

from . import _lib
from .subgraphs import SubgraphBatch


class _GcnFunction(torch.autograd.Function):
    """logits = Classifier(params) on one SubgraphBatch; differentiable w.r.t. the flat parameter vector
    (features carry no gradient in the reference either: they come from numpy, meta.py:119)."""

    @staticmethod
    def forward(ctx, flat, batch, model, x0, centre):
        lib = _lib.lib()
        dev = flat.device
        P = int(lib.gm_model_param_count(C.byref(model)))
        if flat.numel() != P:
            raise ValueError('parameter vector has %d elements, config needs %d' % (flat.numel(), P))
        flat = flat.contiguous().float()
        ws_bytes = int(lib.gm_gcn_ws_bytes(batch.handle, C.byref(model)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        logits = torch.empty(batch.subs, model.n_out, dtype=torch.float32, device=dev)
        _lib.check(lib.gm_gcn_forward(batch.handle, C.byref(model), _lib.ptr(flat), 0, _lib.ptr(x0), _lib.ptr(centre), _lib.ptr(logits),
                                      _lib.ptr(ws), ws_bytes, _lib.stream_ptr()), 'gm_gcn_forward')
        ctx.batch, ctx.model, ctx.ws, ctx.x0, ctx.centre, ctx.P = batch, model, ws, x0, centre, P
        ctx.save_for_backward(flat)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (flat,) = ctx.saved_tensors
        lib = _lib.lib()
        b = ctx.batch
        dl = dlogits.contiguous().float()
        dparams = torch.empty(b.sets, ctx.P, dtype=torch.float32, device=flat.device)
        _lib.check(lib.gm_gcn_backward(b.handle, C.byref(ctx.model), _lib.ptr(flat), 0, _lib.ptr(ctx.x0), _lib.ptr(ctx.centre), _lib.ptr(dl),
                                       _lib.ptr(dparams), ctx.P, _lib.ptr(ctx.ws), ctx.ws.numel(), _lib.stream_ptr()), 'gm_gcn_backward')
        return dparams.sum(0), None, None, None, None


class Classifier(nn.Module):
    """learner.py:69-209.  `vars` order and shapes: [W1 [in,out], b1, (W2, b2, ...), Wl [C, H(*2)], bl]."""

    def __init__(self, config):
        super(Classifier, self).__init__()
        self.vars = nn.ParameterList()
        self.config = config
        self.LinkPred_mode = config[-1][0] == 'LinkPred'                       # learner.py:78-79
        self.model = _lib.make_model(config)
        for name, param in self.config:
            if name == 'Linear':                                               # learner.py:83-90
                w = nn.Parameter(torch.ones(param[1], param[0] * (2 if self.LinkPred_mode else 1)))
                init.kaiming_normal_(w)
                self.vars.append(w)
                self.vars.append(nn.Parameter(torch.zeros(param[1])))
            if name == 'GraphConv':                                            # learner.py:91-97
                w = nn.Parameter(torch.Tensor(param[0], param[1]))
                init.xavier_uniform_(w)
                self.vars.append(w)
                self.vars.append(nn.Parameter(torch.zeros(param[1])))

    def forward(self, g, to_fetch, features, vars=None):
        """learner.py:134-194.  g: SubgraphBatch; to_fetch: centre indices ([S] or [S,2]) or None (use the
        batch's own); features: device tensor [n, F0] or None (gather from the HBM-resident store)."""
        if vars is None:
            vars = self.vars
        if not isinstance(g, SubgraphBatch):
            raise TypeError('g must be a gmeta_amd.SubgraphBatch')
        if g.view_of is not None:
            raise ValueError('pass a whole SubgraphBatch (not a task view) to Classifier.forward')
        _lib.require_gpu()
        dev = vars[0].device
        if dev.type != 'cuda':
            raise RuntimeError('Classifier parameters must live on the GPU (no CPU fallback)')
        flat = torch.cat([v.reshape(-1) for v in vars])
        x0 = None if features is None else torch.as_tensor(features).float().to(dev).contiguous()
        centre = None if to_fetch is None else torch.as_tensor(to_fetch).to(dev).to(torch.int32).contiguous()
        h = _GcnFunction.apply(flat, g, self.model, x0, centre)
        return h, h

    def zero_grad(self, vars=None):                                            # learner.py:196-206
        with torch.no_grad():
            for p in (self.vars if vars is None else vars):
                if p.grad is not None:
                    p.grad.zero_()

    def parameters(self):                                                      # learner.py:208-209
        return self.vars
