"""Mirror of G-Meta/meta.py: Meta (ProtoMAML inner/outer loop).  Same constructor, same 11-argument
forward / finetunning, same return values; the whole task loop runs inside one gm_meta_step call
(include/gmeta_hip.h) with every task of the meta-batch batched per kernel launch.  When
torch.distributed is initialised each rank passes ITS shard of the meta-batch and the first-order
meta-gradient is summed by one all-reduce (RCCL over xGMI on MI355X) before the NaN guard and Adam."""
import ctypes as C
import copy
import weakref

import numpy as np
import torch
from torch import nn, optim

from . import _lib
from .learner import Classifier
from .subgraphs import SubgraphBatch


import os as _os
# how the per-step all-reduce is issued: 'plain' (default) queues it behind the meta-step on the stream, no host synchronisation;
# 'async' = async_op + stream-level wait; 'sync' drains the compute stream first (round 2's workaround, see forward_deferred)
_ALLREDUCE_MODE = _os.environ.get('GMETA_ALLREDUCE_MODE', 'plain')


class Meta(nn.Module):
    def __init__(self, args, config):                                          # meta.py:83-99
        super(Meta, self).__init__()
        self.update_lr = args.update_lr
        self.meta_lr = args.meta_lr
        self.n_way = args.n_way
        self.k_spt = args.k_spt
        self.k_qry = args.k_qry
        self.task_num = args.task_num
        self.update_step = args.update_step
        self.update_step_test = args.update_step_test
        self.net = Classifier(config)
        if torch.cuda.is_available():
            self.net = self.net.to('cuda')
        self.meta_optim = self._make_adam()
        self.method = args.method
        self.hoist_z1 = int(getattr(args, 'hoist_z1', 0))
        self.serialize = int(getattr(args, 'serialize', 0))     # 1: one stream (per-kernel timing)
        self.sparse_bwd = int(getattr(args, 'sparse_bwd', 0))   # 1: exact row-sparse backward (flagged schedule)
        self.cone = int(getattr(args, 'cone', 0))               # 1: receptive-field schedule, forward and backward (flagged)
        self.last_stats = {}
        self.fused_adam_kernel = True                           # False: gm_meta_finish + torch's own fused Adam launches (the rule is the same)
        self._ws = None
        self._keep = None
        self._flat_grad = None
        self._flat_theta_buf = None
        self._found_inf = None

    _TRANSIENT = ('_keep', '_ws', '_flat_grad', '_flat_theta_buf', '_found_inf', '_sizes', '_hp', '_rb_ring', '_unchecked', '_adam_m', '_adam_v', '_adam_steps', '_adam_ticket', '_plist')      # device caches / ctypes handles: never copied

    def __deepcopy__(self, memo):
        """train.py:87,127 deep-copies the Meta object (best-model snapshot); parameters, buffers and the optimiser
        state are copied, per-call device caches are not."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in self._TRANSIENT else copy.deepcopy(v, memo)
        return new

    # ---- helpers
    def _make_adam(self):
        params = self.net.parameters()
        self._adam_fused = False
        if all(p.is_cuda for p in params):
            try:                                   # one fused kernel instead of ~7 foreach launches (same update rule)
                opt = optim.Adam(params, lr=self.meta_lr, fused=True)
                self._adam_fused = True
                return opt
            except (TypeError, RuntimeError):
                pass
        return optim.Adam(params, lr=self.meta_lr)

    def _apply(self, fn, *a, **k):                 # .to(device) moves parameters: rebuild the optimiser on the new tensors
        r = super(Meta, self)._apply(fn, *a, **k)
        if getattr(self, 'meta_optim', None) is not None and not self.meta_optim.state:
            self.meta_optim = self._make_adam()
        return r

    def _params(self):
        """The Parameter objects of the learner as a plain list.  Iterating / indexing an nn.ParameterList goes through string keys (~2 us per access:
        26 accesses per meta-step were 50 us of the host prologue, during which the GPU sits idle); the list is rebuilt when the ParameterList
        object or its length changes (nothing else replaces the Parameter objects: .to() / load_state_dict swap their .data in place)."""
        pl = self.net.parameters()
        c = getattr(self, '_plist', None)
        if c is None or c[0] is not pl or c[1] != len(pl):
            c = self._plist = (pl, len(pl), list(pl))
        return c[2]

    def _bind_flat(self, attr, dev, grads):
        """Every parameter (grads=False) or its .grad (grads=True) is a view into ONE flat fp32 buffer: the kernels read
        theta / write the meta-gradient without a gather or scatter launch.  Re-bound whenever something (deepcopy, .to(),
        zero_grad(set_to_none), load_state_dict on fresh tensors) broke the aliasing; values are preserved."""
        params = self._params()
        P = sum(p.numel() for p in params)
        buf = getattr(self, attr, None)
        ok = buf is not None and buf.numel() == P and buf.device == dev
        if ok:
            off = 0
            for p in params:
                t = p.grad if grads else p
                if t is None or t.data_ptr() != buf.data_ptr() + 4 * off or not t.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            buf = torch.zeros(P, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in params:
                    view = buf[off:off + p.numel()].view_as(p)
                    if grads:
                        p.grad = view
                    else:
                        view.copy_(p.detach())
                        p.data = view
                    off += p.numel()
            setattr(self, attr, buf)
        return buf

    def _bind_adam(self, dev):
        """The optimiser state of torch.optim.Adam (meta_optim.state[p] = {'step', 'exp_avg', 'exp_avg_sq'}) as views of flat buffers laid out like
        the flat parameter vector, so that gm_meta_finish_adam updates them in place and meta_optim (state_dict, deepcopy, a later plain
        meta_optim.step()) keeps seeing the true state.  Re-bound -- values preserved -- whenever something broke the aliasing (deepcopy,
        load_state_dict, .to())."""
        params = self._params()
        P = sum(p.numel() for p in params)
        st = self.meta_optim.state
        m_, v_, steps = getattr(self, '_adam_m', None), getattr(self, '_adam_v', None), getattr(self, '_adam_steps', None)
        ok = m_ is not None and m_.numel() == P and m_.device == dev and steps.numel() == len(params)
        if ok:
            off = 0
            for i, p in enumerate(params):
                s = st.get(p)
                if (not s or s['exp_avg'].data_ptr() != m_.data_ptr() + 4 * off or s['exp_avg_sq'].data_ptr() != v_.data_ptr() + 4 * off or
                        s['step'].data_ptr() != steps.data_ptr() + 4 * i):
                    ok = False
                    break
                off += p.numel()
        if not ok:
            m_ = torch.zeros(P, dtype=torch.float32, device=dev); v_ = torch.zeros(P, dtype=torch.float32, device=dev)
            steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for i, p in enumerate(params):
                    s = st.get(p) or {}
                    mv, vv, sv = m_[off:off + p.numel()].view_as(p), v_[off:off + p.numel()].view_as(p), steps[i:i + 1].view(())
                    if 'exp_avg' in s:
                        mv.copy_(s['exp_avg']); vv.copy_(s['exp_avg_sq']); sv.fill_(float(s['step']))
                    st[p] = {'step': sv, 'exp_avg': mv, 'exp_avg_sq': vv}
                    off += p.numel()
            self._adam_m, self._adam_v, self._adam_steps = m_, v_, steps
            self._adam_ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        return m_, v_, steps, self._adam_ticket

    def _bind_grads(self, dev):
        return self._bind_flat('_flat_grad', dev, True)

    def _flat_theta(self):
        p0 = self._params()[0]
        if not p0.is_cuda:
            return torch.cat([p.detach().reshape(-1) for p in self.net.parameters()]).contiguous()
        return self._bind_flat('_flat_theta_buf', p0.device, False)

    def _workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            had = self._ws is not None
            self._ws = None
            if had:                       # rare (a larger meta-batch than any before): hand the old block back instead of
                torch.cuda.empty_cache()  # leaving it parked in torch's caching allocator next to the new, larger one
            self._ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
        return self._ws

    _RB_RING_MAX = 8      # pinned read-back slots kept for reuse (a caller holding more unread handles than this gets throw-away slots)

    def _readback(self, t):
        """Asynchronous device -> pinned-host copy of `t` on the current stream; returns a slot [pinned buffer, event, busy] whose first
        t.numel() floats hold the data once the event has completed.  A slot is busy from here until its handle has been read OR has died
        (train.py reads one handle in `train_result_report_steps`: the others release their slot in _Deferred.__del__), so a training loop
        cycles through one or two slots however rarely it reads.  The ring is bounded: a caller that keeps more than _RB_RING_MAX unread
        handles alive gets extra slots that are dropped with their handle instead of being kept."""
        ring = getattr(self, '_rb_ring', None)
        if ring is None:
            ring = self._rb_ring = []
        slot = next((r for r in ring if not r[2] and r[0].numel() >= t.numel()), None)
        if slot is None:
            slot = [torch.empty(max(t.numel(), 1024), dtype=torch.float32, pin_memory=True), torch.cuda.Event(), False]
            if len(ring) < self._RB_RING_MAX:
                ring.append(slot)
        slot[2] = True
        slot[0][:t.numel()].copy_(t, non_blocking=True)
        slot[1].record()
        return slot, t.numel()

    def _run(self, x_spt, y_spt, x_qry, y_qry, K, need_grad):
        """One gm_meta_step over the local tasks.  Returns (out tensor [P + 2(K+1) + 1 + T(K+1) + 1], P, T); the last float is the
        violation word of the opt-in two-piece kernels (0 = none, see include/gmeta_hip.h)."""
        _lib.require_gpu()
        lib = _lib.lib()
        theta = self._flat_theta()
        dev = theta.device
        if dev.type != 'cuda':
            raise RuntimeError('Meta parameters must live on the GPU (call .to("cuda")); there is no CPU fallback')
        model = self.net.model
        P = int(lib.gm_model_param_count(C.byref(model)))
        if len(x_spt) == 0:               # an empty task shard (more ranks than tasks in a trailing meta-batch): contributes zeros
            return torch.zeros(P + 2 * (K + 1) + 2, dtype=torch.float32, device=dev), P, 0
        for b in x_spt:
            if not isinstance(b, SubgraphBatch):
                raise TypeError('x_spt / x_qry must be lists of gmeta_amd.SubgraphBatch (from gmeta_amd.Subgraphs)')
        for b in x_qry:
            if not isinstance(b, SubgraphBatch):
                raise TypeError('x_spt / x_qry must be lists of gmeta_amd.SubgraphBatch (from gmeta_amd.Subgraphs)')
        S, Q = SubgraphBatch.concat(list(x_spt)), SubgraphBatch.concat(list(x_qry))
        T = S.sets
        ys, yq = _labels(y_spt), _labels(y_qry)
        if len(ys) != S.subs or len(yq) != Q.subs:
            raise ValueError('label count does not match the number of subgraphs')
        # (this prologue sits between the read-back of one step and the first launch of the next, with the GPU idle: 90 -> 45 us on the FirstMM shape)
        hk = (float(self.update_lr), int(K), int(self.k_spt), int(need_grad), int(self.hoist_z1), int(self.serialize), int(self.sparse_bwd), int(self.cone))
        hpc = getattr(self, '_hp', None)
        if hpc is None or hpc[0] != hk:
            self._hp = hpc = (hk, _lib.HParams(*hk))
        hp = hpc[1]
        # output / workspace sizes depend on the two batches (their launch tables, not only their shapes), the model and the hyper-parameters:
        # remembered ON the support batch object, together with a weak reference to the query batch they were computed for (ids of freed
        # objects are recycled; a dead or different partner recomputes) -- two FFI calls (one of them a full planning pass) less on the host
        # path between the read-back of one step and the first launch of the next
        key = (S.rows, Q.rows, S.subs, Q.subs, S.sets, Q.sets, P, int(K), int(need_grad), int(self.hoist_z1), int(self.serialize), int(self.sparse_bwd),
               int(self.cone), int(lib.gm_get_gemm_mode()), int(lib.gm_get_split_pieces()), int(lib.gm_tuning_epoch()))
        sizes = getattr(S, '_meta_sizes', None)
        if sizes is None or sizes[0] != key or sizes[1]() is not Q:
            n_out = int(lib.gm_meta_out_floats(S.handle, C.byref(model), C.byref(hp)))
            ws_bytes = int(lib.gm_meta_ws_bytes(S.handle, Q.handle, C.byref(model), C.byref(hp)))
            if ws_bytes < 0 or n_out < 0:
                _lib.check(-1, 'gm_meta_ws_bytes')
            S._meta_sizes = (key, weakref.ref(Q), n_out, ws_bytes)
        else:
            n_out, ws_bytes = sizes[2], sizes[3]
        ws = self._workspace(ws_bytes, dev)
        out = torch.empty(n_out, dtype=torch.float32, device=dev)
        _lib.check(lib.gm_meta_step(S.handle, Q.handle, _lib.ptr(ys), _lib.ptr(yq), C.byref(model), C.byref(hp), _lib.ptr(theta),
                                    _lib.ptr(out), out.numel(), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), 'gm_meta_step')
        self._keep = (S, Q)             # keep concatenated batches alive until the stream has consumed them
        return out, P, T

    @staticmethod
    def _dist_on():
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    # ---- meta.py:101-173
    def forward_deferred(self, x_spt, y_spt, x_qry, y_qry, *unused):
        """Meta.forward without the device->host read: everything of meta.py:101-173 is queued on the stream -- the K-step
        inner loop over the local task shard, the all-reduce, the mean, the NaN guard and the Adam step all stay on the
        device -- and a handle is returned whose .accs() performs the one read.  `maml(...)` == forward_deferred(...).accs();
        a training loop that only looks at the accuracies every few steps (train.py:110) can keep the GPU busy while the host
        prepares the next meta-batch."""
        K = self.update_step
        if K < 2:
            raise ValueError('update_step must be >= 2: losses_q[0] and [1] are computed under no_grad (meta.py:129-141), '
                             'so the reference cannot back-propagate with fewer steps')
        prev = getattr(self, '_unchecked', None)
        if prev is not None:
            # opt-in two-piece mode only: the previous step's violation word is looked at HERE, before the next step is queued -- whether or
            # not the caller ever reads that step's accuracies (train.py reads one step in 30) -- and a violated step is re-run three-piece
            self._unchecked = None
            prev.accs()
        out, P, T = self._run(x_spt, y_spt, x_qry, y_qry, K, True)
        K1 = K + 1
        self._issued = getattr(self, '_issued', 0) + 1
        two_piece = _lib.lib().gm_get_split_pieces() == 2    # opt-in fast mode: a violation of its bounds re-runs the step (see _Deferred.accs)
        rerun = (self._issued, x_spt, y_spt, x_qry, y_qry) if two_piece else None
        head = out[:P + 2 * K1 + 1]                           # [grad | losses_q | corrects | task count], contiguous view
        if self._dist_on() and (torch.distributed.get_world_size() > 1 or getattr(self, 'force_allreduce', False)):
            # Round 2 drained the compute stream before the all-reduce (a +7 ms slow wait path when the collective was queued behind work on a
            # same-priority side stream).  With the query stream on its own low-priority hardware queue that path is gone: measured with a
            # 1-rank RCCL group (GMETA_FORCE_DIST=1), 4-task shard: no group 4.93 ms, plain 4.93, async 4.96, drained first 5.04 -- so the
            # all-reduce is simply queued on the stream and the host never waits (GMETA_ALLREDUCE_MODE=sync restores the drain).
            mode = _ALLREDUCE_MODE
            if mode == 'sync' and head.is_cuda:
                torch.cuda.current_stream().synchronize()
            if mode == 'async':
                torch.distributed.all_reduce(head, op=torch.distributed.ReduceOp.SUM, async_op=True).wait()     # stream-level wait only
            else:
                torch.distributed.all_reduce(head, op=torch.distributed.ReduceOp.SUM)
        if head.is_cuda and self._adam_fused:
            # The read-back of [losses_q | corrects | count | per-task | violation] is queued BEFORE the guard and the optimiser kernels (it does not
            # depend on them): the host gets the accuracies ~50 us earlier and prepares the next meta-step while Adam still runs
            rb = self._readback(out[P:])
            fg = self._bind_grads(head.device)               # (stands for meta_optim.zero_grad(); loss_q.backward())
            if getattr(self, '_found_inf', None) is None or self._found_inf.device != head.device:
                self._found_inf = torch.zeros((), dtype=torch.float32, device=head.device)      # 0-dim: what GradScaler hands a fused optimiser
            g = self.meta_optim.param_groups[0]
            own = (self.fused_adam_kernel and 'step' not in vars(self.meta_optim) and len(self.meta_optim.param_groups) == 1 and not g.get('amsgrad') and
                   not g.get('maximize') and g.get('weight_decay', 0) == 0 and not isinstance(g['lr'], torch.Tensor))
            if own:
                # mean + NaN guard + the Adam rule in ONE launch (gm_meta_finish_adam) on the optimiser's own state tensors (views of flat
                # buffers): between two meta-steps the stream carries one kernel instead of four, and the host does not spend ~100 us inside
                # optimizer.step() -- on the small configurations that gap was where the GPU sat empty
                m_, v_, steps, ticket = self._bind_adam(head.device)
                theta = self._flat_theta()
                _lib.check(_lib.lib().gm_meta_finish_adam(_lib.ptr(head), P, K1, _lib.ptr(theta), _lib.ptr(m_), _lib.ptr(v_), _lib.ptr(fg), _lib.ptr(steps), steps.numel(),
                                                          float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), _lib.ptr(self._found_inf),
                                                          _lib.ptr(ticket), _lib.stream_ptr()), 'gm_meta_finish_adam')
            else:
                # (a caller that hooked meta_optim.step -- the usual way to look at the meta-gradient -- or changed the optimiser's options:)
                # mean + NaN guard on the device (gm_meta_finish), then torch's fused Adam with `found_inf`: the kernel skips the
                # update and the step counter is rolled back when the flag is set == `if torch.isnan(loss_q): pass` (meta.py:163-169)
                _lib.check(_lib.lib().gm_meta_finish(_lib.ptr(head), P, K1, _lib.ptr(fg), _lib.ptr(self._found_inf), _lib.stream_ptr()), 'gm_meta_finish')
                self.meta_optim.found_inf = self._found_inf
                self.meta_optim.grad_scale = None
                self.meta_optim.step()
            d = _Deferred(self, rb, K1, applied=True, rerun=rerun)
            if rerun is not None:
                self._unchecked = d          # (strong reference: checked at the start of the next step even if the caller drops the handle)
            return d
        # Non-fused Adam (optim.Adam(fused=True) unavailable) or CPU tensors: the NaN guard needs the loss on the host, so the update is
        # applied HERE -- every meta-batch steps the optimiser like meta.py:163-169, whether or not the caller ever reads the
        # accuracies (train.py only reads them on report steps); only the handle's bookkeeping is left for .accs()
        d = _Deferred(self, out, K1, applied=False, P=P, rerun=rerun)
        d.accs()
        return d

    def forward_ProtoMAML(self, x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat):
        return self.forward_deferred(x_spt, y_spt, x_qry, y_qry).accs()

    # ---- meta.py:175-234
    def finetunning_ProtoMAML(self, x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat):
        K = self.update_step_test
        K1 = K + 1
        return self._eval_accs(x_spt[:1], y_spt[:1], x_qry[:1], y_qry[:1], K)[0]        # `[0]` of every argument (meta.py:182-191)

    def _eval_accs(self, x_spt, y_spt, x_qry, y_qry, K):
        """Per-task accuracies [T, K+1] of a forward-only gm_meta_step (host array).  With the opt-in two-piece kernels a step whose bounds
        were violated (last float of `out`) is run again with the three-piece kernels: the reference's fp32 path has no such limit."""
        K1 = K + 1
        out, P, T = self._run(x_spt, y_spt, x_qry, y_qry, K, False)
        tail = out[P + 2 * K1 + 1:].cpu().numpy().astype(np.float64)
        if tail[-1] != 0.0:
            lib = _lib.lib()
            lib.gm_set_split_pieces(3)
            try:
                out, P, T = self._run(x_spt, y_spt, x_qry, y_qry, K, False)
                tail = out[P + 2 * K1 + 1:].cpu().numpy().astype(np.float64)
            finally:
                lib.gm_set_split_pieces(2)
        return tail[:T * K1].reshape(T, K1)

    def finetunning_batch(self, x_spt, y_spt, x_qry, y_qry, shard=False):
        """All given evaluation tasks in ONE call (the reference loops 100 val/test tasks one at a time,
        train.py:118-121); returns accs [T, K_test+1].  shard=True under torch.distributed: every rank fine-tunes a
        contiguous slice of the tasks (they are independent and nothing is updated, meta.py:181) and the per-task
        accuracies are all-gathered, so every rank returns the full [T, K_test+1] array."""
        K = self.update_step_test
        K1 = K + 1
        n = len(x_spt)
        world = torch.distributed.get_world_size() if (shard and self._dist_on()) else 1
        if world == 1 and not (shard and self._dist_on() and getattr(self, 'force_allreduce', False)):
            return self._eval_accs(x_spt, y_spt, x_qry, y_qry, K)
        rank = torch.distributed.get_rank()
        bounds = np.linspace(0, n, world + 1).round().astype(int)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        dev = self.net.parameters()[0].device
        mine = torch.from_numpy(self._eval_accs(x_spt[lo:hi], y_spt[lo:hi], x_qry[lo:hi], y_qry[lo:hi], K)).to(dev, torch.float32) if hi > lo \
            else torch.zeros(0, K1, dtype=torch.float32, device=dev)
        return gather_rows(mine, bounds, K1).cpu().numpy().astype(np.float64)

    def forward(self, x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat):
        if self.method == 'G-Meta':
            accs = self.forward_ProtoMAML(x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat)
        return accs

    def finetunning(self, x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat):
        if self.method == 'G-Meta':
            accs = self.finetunning_ProtoMAML(x_spt, y_spt, x_qry, y_qry, c_spt, c_qry, n_spt, n_qry, g_spt, g_qry, feat)
        return accs


def _labels(ys):
    """The per-task label arrays of a meta-batch as one contiguous int32 vector (torch CPU tensors or numpy arrays)."""
    if len(ys) and all(isinstance(y, torch.Tensor) and not y.is_cuda for y in ys):
        ys = [y.numpy() for y in ys]                                               # (views: no copy)
    return np.concatenate([np.asarray(y).reshape(-1) for y in ys]).astype(np.int32, copy=False)


def gather_rows(mine, bounds, width):
    """all_gather of per-rank row blocks of unequal height: rank r owns rows [bounds[r], bounds[r+1]) of the result."""
    world = len(bounds) - 1
    cap = int(max(bounds[r + 1] - bounds[r] for r in range(world)))
    pad = torch.zeros(cap, width, dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    torch.distributed.all_gather(parts, pad)
    return torch.cat([parts[r][:int(bounds[r + 1] - bounds[r])] for r in range(world)])


class _Deferred:
    """Result handle of Meta.forward_deferred: .accs() reads losses/accuracies back (the only device->host sync of a
    meta-step) and returns np.array(corrects) / task_num (meta.py:171).

    Opt-in two-piece kernels only (gm_set_split_pieces(2)): the read-back also carries the step's violation word.  A violated bound (a fast
    weight that outgrew the step's weight bound, an operand maximum beyond the scale range) made gm_meta_step report a NaN query loss, so
    the optimiser step was skipped on every rank; the step is then run AGAIN with the three-piece kernels (rerun = the step's inputs) --
    the reference's fp32 arithmetic has no such limit and only skips on a true NaN (meta.py:163-169)."""

    def __init__(self, meta, buf, K1, applied, P=0, rerun=None):
        self._meta, self._buf, self._K1, self._applied, self._P, self._rerun = meta, buf, K1, applied, P, rerun
        self._accs = None

    def __del__(self):
        # an unread handle gives its pinned read-back slot back (Meta._readback): the copy into it was queued before any later step's copy
        # on the same stream, so a later reuse cannot be overtaken by it
        # (only while this handle still OWNS the slot: accs() gives it back and drops the reference in one statement, so a handle whose accs() raised
        # after that point can never mark a slot free that has meanwhile gone to a newer handle)
        buf = getattr(self, '_buf', None)
        if getattr(self, '_applied', False) and isinstance(buf, tuple):
            self._buf = None
            buf[0][2] = False

    def accs(self):
        if self._accs is not None:
            return self._accs
        m, K1 = self._meta, self._K1
        if self._buf is None:
            raise RuntimeError('gmeta_amd: the read-back of this meta-step was already consumed by an accs() call that failed')
        if self._applied:                                     # device path: Adam already queued, buf = (pinned [losses_q | corrects | count | per-task | violation], event)
            slot, n = self._buf
            slot[1].synchronize()           # (polling the event instead measured no different: the wait's wake-up is not what the step start waits for)
            tail = slot[0][:n].numpy().astype(np.float64)
            self._buf, slot[2] = None, False                  # (astype copied: the pinned slot may be reused -- and is no longer this handle's to free)
        else:                                                 # host path (non-fused Adam / CPU tensors): guard + step here
            head, P = self._buf, self._P
            tail = head[P:].cpu().numpy().astype(np.float64)
        task_num = float(tail[2 * K1])
        loss_q = tail[K1 - 1] / task_num                      # losses_q[-1] / task_num (meta.py:161)
        if self._rerun is not None and np.isnan(loss_q):
            viol = float(tail[-1])
            if m._dist_on() and torch.distributed.get_world_size() > 1:      # every rank sees the NaN of the reduced loss: agree on its cause
                v = torch.tensor([viol], dtype=torch.float32, device='cuda')
                torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX)
                viol = float(v.item())
            if viol != 0.0:
                issued, x_spt, y_spt, x_qry, y_qry = self._rerun
                self._rerun = None
                if getattr(m, '_issued', 0) != issued:
                    raise RuntimeError('gmeta_amd: a bound of the two-piece kernels was violated in a meta-step whose read-back was deferred past the next '
                                       'step; re-run with GM_SPLIT_PIECES=3 (the default) or read .accs() before queueing the next step')
                lib = _lib.lib()
                lib.gm_set_split_pieces(3)
                if getattr(m, '_unchecked', None) is self:       # this handle IS being checked: the re-run must not come back to it
                    m._unchecked = None
                try:
                    self._accs = m.forward_deferred(x_spt, y_spt, x_qry, y_qry).accs()
                finally:
                    lib.gm_set_split_pieces(2)
                m.last_stats['rerun_three_piece'] = viol
                self._buf = None
                return self._accs
        m.last_stats = {'loss_q': loss_q, 'losses_q': tail[:K1] / task_num, 'task_num': task_num}
        if not self._applied and not np.isnan(loss_q):        # meta.py:163-169
            fg = m._bind_grads(head.device)
            torch.div(head[:P], task_num, out=fg)
            m.meta_optim.step()
        self._accs = tail[K1:2 * K1] / task_num               # np.array(corrects) / task_num (meta.py:171)
        self._buf = None
        self._rerun = None
        return self._accs
