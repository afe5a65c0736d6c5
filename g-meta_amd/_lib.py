"""ctypes binding of libgmeta_hip.so (include/gmeta_hip.h).  Fails loudly if the library is absent."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GMETA_HIP_LIB') or os.path.join(HERE, 'libgmeta_hip.so')      # override: kernel experiments with variant builds

GM_MAX_GCN = 4
(F_SUB_OFF, F_SET_SUB_OFF, F_PARENT, F_GRAPH, F_INDPTR, F_INDICES, F_INDPTR_T, F_INDICES_T, F_CENTRE, F_NORM,
 F_FEAT_ROW) = range(11)


class Seed(C.Structure):
    _fields_ = [('graph', C.c_int32), ('i', C.c_int32), ('j', C.c_int32)]


class Model(C.Structure):
    _fields_ = [('n_gcn', C.c_int32), ('dims', C.c_int32 * (GM_MAX_GCN + 1)), ('n_out', C.c_int32), ('link_pred', C.c_int32)]


class HParams(C.Structure):
    _fields_ = [('update_lr', C.c_float), ('update_step', C.c_int32), ('k_spt', C.c_int32), ('need_meta_grad', C.c_int32),
                ('hoist_z1', C.c_int32), ('serialize', C.c_int32), ('sparse_bwd', C.c_int32), ('cone', C.c_int32)]


vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
PROTOTYPES = {
    'gm_last_error': (C.c_char_p, []),
    'gm_version': (C.c_int, []),
    'gm_store_create': (C.c_int, [i32, vp, vp, vp, vp, i32, vp]),
    'gm_store_destroy': (None, [vp]),
    'gm_extract': (C.c_int, [vp, vp, i32, vp, i32, i32, i32, u64, i32, vp, vp]),
    'gm_extract_pair': (C.c_int, [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, u64, i32, vp, vp, vp]),
    'gm_batch_from_nodes': (C.c_int, [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp]),
    'gm_batch_concat': (C.c_int, [vp, i32, vp, vp]),
    'gm_batch_destroy': (None, [vp]),
    'gm_batch_prepare_cone': (C.c_int, [vp, i32, vp]),
    'gm_batch_prepare_cone_pair': (C.c_int, [vp, vp, i32, vp]),
    'gm_batch_cone_dims': (C.c_int, [vp, i32, vp, vp, vp]),
    'gm_batch_cone_read': (C.c_int, [vp, i32, i32, i32, vp, i64]),
    'gm_batch_dims': (C.c_int, [vp, vp, vp, vp, vp, vp]),
    'gm_batch_read': (C.c_int, [vp, i32, vp, i64]),
    'gm_batch_device_ptr': (C.c_int, [vp, i32, vp]),
    'gm_gather_features': (C.c_int, [vp, vp, vp]),
    'gm_aggregate': (C.c_int, [vp, i32, i32, vp, i32, vp, vp, vp, vp]),
    'gm_aggregate_bytes': (i64, [vp, i32]),
    'gm_model_param_count': (i64, [vp]),
    'gm_gcn_ws_bytes': (i64, [vp, vp]),
    'gm_gcn_forward': (C.c_int, [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]),
    'gm_gcn_backward': (C.c_int, [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    'gm_dense_update': (C.c_int, [vp, vp, i32, vp, i64, i32, vp, i32, vp]),
    'gm_proto_loss_spt': (C.c_int, [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    'gm_proto_loss_qry': (C.c_int, [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
    'gm_meta_ws_bytes': (i64, [vp, vp, vp, vp]),
    'gm_meta_out_floats': (i64, [vp, vp, vp]),
    'gm_meta_step': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    'gm_meta_finish': (C.c_int, [vp, i64, i32, vp, vp, vp]),
    'gm_meta_finish_adam': (C.c_int, [vp, i64, i32, vp, vp, vp, vp, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp]),
    'gm_set_gemm_mode': (None, [i32]),
    'gm_get_gemm_mode': (i32, []),
    'gm_get_split_pieces': (i32, []),
    'gm_set_split_pieces': (None, [i32]),
    'gm_set_tuning': (C.c_int, [C.c_char_p, i32]),
    'gm_get_tuning': (i32, [C.c_char_p]),
    'gm_tuning_epoch': (i32, []),
    'gm_set_fuse_agg': (None, [i32]),
    'gm_get_fuse_agg': (i32, []),
    'gm_profile_enable': (None, [i32]),
    'gm_profile_aggregate': (C.c_int, [vp, vp, vp]),
    'gm_profile_read': (C.c_int, [i32, vp, vp, vp]),
    'gm_profile_read_launches': (C.c_int, [i32, vp, vp, i32]),
}
# include/gmeta_hip_probes.h: exported only by the probe build (build.py --probes -> libgmeta_hip_probes.so, loaded through GMETA_HIP_LIB by tools/)
PROBE_PROTOTYPES = {
    'gm_debug_stamp': (C.c_int, [vp, vp]),
    'gm_stream_debug': (C.c_int, [i32, vp, i32]),
    'gm_head_loss_debug': (C.c_int, [i32, vp]),
}

_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('gmeta_amd: %s is missing -- run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)           # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in PROBE_PROTOTYPES.items():
            if hasattr(l, name):            # the probe build only
                fn = getattr(l, name)
                fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().gm_last_error().decode('utf-8', 'replace')
        exc = ValueError if rc == -1 else RuntimeError
        raise exc('%s failed (code %d): %s' % (what or 'libgmeta_hip', rc, msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('gmeta_amd needs an AMD GPU (gfx950); torch.cuda.is_available() is False and there is no CPU fallback')


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, 'data_ptr'):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def make_model(config):
    """train.py:67-75 config list -> gm_model_t.  Mirrors learner.py:78-97 parsing."""
    gcn = [p for n, p in config if n == 'GraphConv']
    lin = [p for n, p in config if n == 'Linear']
    if any(n == 'Attention' for n, _ in config):
        raise NotImplementedError("the 'Attention' branch of learner.py:98-131 is dead code in the reference (uses an undefined name)")
    if not gcn or len(lin) != 1 or len(gcn) > GM_MAX_GCN:
        raise ValueError('config must hold 1..%d GraphConv entries and one Linear entry' % GM_MAX_GCN)
    m = Model()
    m.n_gcn = len(gcn)
    for l, (fi, fo) in enumerate(gcn):
        if l and m.dims[l] != fi:
            raise ValueError('GraphConv dims do not chain')
        m.dims[l], m.dims[l + 1] = fi, fo
    if lin[0][0] != m.dims[m.n_gcn]:
        raise ValueError('Linear input dim must equal the last GraphConv output dim')
    m.n_out = lin[0][1]
    m.link_pred = 1 if config[-1][0] == 'LinkPred' else 0
    return m
