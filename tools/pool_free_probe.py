#!/usr/bin/env python3
"""hipFreeAsync host cost by block size / order (ROCm 7.2 default memory pool, release threshold raised): what makes one free of a meta-batch cost 1.2 - 1.7 ms."""
import ctypes as C, time, sys
hip = C.CDLL('libamdhip64.so')
hip.hipMallocAsync.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
hip.hipFreeAsync.argtypes = [C.c_void_p, C.c_void_p]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
pool = C.c_void_p(); hip.hipDeviceGetDefaultMemPool(C.byref(pool), 0)
keep = C.c_uint64(2 ** 64 - 1); hip.hipMemPoolSetAttribute(pool, 4, C.byref(keep))     # hipMemPoolAttrReleaseThreshold
s = C.c_void_p(); hip.hipStreamCreate(C.byref(s))
def alloc(mb, touch=True):
    p = C.c_void_p(); t = time.perf_counter(); rc = hip.hipMallocAsync(C.byref(p), int(mb * (1 << 20)), s); dt = time.perf_counter() - t
    assert rc == 0, rc
    if touch: hip.hipMemsetAsync(p, 0, int(mb * (1 << 20)), s)
    return p, dt
def free(p):
    t = time.perf_counter(); rc = hip.hipFreeAsync(p, s); dt = time.perf_counter() - t; assert rc == 0; return dt
store, _ = alloc(128)
for rep in range(4):
    for sizes in ([182 + rep, 22.5 + rep * 0.7], [124, 31, 36, 15, 8], [300, 100, 50, 25]):
        ps = [alloc(m) for m in sizes]
        hip.hipStreamSynchronize(s)
        fr = [free(p) for p, _ in ps]
        hip.hipStreamSynchronize(s)
        print('sizes MB %-22s alloc us %-30s free us %s' % (sizes, [round(d * 1e6) for _, d in ps], [round(d * 1e6) for d in fr]))
