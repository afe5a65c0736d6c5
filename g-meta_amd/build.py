"""Builds libgmeta_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build(); the built
.so is git-ignored but travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libgmeta_hip.so')
SOURCES = ['common.hip', 'store.hip', 'extract.hip', 'agg.hip', 'agg_stream.hip', 'gemm.hip', 'model.hip', 'cone.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probes=False):
    """probes=True: the same sources with -DGM_PROBES -> libgmeta_hip_probes.so (include/gmeta_hip_probes.h; tools/ only, never the product path)."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, 'build', 'probes') if probes else os.path.join(CSRC, 'build')
    out = os.path.join(HERE, 'libgmeta_hip_probes.so') if probes else OUT
    flags = FLAGS + (['-DGM_PROBES'] if probes else [])
    os.makedirs(objdir, exist_ok=True)
    # every header a translation unit can include: editing any of them (gemm_split.h is only included by gemm.hip) rebuilds the objects
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(HERE, '..', 'include', f) for f in ('gmeta_hip.h', 'gmeta_hip_probes.h')]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        if force or _newer(obj, [src] + headers):
            jobs.append([hipcc] + flags + ['-c', src, '-o', obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stderr))
        return r
    if jobs:
        if verbose:
            print('[gmeta_amd] compiling %d HIP translation unit(s) for gfx950' % len(jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _newer(out, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, probes='--probes' in sys.argv))
