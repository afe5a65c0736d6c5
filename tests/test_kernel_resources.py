"""CPU: resource budget of the stream aggregate kernel, read from the KERNEL DESCRIPTORS of the built object (not from the metadata: hipcc pads
`amdhsa_next_free_vgpr` beyond the registers the code uses when it can derive an occupancy bound, e.g. from a static LDS size -- that is exactly how
the first version of this kernel ended up with 136 registers per wave and never ran beside anything).

k_agg_stream exists to fit into what a persistent split-GEMM workgroup leaves of a CU (tools/coreside_micro.hip: 16 waves x 120 VGPRs leave 32 registers
per SIMD lane; a workgroup that needs 35 waits for the GEMM to end): every instantiation must allocate <= 32 VGPRs, and the GEMM must not grow past 120."""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
TOOLS = ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf', 'llvm-objdump')
CXXFILT = os.path.join(LLVM, 'llvm-cxxfilt') if os.path.exists(os.path.join(LLVM, 'llvm-cxxfilt')) else shutil.which('c++filt')


def _device_elf(obj, tmp):
    fb, dev = os.path.join(tmp, 'fb.bin'), os.path.join(tmp, 'dev.o')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fb, obj], check=True)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fb, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                    '--output=' + dev], check=True, cwd=tmp)
    return dev


def _kernel_vgprs(dev):
    """{kernel: allocated VGPRs} from compute_pgm_rsrc1 of every <kernel>.kd (granulated count in bits 5:0, granule 8 on gfx950)."""
    sec = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-S', '-W', dev], check=True, capture_output=True, text=True).stdout
    m = re.search(r'\]\s+\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)', sec)
    addr, off, size = (int(x, 16) for x in m.groups())
    data = open(dev, 'rb').read()
    out = {}
    sym = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-s', '-W', dev], check=True, capture_output=True, text=True).stdout
    for line in sym.splitlines():
        f = line.split()
        if len(f) >= 8 and f[-1].endswith('.kd'):
            a = int(f[1], 16)
            assert addr <= a < addr + size
            rsrc1 = struct.unpack_from('<I', data, off + (a - addr) + 48)[0]
            out[f[-1][:-3]] = ((rsrc1 & 0x3f) + 1) * 8
    return out


def _demangled(names):
    out = subprocess.run([CXXFILT] + list(names), check=True, capture_output=True, text=True).stdout.split('\n')
    return dict(zip(names, out))


@pytest.fixture(scope='module')
def objs():
    missing = [t for t in TOOLS if not os.path.exists(os.path.join(LLVM, t))]
    if missing or not CXXFILT:
        pytest.skip('ROCm LLVM tools not found under %s: %s' % (LLVM, ', '.join(missing)))
    build = os.path.join(ROOT, 'g-meta_amd', 'csrc', 'build')
    if not os.path.exists(os.path.join(build, 'agg_stream.o')):
        if not (os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('hipcc')):
            pytest.skip('no built objects and no hipcc to build them')
        import __graft_entry__
        __graft_entry__.build()
    return build


def test_stream_aggregate_allocates_at_most_32_vgprs(objs):
    with tempfile.TemporaryDirectory() as tmp:
        regs = _kernel_vgprs(_device_elf(os.path.join(objs, 'agg_stream.o'), tmp))
    stream = {k: v for k, v in regs.items() if 'k_agg_stream' in k}
    assert len(stream) >= 4
    assert all(v <= 32 for v in stream.values()), stream


def test_persistent_split_gemm_leaves_32_vgprs_per_simd_lane(objs):
    with tempfile.TemporaryDirectory() as tmp:
        regs = _kernel_vgprs(_device_elf(os.path.join(objs, 'gemm.o'), tmp))
    # the 128 x 256 three-piece tiles of the large launches: k_gemm_split_p<GATHER, MI = 2, WC = 4, NP = 3>, selected by DEMANGLED name
    names = _demangled([k for k in regs if 'k_gemm_split_p' in k])
    big = {k: regs[k] for k, d in names.items() if re.search(r'k_gemm_split_p<(true|false), 2, 4, 3>', d)}
    assert len(big) >= 1, names
    assert all(4 * v + 32 <= 512 for v in big.values()), big                                        # 16 waves = 4 per SIMD


def test_weight_gradient_loader_never_touches_a_register_with_a_load_in_flight(objs):
    """k_wgrad_split keeps two stages of operand rows in flight in registers the COMPILER allocates around hand-counted inline-asm loads: nothing tells it
    that a destination register is not there yet, so a phi copy on the loop back-edge or a reused temporary silently reads / clobbers rows in flight (a round-6
    variant of the loader came out with eleven such copies at its loop head and faulted on the GPU).  tools/check_inflight_regs.py replays every
    instantiation's ISA -- prologue, loop body twice -- against the hardware rule (loads return in order, vmcnt(N) retires all but the newest N)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_inflight_regs as chk
    with tempfile.TemporaryDirectory() as tmp:
        dev = chk.device_elf(os.path.join(objs, 'gemm.o'), tmp)
        names = sorted(set(n for n in chk.kernels(dev) if 'k_wgrad_split' in n))
        assert len(names) >= 12, names                        # {1,2} x {1,2} tiles x (three-piece, two-piece, three-piece table-formed A)
        bad = {n: chk.check(chk.disasm(dev, n))[:3] for n in names}
    assert not any(bad.values()), {n: h for n, h in bad.items() if h}
