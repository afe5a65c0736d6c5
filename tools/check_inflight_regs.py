#!/usr/bin/env python3
"""Static scoreboard check of the hand-counted load queues (k_wgrad_split, the feeders of k_gemm_split_p): the compiler does not know that the
destination of an inline-asm `global_load` is not there yet -- it is free to copy such a register (a phi copy on the loop back-edge, a spill, a
rematerialised address) or to reuse it as a temporary before the s_waitcnt that covers the load.  Either reads garbage or gets overwritten when the
load lands (round 6: a variant of the weight-gradient loader came out with eleven v_mov of in-flight registers at its loop head and faulted).

The check walks the kernel's ISA as straight-line code -- prologue, then the main loop body twice (back-edge) -- with the hardware's rule: vector
memory loads return in order, `s_waitcnt vmcnt(N)` retires all but the newest N.  Any instruction that reads or writes a VGPR with a load still in
flight is reported.  Usage: check_inflight_regs.py <object.o> <mangled-kernel-substring> [...]; exit status 1 on a hazard."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')


def device_elf(obj, tmp):
    fb, dev = os.path.join(tmp, 'fb.bin'), os.path.join(tmp, 'dev.o')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fb, obj], check=True)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fb, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                    '--output=' + dev], check=True, cwd=tmp)
    return dev


def kernels(dev):
    sym = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-s', '-W', dev], check=True, capture_output=True, text=True).stdout
    return [l.split()[-1] for l in sym.splitlines() if ' FUNC ' in l and l.split()[-1].startswith('_Z')]


def disasm(dev, name):
    out = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--disassemble-symbols=' + name, dev], check=True, capture_output=True, text=True).stdout
    ins = []
    for l in out.splitlines():
        m = re.match(r'\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):', l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).strip()))
    return ins


def vregs(tok):
    """VGPR numbers named by an operand token: v12, v[4:7]"""
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def operands(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [t.strip() for t in re.split(r',\s*', parts[1])]


def branch_target(addr, text):
    m = re.match(r's_c?branch\S*\s+(-?\d+)', text)
    if not m:
        return None
    off = int(m.group(1))
    if off >= 32768:
        off -= 65536
    return addr + 4 + 4 * off


def check(ins, verbose=False):
    """-> list of hazards.  The main loop = the backward branch with the longest span."""
    loops = []
    for i, (a, t) in enumerate(ins):
        tgt = branch_target(a, t)
        if tgt is not None and tgt <= a:
            j = next((k for k, (b, _) in enumerate(ins) if b == tgt), None)
            if j is not None:
                loops.append((i - j, j, i))
    order = list(range(len(ins)))
    if loops:
        _, j, i = max(loops)
        order = list(range(0, i + 1)) + list(range(j, i + 1)) + list(range(i + 1, len(ins)))      # body twice: the back-edge is taken once
    fifo = []            # in-flight loads, oldest first: (dest regs, text)
    hazards = []
    for k in order:
        a, t = ins[k]
        op, ops = operands(t)
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if len(fifo) > n:
                    fifo = fifo[len(fifo) - n:]
            continue
        busy = {r: txt for regs, txt in fifo for r in regs}
        if busy:
            touched = [r for tok in ops for r in vregs(tok)]
            if op.startswith('global_load') and 'lds' not in op:
                touched = [r for tok in ops for r in vregs(tok)]      # destination included: a second load into a register still in flight is a hazard too
            bad = sorted(set(r for r in touched if r in busy))
            if bad:
                hazards.append((a, t, ['v%d <- %s' % (r, busy[r]) for r in bad]))
        if (op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load')) and 'lds' not in op:
            fifo.append((vregs(ops[0]), t))
        elif op.startswith('global_load') or op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('flat_store'):
            fifo.append(([], t))             # stores / LDS-DMA loads occupy a vmcnt slot, no destination register
    return hazards


def main(argv):
    obj, pats = argv[1], argv[2:]
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        dev = device_elf(obj, tmp)
        for name in kernels(dev):
            if pats and not any(p in name for p in pats):
                continue
            hz = check(disasm(dev, name))
            print('%-70s %s' % (name[:70], 'ok' if not hz else '%d hazard(s)' % len(hz)))
            for a, t, why in hz[:12]:
                print('    %x: %s   [%s]' % (a, t, '; '.join(why)))
            rc |= 1 if hz else 0
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv))
