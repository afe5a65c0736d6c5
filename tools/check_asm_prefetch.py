#!/usr/bin/env python3
"""Sanity check for kernels that prefetch with inline-asm global loads and hand-counted s_waitcnt vmcnt (gemm_split.h feeders, k_wgrad_split):
walks a kernel's ISA linearly (every basic block in layout order, loop bodies twice is NOT modelled) and reports instructions that READ or
OVERWRITE a VGPR while an asm load into it is still outstanding -- i.e. a compiler-inserted copy / reuse between the asm issue and the asm wait.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/gemm.s g-meta_amd/csrc/gemm.hip
    python tools/check_asm_prefetch.py /tmp/gemm.s _Z13k_wgrad_splitILi2ELi2EEv6WgradK"""
import re
import sys


def regs_of(text):
    out = set(int(x) for x in re.findall(r'\bv(\d+)\b', text))
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', text):
        out |= set(range(int(a), int(b) + 1))
    return out


def main(path, kernel):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ':'))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    outstanding = []          # dst regs of asm loads, oldest first
    in_asm = False
    bad = 0
    for i in range(start, end):
        t = lines[i].strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True; continue
        if t.startswith(';;#ASMEND'):
            in_asm = False; continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        if in_asm:
            m = re.match(r'global_load_dword(?:x(\d))? (v\d+|v\[\d+:\d+\]),', t)
            if m:
                outstanding.append(regs_of(m.group(2)))
                continue
            m = re.search(r's_waitcnt vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                outstanding = outstanding[len(outstanding) - n:] if n and len(outstanding) > n else ([] if not n else outstanding)
            continue
        m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', t)        # compiler waits count the asm loads too
        if m:
            n = int(m.group(1))
            outstanding = outstanding[len(outstanding) - n:] if n and len(outstanding) > n else ([] if not n else outstanding)
            continue
        live = set().union(*outstanding) if outstanding else set()
        hit = regs_of(t) & live
        if hit:
            bad += 1
            if bad <= 20:
                print('line %d: %s   <- v%s has an asm load in flight' % (i + 1, t[:90], sorted(hit)))
    print('%s: %d instruction(s) touch a register with an asm load in flight' % (kernel, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1], sys.argv[2]))
