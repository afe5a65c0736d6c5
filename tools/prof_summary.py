#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) per kernel: calls, total, avg, min, max, %.
    python tools/prof_summary.py gpurun_out/prof_x/x_results.db [> profiles/rNN_name.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, d.group_segment_size, '
                      'd.workgroup_size_x, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id').fetchall()
    agg = {}
    for name, st, en, vg, ag, sg, lds, wg, grid in rows:
        a = agg.setdefault(short(name), {'n': 0, 't': 0, 'mn': 1e30, 'mx': 0, 'vg': vg, 'ag': ag, 'sg': sg, 'lds': lds, 'wg': wg})
        d = en - st
        a['n'] += 1; a['t'] += d; a['mn'] = min(a['mn'], d); a['mx'] = max(a['mx'], d); a['lds'] = max(a['lds'], lds)
    tot = sum(a['t'] for a in agg.values())
    print('%-92s %7s %12s %10s %10s %10s %6s  %s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%', 'vgpr/agpr/sgpr lds wg'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['t']):
        print('%-92s %7d %12.3f %10.1f %10.1f %10.1f %6.2f  %d/%d/%d %d %d' % (k, a['n'], a['t'] / 1e6, a['t'] / a['n'] / 1e3, a['mn'] / 1e3,
                                                                          a['mx'] / 1e3, 100.0 * a['t'] / tot, a['vg'], a['ag'], a['sg'], a['lds'], a['wg']))
    print('TOTAL kernel time %.3f ms over %d dispatches' % (tot / 1e6, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1])
