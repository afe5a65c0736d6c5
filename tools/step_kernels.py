#!/usr/bin/env python3
"""Every kernel of the last meta-step in a rocprofv3 --kernel-trace rocpd db: queue, start, duration, grid, name; then per-(name, grid) totals."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start').fetchall()
fin = [i for i, r in enumerate(rows) if 'k_finalize' in r[0]]
step = rows[fin[-2] + 1:fin[-1] + 1]
t0 = step[0][1]
qs = sorted(set(r[3] for r in step))
print('kernels', len(step), 'wall us %.1f' % ((max(r[2] for r in step) - t0) / 1e3))
agg = {}
for r in step:
    nm = re.sub(r'\(.*$', '', r[0]).replace('void ', '')[:44]
    if len(sys.argv) > 2: print('q%d %-46s blocks %6d start %8.1f dur %7.1f' % (qs.index(r[3]), nm, r[4] // max(1, r[5]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
    a = agg.setdefault((qs.index(r[3]), nm, r[4] // max(1, r[5])), [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
print('--- per (queue, kernel, blocks)')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('q%d %-46s blocks %6d n %4d total %8.1f us avg %7.1f' % (k[0], k[1], k[2], v[0], v[1], v[1] / v[0]))
for q in range(len(qs)):
    ks = [r for r in step if qs.index(r[3]) == q]
    print('queue', q, 'n', len(ks), 'busy us %.1f' % (sum(r[2] - r[1] for r in ks) / 1e3), 'span %.1f..%.1f' % ((ks[0][1] - t0) / 1e3, (ks[-1][2] - t0) / 1e3))
