#!/usr/bin/env python3
"""Per-queue timeline of the last meta-step in a rocprofv3 --kernel-trace rocpd db."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute('select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start').fetchall()
fin = [i for i, r in enumerate(rows) if 'k_finalize' in r[0]]
step = rows[fin[-2] + 1:fin[-1] + 1]
t0 = step[0][1]
print('kernels', len(step), 'wall us %.1f' % ((step[-1][2] - t0) / 1e3))
for q in sorted(set(r[3] for r in step)):
    ks = [r for r in step if r[3] == q]
    print('queue', q, 'n', len(ks), 'busy us %.1f' % (sum(r[2] - r[1] for r in ks) / 1e3), 'first %.1f last %.1f' % ((ks[0][1] - t0) / 1e3, (ks[-1][2] - t0) / 1e3))
main = [r for r in step if r[3] == step[0][3]]
agg = {}
for r in main:
    nm = re.sub(r'\(.*$', '', r[0]).replace('void ', '')[:34]
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
print('--- main queue totals')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%-36s n %4d  total %8.1f us  avg %7.1f' % (k, v[0], v[1], v[1] / v[0]))
gaps = sum(max(0, main[i + 1][1] - main[i][2]) for i in range(len(main) - 1)) / 1e3
print('main queue idle gaps total %.1f us' % gaps)
prev = None
for r in main[:n]:
    nm = re.sub(r'\(.*$', '', r[0]).replace('void ', '')[:40]
    print('%-42s start %8.1f dur %7.1f gap %6.1f' % (nm, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev) / 1e3 if prev else 0))
    prev = r[2]
