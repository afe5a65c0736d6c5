#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters, from a counter_collection.csv or a rocpd .db.
    python tools/pmc_summary.py <csv|db> [<csv|db> ...]          # per-kernel table of every counter found
    python tools/pmc_summary.py --agg-traffic <fetch> <write>    # profiles/agg_traffic.json (aggregate kernels only)"""
import collections
import csv
import json
import re
import sqlite3
import sys


def rows_of(path):
    if path.endswith('.db'):
        db = sqlite3.connect(path)
        for name, disp, cn, cv in db.execute('select name, dispatch_id, counter_name, counter_value from pmc_events'):
            yield name, disp, cn, float(cv)
    else:
        for row in csv.DictReader(open(path)):
            yield row['Kernel_Name'], row['Dispatch_Id'], row['Counter_Name'], float(row['Counter_Value'])


def collect(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for p in paths:
        for name, disp, cn, cv in rows_of(p):
            k = re.sub(r'\(.*$', '', name)[:60]
            acc[k][cn] += cv
            cnt[k][cn].add(disp)
    return acc, cnt


def main(argv):
    if argv and argv[0] == '--agg-traffic':
        acc, cnt = collect(argv[1:3])
        taken = argv[3] if len(argv) > 3 else 'unlabelled'
        tot = collections.defaultdict(float); launches = 0
        for k in acc:
            if 'k_agg' in k:
                for c, v in acc[k].items():
                    tot[c] += v
                if 'k_agg_win' in k or 'k_agg<' in k or 'k_agg_stream' in k:
                    launches += len(cnt[k].get('FETCH_SIZE', ()))
        fetch_raw = tot['FETCH_SIZE'] * 1024 / launches; write = tot['WRITE_SIZE'] * 1024 / launches
        json.dump({'hbm_bytes_per_launch': int(2 * fetch_raw + write), 'fetch_raw_bytes_per_launch': int(fetch_raw), 'write_bytes_per_launch': int(write),
                   'launches': launches, 'taken': taken,
                   'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the aggregate kernels (k_agg_win and, since round 5, k_agg_stream; hub rows ride in both) of '
                           'bench.py --serialize 1 --steps 1 --warmup 1; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests '
                           'at 64 B; Infinity-Cache hits are included, so this is fabric-side traffic, an upper bound on HBM bytes); KB -> bytes x1024'},
                  sys.stdout, indent=1)
        print()
        return
    acc, cnt = collect(argv)
    names = sorted({c for v in acc.values() for c in v})
    print('%-62s %6s ' % ('kernel', 'disp') + ' '.join('%22s' % n for n in names))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        n = max(len(s) for s in cnt[k].values())
        print('%-62s %6d ' % (k, n) + ' '.join('%22.4g' % (v.get(c, 0) / max(len(cnt[k].get(c, ())), 1)) for c in names))


if __name__ == '__main__':
    main(sys.argv[1:])
