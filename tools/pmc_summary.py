#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters (counter_collection.csv).
    python tools/pmc_summary.py gpurun_out/pmc_x/x_counter_collection.csv"""
import collections
import csv
import re
import sys


def main(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', row['Kernel_Name'])[:60]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[k].add(row['Dispatch_Id'])
    names = sorted({c for v in acc.values() for c in v})
    print('%-62s %6s ' % ('kernel', 'disp') + ' '.join('%22s' % n for n in names))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        n = len(cnt[k])
        print('%-62s %6d ' % (k, n) + ' '.join('%22.4g' % (v.get(c, 0) / n) for c in names))


if __name__ == '__main__':
    main(sys.argv[1])
