#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc run (rocpd sqlite).  usage: tools/pmc_kernel_table.py <db> [name filter]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
tab = {}
for k, c, n, v in rows:
    if flt in k:
        tab.setdefault(k, {})[c] = (n, v)
for k, d in tab.items():
    print(k[:110])
    for c in sorted(d):
        print('    %-32s %6d launches  avg %.4g' % (c, d[c][0], d[c][1]))
