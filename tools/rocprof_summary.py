#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a per-kernel table.
usage: tools/rocprof_summary.py <results.db> [steps_including_warmup] > profiles/<name>.txt"""
import sqlite3
import sys

db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('# rocprofv3 --kernel-trace --stats summary of %s' % db)
print('# total kernel time %.3f ms over %d step(s) (warm-up included) = %.3f ms/step' % (tot / 1e6, steps, tot / 1e6 / steps))
print('%-100s %7s %11s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
for n, c, s, a, mn, mx in rows:
    print('%-100s %7d %11.3f %10.1f %10.1f %10.1f %6.2f' % (n[:100], c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
