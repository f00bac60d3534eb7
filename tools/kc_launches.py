#!/usr/bin/env python3
"""Per-launch durations of the GEMM kernels in ONE training step of a rocprofv3 kernel trace (rocpd sqlite): which shapes of the step are
slower than the stand-alone numbers of tools/gemm_bench.  usage: tools/kc_launches.py <results.db> [steps_in_trace]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
q = "select name, start, end%s from kernels order by start" % ((', ' + gx) if gx else '')
rows = list(cur.execute(q))
per = len(rows) // steps
last = rows[-per:]
print('# %d launches per step; columns: %s' % (per, cols))
for r in last:
    if 'gemm' in r[0]:
        short = r[0].split('(')[0].replace('void ', '')[:60]
        print('%-62s grid %6s  %8.1f us' % (short, r[3] if gx else '?', (r[2] - r[1]) / 1e3))
