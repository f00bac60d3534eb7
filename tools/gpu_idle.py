#!/usr/bin/env python3
"""GPU idle time inside the timed steps of a rocprofv3 kernel trace (rocpd sqlite): union of all kernel intervals (any stream) vs the
span from the first to the last kernel of the window.  usage: tools/gpu_idle.py <db> [skip_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
try:
    rows = sorted(db.execute('select start, end, name from kernels').fetchall())
except Exception:
    rows = sorted((a, b, '?') for a, b in db.execute('select start, end from %s' % kd).fetchall())
steps = [e for _, e, nm in rows if nm.startswith('adamw_kernel')]        # one optimiser launch per training step
if len(steps) >= 4 and skip >= 0:
    nwin = min(6, len(steps) - 2)                    # window: the last `nwin` whole steps (between the ends of two AdamW launches): no set-up, no warm-up / timed-loop sync
    lo, hi = steps[-1 - nwin], steps[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print('window: the last %d training steps (%.3f ms per step)' % (nwin, (hi - lo) / 1e6 / nwin))
else:
    rows = rows[int(len(rows) * abs(skip)):]         # no optimiser in the trace: drop the first part (set-up, warm-up)
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps, where, last = 0, rows[0][0], rows[0][1], [], [], rows[0][2]
for s, e, nm in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); where.append((s - cur_e, last, nm)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last = nm
busy += cur_e - cur_s
gaps.sort()
print('kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)  gaps %d  median gap %.2f us  p90 %.2f us  max %.1f us' %
      (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, len(gaps),
       gaps[len(gaps) // 2] / 1e3 if gaps else 0, gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0, gaps[-1] / 1e3 if gaps else 0))
big = [g for g in gaps if g > 20e3]
print('gaps > 20 us: %d, total %.3f ms' % (len(big), sum(big) / 1e6))

for g, a, b in sorted(where, reverse=True)[:12]:
    print('  %8.1f us  after %-60s before %s' % (g / 1e3, a[:60], b[:60]))
