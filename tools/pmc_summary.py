#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, with
--kernel-trace only), corrected as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read stream, so it is doubled (WRITE_SIZE checked against AdamW: 12 B/param).
usage: tools/pmc_summary.py <fetch.db> <write.db> [out.json]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    return {r[0]: (r[1], r[2]) for r in cur.execute(
        "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


f = per_kernel(sys.argv[1], 'FETCH_SIZE')
w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
print('%-90s %7s %14s %14s %14s' % ('kernel', 'calls', 'read_MB(x2)', 'write_MB', 'total_MB/launch'))
for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, (0, 0))[1]) * f[k][0]):
    rd = f[k][1] * 2 * 1024.0
    wr = w.get(k, (0, 0.0))[1] * 1024.0
    out[k] = {'launches': f[k][0], 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr}
    print('%-90s %7d %14.2f %14.2f %14.2f' % (k[:90], f[k][0], rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
if len(sys.argv) > 3:
    from bench import csrc_fingerprint          # the kernels these counters belong to: bench.py refuses the figures once csrc/ has changed
    out['_meta'] = {'csrc_fingerprint': csrc_fingerprint(), 'counters': 'FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes, separate --pmc passes with --kernel-trace only'}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
