"""Developer tool: the kernel sequence of the last training step of a rocprofv3 --kernel-trace database, with run-length compression --
which launches surround the small copy / fill kernels.  usage: step_sequence.py <results.db> [name substring to centre on]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
names = [r[0] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('adamw_kernel')]
lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(names))
seq = rows[lo:hi]
out, i = [], 0
while i < len(seq):
    j = i
    while j + 1 < len(seq) and seq[j + 1][0] == seq[i][0]:
        j += 1
    out.append((seq[i][0][:70], j - i + 1, sum(e - s for _, s, e in seq[i:j + 1]) / 1e3))
    i = j + 1
want = sys.argv[2] if len(sys.argv) > 2 else 'copyBuffer'
for k, (n, c, us) in enumerate(out):
    if want in n:
        a = out[k - 1][0] if k else ''
        b = out[k + 1][0] if k + 1 < len(out) else ''
        print('%3d x %-30s %7.1f us   after %-50s before %s' % (c, n[:30], us, a[:50], b[:50]))
print('launches in the step:', len(seq))
