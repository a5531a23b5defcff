"""Timeline of ONE forward step from a rocprofv3 kernel-trace rocpd database: the dispatches between the last two launches
of the step's first kernel (pack_image), in start order, with duration, gap to the previous end and queue.
    python tools/rocpd_timeline.py <dir-or-db> [first-kernel-substring]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[-1]
first = sys.argv[2] if len(sys.argv) > 2 else "stem_pool_kernel"
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # launches of that kernel per step
db = sqlite3.connect(path)
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
kcols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
name_col = 'display_name' if 'display_name' in scol else 'kernel_name'
qcol = 'queue_id' if 'queue_id' in kcols else ('stream_id' if 'stream_id' in kcols else 'tid')
rows = list(cur.execute('select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, qcol, kd, ks)))
idx = [i for i, r in enumerate(rows) if first in r[0]]
a, b = idx[-2 * per_step], idx[-per_step]
step = rows[a:b]
t0 = step[0][1]
prev_end = t0
busy = 0
print('# step: %d dispatches, wall %.1f us' % (len(step), (rows[b][1] - t0) / 1e3))
print('%8s %8s %7s  q  kernel' % ('start_us', 'dur_us', 'gap_us'))
for name, s, e, q in step:
    short = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:70]
    print('%8.1f %8.1f %7.1f  %s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short))
    prev_end = max(prev_end, e)
