"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd database: one row per kernel, one column per counter.
    python tools/rocpd_pmc_table.py <dir-or-db> [name-filter]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
db = sqlite3.connect(path)
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
ip = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
kcols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
name_col = 'display_name' if 'display_name' in scol else 'kernel_name'
evcol = 'event_id' if 'event_id' in kcols else 'id'
q = ('select s.%s, i.name, count(*), avg(e.value), avg(d.end - d.start), d.grid_size_x, d.workgroup_size_x from %s e join %s i on e.pmc_id = i.id '
     'join %s d on e.event_id = d.%s join %s s on d.kernel_id = s.id group by s.%s, i.name, d.grid_size_x'
     % (name_col, pe, ip, kd, evcol, ks, name_col))
rows = {}
names = []
for kname, cname, n, avg, ns, gx, wx in cur.execute(q):
    if flt and flt not in kname:
        continue
    key = (kname[:90], gx, wx)
    rows.setdefault(key, {'n': n, 'ns': ns})[cname] = avg
    if cname not in names:
        names.append(cname)
names.sort()
print('kernel | grid | wg | n | avg_us | ' + ' | '.join(names))
for (k, gx, wx), v in sorted(rows.items(), key=lambda kv: -kv[1]['ns']):
    print('%s | %d | %d | %d | %.1f | ' % (k, gx, wx, v['n'], v['ns'] / 1e3) + ' | '.join('%.4g' % v.get(c, float('nan')) for c in names))
