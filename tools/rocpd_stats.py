"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (CSV on stdout)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
name_col = 'display_name' if 'display_name' in scol else 'kernel_name'
q = 'select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc' % (name_col, kd, ks, name_col)
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print('name,calls,total_ns,avg_ns,min_ns,max_ns,pct')
for r in rows:
    print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (r[0][:160], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
