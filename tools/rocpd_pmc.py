"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database (kernel-trace + --pmc run): CSV on stdout."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
counter = sys.argv[2]
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
q = ('select s.%s, count(*), sum(e.value), avg(e.value), avg(d.end - d.start) from %s e join %s i on e.pmc_id = i.id '
     'join %s d on e.event_id = d.%s join %s s on d.kernel_id = s.id where i.name = ? group by s.%s order by 3 desc'
     % (name_col, pe, ip, kd, evcol, ks, name_col))
print('kernel,dispatches,%s_sum,%s_avg_per_dispatch,avg_ns' % (counter, counter))
for r in cur.execute(q, (counter,)):
    print('"%s",%d,%.1f,%.1f,%.0f' % (r[0][:150], r[1], r[2], r[3], r[4]))
