"""profiles/kstats.py DB [pattern...]: per-kernel totals from a rocprofv3 result database (rocpd sqlite), and with
patterns the duration of every launch of the matching kernels in launch order."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
if len(sys.argv) == 2:
    q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1000.0, avg(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
    for r in c.execute(q).fetchall()[:30]:
        print("%-64s calls %4d total %9.1f us avg %8.1f us" % (r[0][:64], r[1], r[2], r[3]))
for pat in sys.argv[2:]:
    rows = c.execute(f"select (d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' order by d.start").fetchall()
    print(pat, " ".join("%.0f" % r[0] for r in rows))
