"""Per-kernel totals of a rocprofv3 results database (--kernel-trace): python scripts/db_kernels.py results.db [name-substring-for-period]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), max(d.end-d.start) from rocpd_kernel_dispatch d "
                  "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
for r in rows[:16]:
    print(f"{r[0][:72]:72s} n={r[1]:6d} total={r[2]/1e6:9.2f} ms avg={r[3]/1e3:9.1f} us max={r[4]/1e3:9.1f}")
if len(sys.argv) > 2:
    l = db.execute("select d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
                   "where s.kernel_name like ? order by d.start", (f"%{sys.argv[2]}%",)).fetchall()
    k = min(100, len(l))
    print(f"{sys.argv[2]}: last {k} launches avg {sum(e - s for s, e in l[-k:]) / k / 1e3:.1f} us, period {(l[-1][0] - l[-k][0]) / max(1, k - 1) / 1e3:.1f} us")
