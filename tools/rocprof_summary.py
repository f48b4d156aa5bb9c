"""Summarise a rocprofv3 --kernel-trace database (rocpd sqlite) per kernel:
count, avg/min/max duration (us), share of GPU kernel time.  usage: rocprof_summary.py in.db out.csv"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                 "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
                 "from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows) or 1
with open(out, "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,total_ms,percent,vgpr,agpr,sgpr,lds_bytes,workgroup,grid\n")
    for r in rows:
        f.write('"%s",%d,%.2f,%.2f,%.2f,%.3f,%.2f,%s,%s,%s,%s,%s,%s\n' % (
            r[0].replace('"', "'"), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6, 100.0 * r[5] / tot,
            r[6], r[7], r[8], r[9], r[10], r[11]))
print("wrote", out)
