"""Dumps rocprofv3's kernel summary (top_kernels view of the results db) as CSV: name,calls,total_us,avg_us,pct."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("kernel,calls,total_us,avg_us,pct\n")
for r in rows:
    out.write('"%s",%d,%.3f,%.3f,%.2f\n' % (r[0].replace("mi355x::", ""), r[1], r[2], r[3], r[4]))
