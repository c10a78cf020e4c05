"""Per-kernel summary (calls, total, average, share, min, max) from a rocprofv3 rocpd database (`kernels` view).
  python tools/rocpd_kernel_stats.py <dir-or-db> "<header comment>" > profiles/<name>.csv"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
rows = {}
for db in dbs:
    con = sqlite3.connect(db)
    for name, start, end in con.execute("SELECT name, start, end FROM kernels"):
        rows.setdefault(name, []).append(end - start)
total = sum(sum(v) for v in rows.values()) or 1
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%d,%d,%.2f,%d,%d' % (name, len(v), sum(v), sum(v) // len(v), 100.0 * sum(v) / total, min(v), max(v)))
