"""Per-kernel statistics (the --stats table) from a rocprofv3 rocpd database:
   python scripts/rocpd_stats.py results.db [out.csv]
Same columns as rocprofv3's kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev."""
import csv
import math
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = {}
for n, s, e in db.execute(f"select {name}, start, end from kernels"):
    rows.setdefault(n, []).append(e - s)
total = sum(sum(v) for v in rows.values()) or 1
out = []
for n, v in rows.items():
    avg = sum(v) / len(v)
    sd = math.sqrt(sum((x - avg) ** 2 for x in v) / (len(v) - 1)) if len(v) > 1 else 0.0
    out.append((n, len(v), sum(v), round(avg, 3), round(100.0 * sum(v) / total, 4), min(v), max(v), round(sd, 3)))
out.sort(key=lambda r: -r[2])
f = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
w.writerows(out)
