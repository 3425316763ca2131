#!/usr/bin/env python3
"""Per-kernel calls / total / average from a rocprofv3 results database (the `--stats` summary as a table).
usage: kstats.py <dir-or-db> [min share %]"""
import glob
import os
import re
import sqlite3
import sys

path = sys.argv[1]
db = path if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[-1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
minshare = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
print("%-100s %6s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "min us", "max us", "%"))
for name, n, t, mn, mx in rows:
    if 100.0 * t / tot < minshare:
        continue
    nm = re.sub(r"^void ", "", name)
    nm = re.sub(r"\(.*$", "", nm)
    print("%-100s %6d %10.3f %9.1f %9.1f %9.1f %6.2f" % (nm[:100], n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print("total %.3f ms over %d launches" % (tot / 1e6, sum(r[1] for r in rows)))
