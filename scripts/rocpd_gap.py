#!/usr/bin/env python3
"""Find the largest hole in a rocprofv3 rocpd trace (kernel dispatch gaps or over-long kernels) and list every
record of every table/view with start/end columns that overlaps it.  usage: rocpd_gap.py <db> [min_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("tables/views:", ", ".join(names))
rows = db.execute("select name, start, end from kernels order by start").fetchall()
print("kernel dispatches:", len(rows))
t0 = rows[0][1]
holes = []
for i, (n, s, e) in enumerate(rows):
    if (e - s) / 1e6 > min_ms:
        holes.append((s, e, f"kernel {n[:60]} ran {(e - s) / 1e6:.1f} ms"))
    if i and (s - rows[i - 1][2]) / 1e6 > min_ms:
        holes.append((rows[i - 1][2], s, f"no kernel running for {(s - rows[i - 1][2]) / 1e6:.1f} ms after {rows[i - 1][0][:40]} before {n[:40]}"))
for (a, b, what) in holes:
    print(f"\n=== hole at +{(a - t0) / 1e6:.1f} ms .. +{(b - t0) / 1e6:.1f} ms: {what}")
    for t in names:
        try:
            cols = [c[1] for c in db.execute(f"pragma table_info('{t}')")]
        except sqlite3.Error:
            continue
        if "start" not in cols or "end" not in cols or t == "kernels":
            continue
        label = "name" if "name" in cols else cols[0]
        try:
            rs = db.execute(f"select {label}, start, end from '{t}' where end >= ? and start <= ? order by start limit 40", (a - 2000000, b + 2000000)).fetchall()
        except sqlite3.Error as ex:
            print("  ", t, "query failed", ex)
            continue
        for r in rs:
            print(f"   [{t}] {str(r[0])[:70]}  +{(r[1] - t0) / 1e6:.2f} .. +{(r[2] - t0) / 1e6:.2f} ms ({(r[2] - r[1]) / 1e6:.2f} ms)")
