#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel stats + PMC counters) as text for profiles/."""
import sqlite3
import sys


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["name,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for r in rows:
        out.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0*r[2]/tot:.2f}")
    return "\n".join(out)


def pmc(path):
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info(counters_collection)")]
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                      "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall() \
        if "kernel_name" in cols else []
    out = ["kernel,counter,dispatches,sum,avg_per_dispatch"]
    for r in rows:
        out.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.6g},{r[4]:.6g}")
    if not rows:
        out.append("# columns: " + ",".join(cols))
    return "\n".join(out)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    print(kernel_stats(path) if mode == "stats" else pmc(path))
