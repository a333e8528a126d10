#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite file: per-kernel time statistics and (if present) PMC counters."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-70s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for r in rows:
        print("%-70s %6d %12d %12.0f %12d %12d %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if ccols:
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            rows = cur.execute("select %s, counter_name, count(*), avg(value) from counters_collection "
                               "group by 1, 2 order by 1, 2" % kn).fetchall()
            if rows:
                print("\n%-50s %-28s %6s %18s" % ("kernel", "counter", "n", "avg value per dispatch"))
                for r in rows:
                    print("%-50s %-28s %6d %18.1f" % (r[0][:50], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        print("no counters:", e)


if __name__ == "__main__":
    main(sys.argv[1])
