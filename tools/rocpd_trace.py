#!/usr/bin/env python3
"""Print the gr80211 kernels of a rocprofv3 rocpd sqlite file in start order with start/end relative to the
first one (ms) -- shows which launches overlapped."""
import sqlite3
import sys


def main(path, last=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    extra = [c for c in ("queue_id", "stream_id") if c in cols]
    rows = cur.execute("select %s, start, end %s from kernels where %s like '%%gr80211%%' order by start"
                       % (name_col, "".join(", " + c for c in extra), name_col)).fetchall()
    rows = rows[-last:]
    t0 = rows[0][1]
    for r in rows:
        print("%-28s %10.3f %10.3f %8.3f  %s" % (r[0].split("(")[0][-28:], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6,
                                                 (r[2] - r[1]) / 1e6, " ".join(str(x) for x in r[3:])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
