#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2) rocpd SQLite result into the per-kernel stats
table that `--stats` prints: calls, total / average / min / max duration.

    python profiles/summarize_rocpd.py <results.db> > profiles/<name>.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    q = ("select %s, count(*), sum(end-start), avg(end-start), min(end-start), "
         "max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("%-72s %6s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us",
                                                "min_us", "max_us", "pct"))
    for name, calls, total, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "")
        short = short.split("(")[0][-72:]
        print("%-72s %6d %12.1f %12.2f %10.2f %10.2f %6.2f" %
              (short, calls, total / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot))


if __name__ == "__main__":
    main(sys.argv[1])
