#!/usr/bin/env python
"""Summarise kernel dispatch durations of a rocprofv3 rocpd (sqlite) trace: one line per kernel
(name, calls, total/avg/min/max ns) -- the same columns as rocprofv3 --stats kernel_stats.csv."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, mn, mx in rows:
        print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}')


if __name__ == "__main__":
    main(sys.argv[1])
