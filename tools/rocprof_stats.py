#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel calls / total / avg / min, like --stats.
usage: python tools/rocprof_stats.py <results.db> [--last-ms X]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:7d} {r[2] / 1e3:11.1f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / tot:6.1f}")
    ks = list(cur.execute("select name,start,end from kernels order by start"))
    if len(ks) > 300:
        last = ks[-258:]
        gaps = sorted((last[i + 1][1] - last[i][2]) / 1e3 for i in range(len(last) - 1))
        print(f"last 258 dispatches: span {(last[-1][2] - last[0][1]) / 1e3:.1f} us, sum of kernel durations {sum(r[2] - r[1] for r in last) / 1e3:.1f} us, median gap {gaps[len(gaps) // 2]:.2f} us")


if __name__ == "__main__":
    main()
