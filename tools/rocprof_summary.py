#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max duration) of a rocprofv3 rocpd database.
Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db [> profiles/rNN_name.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<78} {'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for n, calls, tot, avg, mn, mx in rows:
        short = n if len(n) <= 78 else n[:75] + "..."
        print(f"{short:<78} {calls:>7} {tot / 1e3:>11.1f} {avg / 1e3:>9.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100 * tot / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
