#!/usr/bin/env python3
"""Per-kernel average of PMC counters from a rocprofv3 rocpd database (counter-collection pass).
Usage: python tools/rocprof_pmc.py gpurun_out/pmc/x_results.db"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {kname}, counter_name, dispatch_id, sum(value) from counters_collection "
                     f"group by {kname}, counter_name, dispatch_id").fetchall()
    acc = defaultdict(lambda: defaultdict(list))
    for k, cn, _, v in rows:
        acc[k][cn].append(v)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        short = k if len(k) <= 70 else k[:67] + "..."
        parts = [f"{cn}={sum(v) / len(v):.4g}" for cn, v in sorted(d.items())]
        n = len(next(iter(d.values())))
        print(f"{short:<70} n={n:<5} " + "  ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1])
