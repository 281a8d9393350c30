#!/usr/bin/env python3
"""Merges rocprofv3 --pmc passes (one counter family per pass, as the pool requires) into profiles/pmc_traffic.json:
per-kernel average counter value per dispatch.
    python tools/pmc_traffic.py gpurun_out/pmcF/f_results.db gpurun_out/pmcW/w_results.db [-o profiles/pmc_traffic.json]"""
import json
import sqlite3
import sys
from collections import defaultdict


def load(path, acc):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kname = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {kname}, counter_name, dispatch_id, sum(value) from counters_collection "
                     f"group by {kname}, counter_name, dispatch_id").fetchall()
    per = defaultdict(lambda: defaultdict(list))
    for k, cn, _, v in rows:
        per[k][cn].append(v)
    for k, d in per.items():
        for cn, vals in d.items():
            acc[k][cn] = sum(vals) / len(vals)
            acc[k]["dispatches_" + cn] = len(vals)


def main(argv):
    out = "profiles/pmc_traffic.json"
    paths = []
    it = iter(argv)
    for a in it:
        if a == "-o":
            out = next(it)
        else:
            paths.append(a)
    acc = defaultdict(dict)
    for p in paths:
        load(p, acc)
    doc = {"unit": "counter value per dispatch (FETCH_SIZE / WRITE_SIZE in KiB)", "sources": paths,
           "kernels": {k: v for k, v in sorted(acc.items())}}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in doc["kernels"].items():
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            print(f"{k[:70]:<70} fetch {v.get('FETCH_SIZE', float('nan')):10.1f} KiB  write {v.get('WRITE_SIZE', float('nan')):10.1f} KiB")


if __name__ == "__main__":
    main(sys.argv[1:])
