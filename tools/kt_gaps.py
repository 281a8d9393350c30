#!/usr/bin/env python3
"""Timeline of the env step from a rocprofv3 --kernel-trace rocpd database: per position in the step (launch 1 .. 5)
the kernel's duration and the gap to the launch before it (start - previous end), medians over all replayed steps.
Usage: python tools/kt_gaps.py gpurun_out/x/..._results.db"""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
    names = np.array([("post" if "k_post_physics" in n else "lstm" if "k_lstm" in n else "pd" if "k_actuator_pd" in n
                       else "other") for n, _, _ in rows])
    st = np.array([r[1] for r in rows], dtype=np.float64) / 1e3
    en = np.array([r[2] for r in rows], dtype=np.float64) / 1e3
    # steps = runs of [act, act, act, act, post] with nothing else in between
    seq = {}
    i, n = 0, len(rows)
    while i + 5 <= n:
        if all(names[i + k] in ("lstm", "pd") for k in range(4)) and names[i + 4] == "post":
            prev_end = en[i - 1] if i > 0 else np.nan
            for k in range(5):
                seq.setdefault(k, []).append((en[i + k] - st[i + k], st[i + k] - (en[i + k - 1] if k else prev_end)))
            i += 5
        else:
            i += 1
    if not seq:
        print("no five-launch steps found")
        return
    print(f"{len(seq[0])} steps")
    tot = 0.0
    for k in range(5):
        a = np.array(seq[k])
        d, g = np.nanmedian(a[:, 0]), np.nanmedian(a[:, 1])
        tot += d + g
        print(f"launch {k + 1}: kernel median {d:6.2f} us (p10 {np.nanpercentile(a[:, 0], 10):6.2f}, p90 {np.nanpercentile(a[:, 0], 90):6.2f})"
              f"   gap in front of it median {g:5.2f} us (p90 {np.nanpercentile(a[:, 1], 90):5.2f})")
    print(f"sum of medians {tot:.2f} us per step")


if __name__ == "__main__":
    main(sys.argv[1])
