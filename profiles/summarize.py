#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short per-kernel table (per step)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_]+)I", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"# {path}: total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.3f} ms/step")
    print(f"{'kernel':62s} {'calls/step':>10s} {'avg_us':>9s} {'ms/step':>8s} {'%':>6s}")
    for r in rows:
        t = int(r["TotalDurationNs"])
        if t / tot < 0.002:
            continue
        print(f"{short(r['Name']):62s} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f} "
              f"{t / 1e6 / steps:8.3f} {100.0 * t / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
