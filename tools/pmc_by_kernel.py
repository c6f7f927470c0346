#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel name:
    python tools/pmc_by_kernel.py counter_collection.csv [out.csv]
prints / writes launches, counter name, total and per-launch mean of the counter value."""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(path, out=None):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), r["Counter_Name"])
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = [["Kernel", "Counter", "Launches", "Total", "PerLaunch"]]
    for (k, c), (n, tot) in rows:
        lines.append([k, c, n, "%.6g" % tot, "%.6g" % (tot / n)])
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(lines)
    for ln in lines[:25]:
        print("%-44s %-12s %6s %14s %14s" % tuple(ln))


if __name__ == "__main__":
    main(*sys.argv[1:3])
