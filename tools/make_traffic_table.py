#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from the per-kernel FETCH_SIZE / WRITE_SIZE summaries written by
tools/pmc_by_kernel.py:  python tools/make_traffic_table.py fetch.csv write.csv workload_key"""
import csv
import json
import os
import sys


def load(path):
    return {r["Kernel"].replace(", ", ","): (int(r["Launches"]), float(r["PerLaunch"]))
            for r in csv.DictReader(open(path))}


def main(fetch_csv, write_csv, key):
    f, w = load(fetch_csv), load(write_csv)
    out = {}
    for k in f:
        if k in w:
            out[k] = {"launches_per_step": f[k][0], "fetch_size_kib_per_launch": f[k][1],
                      "write_size_kib_per_launch": w[k][1],
                      "hbm_bytes_per_launch": round((2 * f[k][1] + w[k][1]) * 1024)}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    table = {}
    if os.path.exists(path):
        table = json.load(open(path))
    table["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over one step of `python bench.py "
                     "--steps 1 --warmup 0`; FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B for "
                     "16 B/lane streams); KiB units")
    table[key] = out
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
