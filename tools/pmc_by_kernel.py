#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel name:
    python tools/pmc_by_kernel.py counter_collection.csv [out.csv]
prints / writes launches, counter name, total and per-launch mean of the counter value."""
import csv
import re
import sys
from collections import defaultdict


def demangle_local(name):
    """`_ZN12_GLOBAL__N_1<len><kernel>I<L.. template literals>E...` -> `kernel<3,3,1>`: rocprofv3 leaves the
    names of kernels with __bf16 parameters mangled (its demangler does not know the DF16b type code)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while True:
            a = re.match(r"L([ibjlm])(n?\d+)E", rest)
            if not a:
                break
            v = a.group(2).replace("n", "-")
            args.append(("true" if v != "0" else "false") if a.group(1) == "b" else v)
            rest = rest[a.end():]
    return base + ("<" + ", ".join(args) + ">" if args else "")


def short(name):
    name = demangle_local(name)
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


def fix_names(path):
    """Rewrite the kernel column of an aggregated or rocprofv3 statistics CSV with demangled names."""
    rows = list(csv.reader(open(path)))
    for r in rows[1:]:
        r[0] = demangle_local(r[0]) if r[0].startswith("_Z") else r[0]
    with open(path, "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_MINIMAL).writerows(rows)


if __name__ == "__main__":
    if sys.argv[1] == "--fix-names":
        for p in sys.argv[2:]:
            fix_names(p)
    else:
        main(*sys.argv[1:3])
