#!/usr/bin/env python3
"""Join the files of one profile set (tools/refresh_profiles.sh) into a per-kernel digest:

    python tools/profile_digest.py gpurun_out/<tag> [top_n] > profiles/<tag>_digest.json

For the top kernels by time of the rocprofv3 kernel trace: calls, average duration, share of kernel time,
HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (KiB units; FETCH_SIZE doubled, the gfx950
correction for 16 B/lane streams) and the rate they imply, and the matrix-unit occupancy from the SQ pass:
SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) / (average duration x 2.4 GHz x 1024 SIMDs).
The launches of the counter passes are those of ONE step without warm-up, the trace covers warm-up and
timed steps: per-launch figures are comparable, totals are not.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_by_kernel import short  # noqa: E402

CLOCK_GHZ, SIMDS = 2.4, 1024


def per_kernel(path, counter=None):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        if counter is None or r["Counter"] == counter:
            out[short(r["Kernel"])] = (int(r["Launches"]), float(r["PerLaunch"]))
    return out


def main(folder, top=14):
    stats = list(csv.DictReader(open(os.path.join(folder, "kernel_stats.csv"))))
    total = sum(float(r["TotalDurationNs"]) for r in stats)
    fetch = per_kernel(os.path.join(folder, "pmc_fetch_size.csv"))
    write = per_kernel(os.path.join(folder, "pmc_write_size.csv"))
    sq = os.path.join(folder, "pmc_sq_wave_cycles.csv")
    mfma = per_kernel(sq, "SQ_VALU_MFMA_BUSY_CYCLES")
    insts_mfma = per_kernel(sq, "SQ_INSTS_MFMA")
    insts_valu = per_kernel(sq, "SQ_INSTS_VALU")
    kernels = []
    for r in stats[:int(top)]:
        name = short(r["Name"])
        avg_ns = float(r["AverageNs"])
        row = {"kernel": name, "calls": int(r["Calls"]), "avg_us": round(avg_ns / 1e3, 1),
               "share_of_kernel_time": round(float(r["TotalDurationNs"]) / total, 4)}
        if name in fetch and name in write:
            hbm = (2 * fetch[name][1] + write[name][1]) * 1024
            row["pmc_launches"] = fetch[name][0]
            row["hbm_bytes_per_launch"] = round(hbm)
            row["hbm_read_bytes_per_launch"] = round(2 * fetch[name][1] * 1024)
            row["hbm_write_bytes_per_launch"] = round(write[name][1] * 1024)
            row["hbm_GBps_at_avg_duration"] = round(hbm / avg_ns, 1)
        if name in mfma:
            row["mfma_busy_cycles_per_launch"] = round(mfma[name][1])
            row["mfma_busy_frac"] = round(mfma[name][1] / (avg_ns * CLOCK_GHZ * SIMDS), 4)
        if name in insts_mfma and name in insts_valu and insts_valu[name][1] > 0:
            row["valu_per_mfma_instruction"] = round(insts_valu[name][1] / max(insts_mfma[name][1], 1.0), 2)
        kernels.append(row)
    out = {"set": os.path.basename(os.path.normpath(folder)), "kernel_time_ms_in_trace": round(total / 1e6, 1),
           "mfma_busy_frac_definition": "SQ_VALU_MFMA_BUSY_CYCLES / (avg duration x %.1f GHz x %d SIMDs)" % (CLOCK_GHZ, SIMDS),
           "kernels": kernels}
    bench = os.path.join(folder, "bench.json")
    if os.path.exists(bench):
        try:
            b = json.load(open(bench))
            out["bench"] = {k: b.get(k) for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config")}
        except ValueError:
            pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
