#!/bin/bash
# Copy what tools/refresh_profiles.sh left under gpurun_out/<tag>/ into profiles/ (tracked) as <tag>_*:
#   bash tools/install_profiles.sh <tag> [traffic-table key]
# and rebuild the digest (tools/profile_digest.py) and, with a key, that workload's entry of profiles/pmc_traffic.json.
set -e
tag=$1
key=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gpurun_out/$tag
for f in bench.json bench_under_rocprof.json kernel_stats.csv pmc_fetch_size.csv pmc_write_size.csv; do
    cp "$src/$f" "$root/profiles/${tag}_$f"
done
cp "$src/pmc_sq_wave_cycles.csv" "$root/profiles/${tag}_pmc_sq.csv"
python3 "$root/tools/profile_digest.py" "$src" > "$root/profiles/${tag}_digest.json"
if [ -n "$key" ]; then
    python3 "$root/tools/make_traffic_table.py" "$src/pmc_fetch_size.csv" "$src/pmc_write_size.csv" "$key"
fi
echo "installed profiles/${tag}_*"
