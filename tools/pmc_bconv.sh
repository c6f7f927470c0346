#!/bin/bash
# Counter passes over tools/bench_bconv.py (one rocprofv3 run per pass): bash tools/pmc_bconv.sh <out dir>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/${1:-pmc_bconv}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    echo "pass $i: $pass"
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/p$i" -- python3 "$root/tools/bench_bconv.py" --reps 1 > /dev/null 2> "$out/p$i.log"
    python3 "$root/tools/pmc_by_kernel.py" "$out"/p$i/*/*_counter_collection.csv "$out/p$i.csv" | grep bconv || true
    rm -rf "$out/p$i"
done
