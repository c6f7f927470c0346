#!/bin/bash
# Re-measure the default bench workload on the GPU box and leave everything under gpurun_out/<tag>/:
#   bash tools/refresh_profiles.sh <tag>
# bench JSON (with the CPU baseline), rocprofv3 kernel-trace statistics of the same command, and the
# FETCH_SIZE / WRITE_SIZE / SQ counter passes (one rocprofv3 run each) aggregated per kernel.
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$root/bench.py" > "$out/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.log"
cp "$out"/trace/*/*_kernel_stats.csv "$out/kernel_stats.csv"
for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    name=$(echo "$pass" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$out/pmc_$name.log"
    python3 "$root/tools/pmc_by_kernel.py" "$out"/pmc_$name/*/*_counter_collection.csv "$out/pmc_$name.csv" > /dev/null
done
rm -rf "$out"/trace "$out"/pmc_*/
echo done
