#!/bin/bash
# SQ counter pass over tools/bench_bchain.py (one rocprofv3 run): bash tools/pmc_bchain.sh <out dir>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/${1:-pmc_bchain}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc ${MGX_PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY} --output-format csv -d "$out/p1" -- python3 "$root/tools/bench_bchain.py" --reps 1 > /dev/null 2> "$out/p1.log"
python3 "$root/tools/pmc_by_kernel.py" "$out"/p1/*/*_counter_collection.csv "$out/p1.csv" | grep -E "bchain|bconv" || true
rm -rf "$out/p1"
