#!/bin/bash
# Wave-state counter passes over the bf16 inference workload: bash tools/pmc_infer.sh <out dir>
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/${1:-pmc_infer}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    echo "pass $i: $pass"
    timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$out/p$i" -- python3 "$root/bench.py" --infer --precision bfloat16 --clips 16 > /dev/null 2> "$out/p$i.log"
    python3 "$root/tools/pmc_by_kernel.py" "$out"/p$i/*/*_counter_collection.csv "$out/p$i.csv" | grep gblock_fwd || true
    rm -rf "$out/p$i"
done
