#!/bin/bash
# Re-measure one bench workload on the GPU box and leave everything under gpurun_out/<tag>/:
#   bash tools/refresh_profiles.sh <tag> [c2|c2bf16|c3|c5]
# The bench line, rocprofv3 kernel-trace statistics of the same command, and the FETCH_SIZE / WRITE_SIZE / SQ
# counter passes (one rocprofv3 run each, counters only) aggregated per kernel.
#   c2      stage 3 fade-in, batch 64, fp32 (the headline workload)
#   c2bf16  the same step with bf16 storage / bf16 MFMA
#   c3      the full-resolution step (stage 6, bf16, micro-batches of 4, lean generator tape kept) at batch 16: the
#           launches are those of the batch-256 step, 16 times fewer of them
#   c5      generator-only inference at stage 6, bf16, chunks of 8
set -e -o pipefail
tag=$1
mode=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
case $mode in
  c2)     args="--no-extras --no-cpu-baseline"; pmc_args="--steps 1 --warmup 0 $args" ;;
  c2bf16) args="--precision bfloat16 --no-extras --no-cpu-baseline"; pmc_args="--steps 1 --warmup 0 $args" ;;
  c3)     args="--stage 6 --batch 16 --micro-batch 4 --precision bfloat16 --steps 1 --warmup 1 --no-extras --no-cpu-baseline"; pmc_args="$args --warmup 0" ;;
  c5)     args="--infer --precision bfloat16 --clips 64"; pmc_args="--infer --precision bfloat16 --clips 16" ;;
  *) echo "unknown mode $mode"; exit 2 ;;
esac
cd /tmp && export TMPDIR=/tmp
echo "[$tag] bench $args"
timeout -k 10 400 python3 "$root/bench.py" $args > "$out/bench.json"
echo "[$tag] kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/bench_under_rocprof.json" 2> "$out/trace.log"
cp "$out"/trace/*/*_kernel_stats.csv "$out/kernel_stats.csv"
for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    name=$(echo "$pass" | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    echo "[$tag] counters: $pass"
    timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python3 "$root/bench.py" $pmc_args > /dev/null 2> "$out/pmc_$name.log"
    python3 "$root/tools/pmc_by_kernel.py" "$out"/pmc_$name/*/*_counter_collection.csv "$out/pmc_$name.csv" > /dev/null
done
rm -rf "$out"/trace "$out"/pmc_*/
echo "[$tag] done"
