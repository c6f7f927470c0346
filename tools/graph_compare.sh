#!/bin/bash
# Eager step against the HIP-graph replay (bench.py --graph) at the reference's batch sizes
# (trainer/task.py:186): stage and batch per line; prints the head of each bench line.
set -e
for cfg in "0 16" "1 8" "2 4" "3 2" "4 2"; do
  set -- $cfg
  extra=""; [ "$1" = "0" ] && extra="--straight"
  timeout -k 10 200 python bench.py --stage $1 --batch $2 --steps 20 --warmup 3 --no-cpu-baseline $extra | cut -c1-140
  timeout -k 10 200 python bench.py --stage $1 --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --graph $extra | cut -c1-140
done
