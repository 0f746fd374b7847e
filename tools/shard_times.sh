#!/bin/bash
# Every shard of an N-rank strong-scaling job run ALONE on the one GPU of the box (bench.py --as-shard R/N): the slowest shard bounds
# the N-GPU iteration time (no data-path collective).  Usage: tools/shard_times.sh [cfg] [N ...]   -> gpurun_out/shards/c<cfg>_<N>_<R>.json
cfg=${1:-3}; shift
ns=${@:-2 4 8}
mkdir -p gpurun_out/shards
for n in $ns; do
  for r in $(seq 0 $((n-1))); do
    timeout -k 10 300 python bench.py --cfg $cfg --growth-every 0 --as-shard $r/$n --graph-unroll 1 --steps 40 --warmup 8 --no-cpu-baseline --no-pmc --no-aux --sustained 0 --window 0 --placement-trials 1 \
      > gpurun_out/shards/c${cfg}_${n}_${r}.json 2> gpurun_out/shards/c${cfg}_${n}_${r}.err || exit 1
  done
done
echo done
