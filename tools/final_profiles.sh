#!/bin/bash
# End-of-round measurements on one MI355X (run through gpurun): default bench, rocprofv3 kernel stats of the same command, the other
# BASELINE maps, the SQ counters of the blend kernels.  Results under gpurun_out/final/ (copied to profiles/r<round>_* afterwards).
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
out=gpurun_out/final; mkdir -p $out
DQO_BENCH_KEEP_KERNEL_STATS=$out/kernel_stats_child.csv timeout -k 10 700 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || exit 1
echo "cfg3 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-aux > $out/ks.log 2>&1 || exit 1
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
echo "kernel stats done"
for c in 2 4 5; do
  steps=50; [ $c = 5 ] && steps=1000   # cfg 5: ten growth steps (every 100 iterations) inside the timed region
  timeout -k 10 500 python bench.py --cfg $c --steps $steps --no-cpu-baseline > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err || exit 1
  echo "cfg$c done"
done
timeout -k 10 300 python bench.py --cfg 5 --growth-every 0 --no-cpu-baseline --no-pmc --no-aux > $out/bench_cfg5_nogrowth.json 2> $out/bench_cfg5_nogrowth.err || exit 1
# two ranks on the one GPU of the box (gloo: RCCL wants one device per rank): a rehearsal of the launcher, the sharded job and its
# self-checks, not a scaling number
DQO_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 50 --warmup 10 > $out/bench_cfg3_2ranks_one_gpu.json 2> $out/bench_cfg3_2ranks_one_gpu.err || echo "2-rank rehearsal failed"
timeout -k 10 300 python tools/window_profile.py 3 1.0 0.1 > $out/window_profile.txt 2>&1 || echo "window profile failed"
echo "all done"
