#!/bin/bash
# Same-box A/B of two library builds (lib/ab_base.so, lib/ab_new.so) on one shard of an N-rank job: tools/ab_shard.sh R/N [cfg]
L=dqo-map_amd/lib
for i in 1 2 3; do
  for v in base new; do
    cp $L/ab_$v.so $L/libdqoraster.so
    timeout -k 10 300 python bench.py --cfg ${2:-3} --growth-every 0 --as-shard $1 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-roofline --no-selfcheck 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard $1 $v', d['ms_per_step'])" || exit 1
  done
done
cp $L/ab_new.so $L/libdqoraster.so
