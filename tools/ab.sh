#!/bin/bash
# Same-box A/B of two builds of the library: dqo-map_amd/lib/ab_base.so against ab_new.so (build each variant, copy it there), three
# alternations per config; prints ms per iteration.      tools/ab.sh [cfg ...]
L=dqo-map_amd/lib
for c in ${@:-3}; do
  for i in 1 2 3; do
    for v in base new; do
      cp $L/ab_$v.so $L/libdqoraster.so
      timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-roofline --no-selfcheck 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c $v', d['ms_per_step'])" || exit 1
    done
  done
done
cp $L/ab_new.so $L/libdqoraster.so
