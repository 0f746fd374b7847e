#!/bin/bash
# Same-box A/B of dropping the long-list sort launch: tools/ab_skip_long.sh [cfg ...]  (DQO_SKIP_LONG_SORT=0 / default)
for c in ${@:-3}; do
  for i in 1 2 3 4 5 6; do
    for v in 0 1; do
      DQO_SKIP_LONG_SORT=$v timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 600 --warmup 40 --no-cpu-baseline --no-pmc --no-aux 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_us',{}); print('cfg$c DQO_SKIP_LONG_SORT=$v', d['ms_per_step'], d['config'].get('selfcheck'), {n:k[n] for n in k if n in ('tile_sort_kernel','tile_sort_wave_kernel','blend_forward_kernel')})" || exit 1
    done
  done
done
