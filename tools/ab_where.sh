#!/bin/bash
# Same-box comparison of the two homes of the late per-Gaussian forward part (DQO_K1_WHERE = 0 preprocess, 1 sort launch):
#   tools/ab_where.sh "<extra bench flags>" [cfg ...]
flags="$1"; shift
for c in ${@:-3}; do
  for i in 1 2 3; do
    for v in 0 1 2; do
      DQO_K1_WHERE=$v timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-selfcheck $flags 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_us',{}); print('cfg$c where=$v', d['ms_per_step'], {n:round(k[n],1) for n in k if n in ('preprocess_kernel','tile_sort_wave_kernel','tile_sort_kernel')})" || exit 1
    done
  done
done
