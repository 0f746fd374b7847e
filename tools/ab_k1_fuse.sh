#!/bin/bash
# Same-box A/B of the early per-Gaussian forward at the head of bin_count_kernel: tools/ab_k1_fuse.sh [cfg ...]  (DQO_K1_FUSE=0 / default)
mkdir -p gpurun_out/r5
for c in ${@:-3}; do
  for i in 1 2 3; do
    for v in 0 1; do
      DQO_K1_FUSE=$v timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-aux 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_us',{}); print('cfg$c DQO_K1_FUSE=$v', d['ms_per_step'], d['config'].get('selfcheck'), {n:k[n] for n in k if n in ('bin_count_kernel','preprocess_kernel','gaussian_tail_kernel')})" || exit 1
    done
  done
done
