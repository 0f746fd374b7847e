#!/bin/bash
# Same-box A/B of an environment switch: tools/ab_env.sh VAR "<extra bench flags>" [cfg ...]  (three alternations per config: VAR unset / VAR=1)
var="$1"; flags="$2"; shift; shift
for c in ${@:-3}; do
  for i in 1 2 3; do
    for v in "" 1; do
      env $var=$v timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-selfcheck $flags 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_us',{}); print('cfg$c $var=$v', d['ms_per_step'], {n:k[n] for n in k if n in ('gaussian_tail_kernel','blend_backward_kernel','preprocess_kernel')})" || exit 1
    done
  done
done
