#!/bin/bash
# Same-box A/B of a bench.py flag: tools/ab_flag.sh "<flag>" [cfg ...] — three alternations of (without, with) per config.
flag="$1"; shift
for c in ${@:-3}; do
  for i in 1 2 3; do
    for v in "" "$flag"; do
      timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-roofline --no-selfcheck $v 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c [${v:-default}]', d['ms_per_step'])" || exit 1
    done
  done
done
