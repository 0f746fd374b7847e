#!/bin/bash
# Same-box A/B of several builds of the library: tools/ab_libs.sh "<lib1.so> <lib2.so> ..." "<extra bench flags>" [cfg ...]
# (files under dqo-map_amd/lib/); three alternations per config; prints ms per iteration and the per-kernel microseconds.
L=dqo-map_amd/lib
libs="$1"; flags="$2"; shift; shift
cp $L/libdqoraster.so $L/ab_keep.so
for c in ${@:-3}; do
  for i in $(seq 1 ${AB_REPS:-3}); do
    for v in $libs; do
      cp $L/$v $L/libdqoraster.so
      timeout -k 10 300 python bench.py --cfg $c --growth-every 0 --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-aux --no-selfcheck $flags 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config'].get('kernel_us',{}); print('cfg$c $v', d['ms_per_step'], {n:k[n] for n in k if n in ('gaussian_tail_kernel','adam_kernel','record_sum_kernel','gaussian_backward_kernel','blend_backward_kernel','blend_forward_kernel','bin_count_kernel','tile_sort_kernel','tile_sort_wave_kernel')})" || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
    done
  done
done
cp $L/ab_keep.so $L/libdqoraster.so
