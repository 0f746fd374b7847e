#!/bin/bash
# Phase profile of gaussian_tail_kernel: builds with -DTAIL_STOP=0..3 (temporary patches, results invalid) under .ab_old/t*.so
L=dqo-map_amd/lib
cp $L/libdqoraster.so $L/ab_keep.so
for v in t0 t1 t2 t3 t4; do
  cp .ab_old/$v.so $L/libdqoraster.so
  echo "== $v"
  timeout -k 10 300 python tools/window_profile.py 3 1.0 0.1 2>&1 | grep "trained fraction" | sed -e 's/header.*//' || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
done
cp $L/ab_keep.so $L/libdqoraster.so
