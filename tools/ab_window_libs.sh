#!/bin/bash
# Same-box A/B of library builds under .ab_old/ on the window loop: tools/ab_window_libs.sh "<name> <name> ..." [fractions]
L=dqo-map_amd/lib
cp $L/libdqoraster.so $L/ab_keep.so
for i in 1 2; do
  for v in $1; do
    cp .ab_old/$v.so $L/libdqoraster.so
    echo "== $v"
    timeout -k 10 300 python tools/window_profile.py 3 ${2:-1.0} 2>&1 | grep "trained fraction" | sed -e 's/header.*//' -e 's/(replays, one launch each);//' || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
  done
done
cp $L/ab_keep.so $L/libdqoraster.so
