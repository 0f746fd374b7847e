#!/bin/bash
# Same-box A/B of library builds on the drop-in op's forward + backward pair (cfg 3, 'deferred' mode, pooled contexts):
# tools/ab_op.sh "<lib1.so> <lib2.so> ..."  (files under dqo-map_amd/lib/); AB_REPS alternations (default 3).
L=dqo-map_amd/lib
cp $L/libdqoraster.so $L/ab_keep.so
for i in $(seq 1 ${AB_REPS:-3}); do
  for v in $1; do
    cp $L/$v $L/libdqoraster.so
    echo -n "$v "
    timeout -k 10 300 python tools/profile_op.py deferred 2>/dev/null | grep "steady state" | cut -c1-110 || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
  done
done
cp $L/ab_keep.so $L/libdqoraster.so
