#!/bin/bash
# Bit-for-bit A/B of library builds: tools/ab_bits.sh "<lib1.so> <lib2.so> ..." [cfg ...]  (files under dqo-map_amd/lib/); prints
# tools/state_hash.py's fingerprints per build — equal lines = equal bits.
L=dqo-map_amd/lib
libs="$1"; shift
cp $L/libdqoraster.so $L/ab_keep.so
for c in ${@:-3}; do
  for v in $libs; do
    cp $L/$v $L/libdqoraster.so
    echo "== $v"
    timeout -k 10 300 python tools/state_hash.py $c 8 2>&1 | grep "^cfg" || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
  done
done
cp $L/ab_keep.so $L/libdqoraster.so
