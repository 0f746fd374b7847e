#!/bin/bash
# Same inputs, several builds of the library (.ab_old/*.so): which part of the per-Gaussian chain moves the ill-conditioned rows?
L=dqo-map_amd/lib
cp $L/libdqoraster.so $L/ab_keep.so
for v in ${LIBS:-m0 m4 m6 m7 m20 m22 m23 m31 m14 m3}; do
  cp .ab_old/$v.so $L/libdqoraster.so
  for c in "1 2000" "3 60000" "5 300000"; do
    echo "== $v cfg/P $c"
    timeout -k 10 300 python tests/diag_row_error.py $c 2>&1 | grep -v "^   row\|amdgpu.ids" || { cp $L/ab_keep.so $L/libdqoraster.so; exit 1; }
  done
done
cp $L/ab_keep.so $L/libdqoraster.so
