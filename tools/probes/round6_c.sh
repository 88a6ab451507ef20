#!/bin/bash
for v in af_leaf512 af_leaf1024; do
  echo "### A = $v, B = af_cur"
  AB_ONLY="AF front end, 8" AB_B_LIB=tools/_ab/libpgtwin_af_cur.so python3 tools/lib_ab.py tools/_ab/libpgtwin_$v.so 1e8 6 4 2>&1 | grep -v "amdgpu.ids" | tail -3
done
for v in fst_no_l1_stores fst_half_l1_stores; do
  for n in 1e8 1e9; do
    echo "### A = $v, B = the tree, $n sites"
    AB_ONLY="fstWindow" python3 tools/lib_ab.py tools/_ab/libpgtwin_$v.so $n 6 4 2>&1 | grep -v "amdgpu.ids" | tail -3
  done
done
