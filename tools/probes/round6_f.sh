#!/bin/bash
for v in af_q8byte af_q2; do
  echo "### A = $v, B = af_q4"
  AB_ONLY="AF front end" AB_B_LIB=tools/_ab/libpgtwin_af_q4.so python3 tools/lib_ab.py tools/_ab/libpgtwin_$v.so 1e8 6 4 2>&1 | grep -v "amdgpu.ids" | tail -4
done
