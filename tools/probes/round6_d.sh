#!/bin/bash
for v in af_p4_w1 af_p4_w1_burst4 af_p2_w1_burst4 af_p4_w1_cap256 af_p4_w1_cap512; do
  echo "### A = $v, B = af_cur"
  AB_ONLY="AF front end, 8" AB_B_LIB=tools/_ab/libpgtwin_af_cur.so python3 tools/lib_ab.py tools/_ab/libpgtwin_$v.so 1e8 6 4 2>&1 | grep -v "amdgpu.ids" | tail -3
done
