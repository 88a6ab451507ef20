#!/bin/bash
# AF front end, 8 populations: every variant library (A) against the current build (B = tools/_ab/libpgtwin_af_cur.so), interleaved
for v in r05final af_leaf512 "$@"; do
  [ -f tools/_ab/libpgtwin_$v.so ] || continue
  echo "### A = $v"
  AB_ONLY="AF front end, 8" AB_B_LIB=tools/_ab/libpgtwin_af_cur.so python3 tools/lib_ab.py tools/_ab/libpgtwin_$v.so 1e8 6 4 2>&1 | grep -v "amdgpu.ids" | tail -4
done
