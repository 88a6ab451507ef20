set -u
O=gpurun_out/r05; mkdir -p $O
for n in 1e8 1e9; do
for v in stag1 stag2 stag3; do
  r=10; [ $n = 1e9 ] && r=6
  AB_ONLY=fstWindow AB_B_LIB=tools/_ab/libpgtwin_$v.so python tools/lib_ab.py tools/_ab/libpgtwin_r05rot2.so $n $r > $O/lib_ab_${v}_$n.md 2>&1; echo "ab $v $n rc=$?"; tail -n 1 $O/lib_ab_${v}_$n.md
done; done
