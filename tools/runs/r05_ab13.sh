set -u
O=gpurun_out/r05; mkdir -p $O
for n in 1e8 1e9; do
  r=10; [ $n = 1e9 ] && r=6
  AB_ONLY=fstWindow AB_B_LIB=tools/_ab/libpgtwin_abswap.so python tools/lib_ab.py tools/_ab/libpgtwin_r05q.so $n $r > $O/lib_ab_abswap_$n.md 2>&1; echo "ab abswap $n rc=$?"; tail -n 1 $O/lib_ab_abswap_$n.md
  AB_ONLY=fused AB_B_LIB=tools/_ab/libpgtwin_gbstag.so python tools/lib_ab.py tools/_ab/libpgtwin_r05q.so $n $r > $O/lib_ab_gbstag_$n.md 2>&1; echo "ab gbstag $n rc=$?"; tail -n 1 $O/lib_ab_gbstag_$n.md
done
