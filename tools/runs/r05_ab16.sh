set -u
O=gpurun_out/r05; mkdir -p $O
for n in 1e8 1e9; do
for v in fqseq fq2; do
  r=10; [ $n = 1e9 ] && r=5
  AB_ONLY=fused AB_B_LIB=tools/_ab/libpgtwin_$v.so python tools/lib_ab.py tools/_ab/libpgtwin_r05final.so $n $r 4 > $O/lib_ab_${v}_$n.md 2>&1; echo "ab $v $n rc=$?"; tail -n 1 $O/lib_ab_${v}_$n.md | cut -c1-200
done; done
