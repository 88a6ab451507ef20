set -u
O=gpurun_out/r05; mkdir -p $O
for n in 1e8 1.25e8 1e9; do
  r=10; [ $n = 1e9 ] && r=6
  python tools/lib_ab.py tools/_ab/libpgtwin_r04.so $n $r 4 > $O/lib_ab_round4_vs_round5_$n.md 2>&1; echo "rc=$?"; tail -n 8 $O/lib_ab_round4_vs_round5_$n.md | cut -c1-170
done
