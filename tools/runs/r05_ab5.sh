set -u
O=gpurun_out/r05; mkdir -p $O
python tools/lib_ab.py tools/_ab/libpgtwin_r05base.so 1e8 10 > $O/lib_ab_rotation_all_1e8.md 2>&1; echo "rc=$?"; tail -n 8 $O/lib_ab_rotation_all_1e8.md
python tools/lib_ab.py tools/_ab/libpgtwin_r05base.so 1e9 6 > $O/lib_ab_rotation_all_1e9.md 2>&1; echo "rc=$?"; tail -n 6 $O/lib_ab_rotation_all_1e9.md
python tools/lib_ab.py tools/_ab/libpgtwin_r05base.so 1.25e8 10 > $O/lib_ab_rotation_all_1.25e8.md 2>&1; echo "rc=$?"; tail -n 8 $O/lib_ab_rotation_all_1.25e8.md
