set -u
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dxy or fused or config3 or smoke" > $O/pytest_dxy_v1.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dxy_v1.log
python tools/lib_ab.py tools/_ab/libpgtwin_r04.so 1e8 10 > $O/lib_ab_v1_1e8.md 2>&1; echo "ab v1 1e8 rc=$?"
for v in v2 v3 v4; do
  AB_ONLY=dxyWindow AB_B_LIB=tools/_ab/libpgtwin_$v.so python tools/lib_ab.py tools/_ab/libpgtwin_r04.so 1e8 10 > $O/lib_ab_${v}_1e8.md 2>&1; echo "ab $v 1e8 rc=$?"
done
for v in v1_pairs v2 v3 v4; do
  AB_ONLY=dxyWindow AB_B_LIB=tools/_ab/libpgtwin_$v.so python tools/lib_ab.py tools/_ab/libpgtwin_r04.so 1e9 6 > $O/lib_ab_${v}_1e9.md 2>&1; echo "ab $v 1e9 rc=$?"
done
tail -n 4 $O/lib_ab_*.md
