set -u
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
python tools/lib_ab.py tools/_ab/libpgtwin_r04.so 1e8 10 4 > $O/lib_ab_fold_1e8_burst4.md 2>&1; echo "ab rc=$?"; tail -n 9 $O/lib_ab_fold_1e8_burst4.md
python tools/lib_ab.py tools/_ab/libpgtwin_r04.so 1e8 10 > $O/lib_ab_fold_1e8.md 2>&1; echo "ab rc=$?"; tail -n 9 $O/lib_ab_fold_1e8.md
timeout -k 10 900 python -m pytest tests -q -m gpu -x --durations=25 > $O/pytest_gpu_first.log 2>&1; echo "pytest rc=$?"; tail -n 40 $O/pytest_gpu_first.log
