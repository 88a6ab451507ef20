set -u
O=gpurun_out/r05; mkdir -p $O
python tools/lib_ab.py tools/_ab/libpgtwin_r05rot2.so 1e8 10 4 > $O/lib_ab_query_frontloaded_1e8_burst4.md 2>&1; echo "rc=$?"; tail -n 8 $O/lib_ab_query_frontloaded_1e8_burst4.md
python tools/lib_ab.py tools/_ab/libpgtwin_r05rot2.so 1e9 6 4 > $O/lib_ab_query_frontloaded_1e9_burst4.md 2>&1; echo "rc=$?"; tail -n 6 $O/lib_ab_query_frontloaded_1e9_burst4.md
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_parity_query.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_parity_query.log
timeout -k 10 600 python tests/gpu_fuzz.py 3000 > $O/gpu_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -n 3 $O/gpu_fuzz.txt
