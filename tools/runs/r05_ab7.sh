set -u
O=gpurun_out/r05; mkdir -p $O
for n in 1e8 1.25e8 2.5e8; do
AB_ONLY=AF python tools/lib_ab.py tools/_ab/libpgtwin_r05rot.so $n 10 > $O/lib_ab_af_rotation_$n.md 2>&1; echo "rc=$?"; tail -n 2 $O/lib_ab_af_rotation_$n.md
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_parity_rot.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_parity_rot.log
