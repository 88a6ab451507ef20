set -u
O=gpurun_out/r05; mkdir -p $O
PGT_TEST_PRETEND_TWO_GPUS=1 timeout -k 10 1000 python -m pytest tests/test_multi_gpu.py tests/test_bench_script.py tests/test_abi.py -q -m gpu -x --durations=10 > $O/pytest_bench_multi.log 2>&1; echo "pytest rc=$?"; tail -n 25 $O/pytest_bench_multi.log
