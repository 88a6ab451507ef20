set -u
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=30 > $O/pytest_gpu_second.log 2>&1; echo "pytest rc=$?"; tail -n 45 $O/pytest_gpu_second.log
