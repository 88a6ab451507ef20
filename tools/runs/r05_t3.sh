set -u
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cli.py -q -m gpu -x -k "dxy" --durations=10 > $O/pytest_dxy_sync.log 2>&1; echo "pytest rc=$?"; tail -n 30 $O/pytest_dxy_sync.log
