set -u
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cli.py -q -m gpu -x -k "live_against or several_gpus" --durations=8 > $O/pytest_cli_trim.log 2>&1; echo "pytest rc=$?"; tail -n 14 $O/pytest_cli_trim.log
