set -u
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 500 python tests/ingest_fuzz.py 420 505 > $O/ingest_fuzz.txt 2>&1; echo "ingest fuzz rc=$?"; tail -n 4 $O/ingest_fuzz.txt
timeout -k 10 400 python tests/cli_end_to_end.py 1e8 > $O/cli_end_to_end_1e8.md 2>&1; echo "e2e rc=$?"; tail -n 12 $O/cli_end_to_end_1e8.md
