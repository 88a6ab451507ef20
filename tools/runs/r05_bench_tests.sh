set -u
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default_first.log 2> $O/bench_default_first.err; echo "bench rc=$?"; tail -c 1500 $O/bench_default_first.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05/bench_default_first.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','ok','degraded','rows_check')})
print(d['roofline'])
print({k:(v.get('ms_per_step'),v.get('roofline_frac')) for k,v in d['extra'].items() if isinstance(v,dict) and 'ms_per_step' in v})
print(d['cpu_baseline'])
PY
PGT_TEST_PRETEND_TWO_GPUS=1 timeout -k 10 900 python -m pytest tests/test_multi_gpu.py tests/test_bench_script.py tests/test_abi.py -q -m gpu -x --durations=10 > $O/pytest_bench_multi.log 2>&1; echo "pytest rc=$?"; tail -n 25 $O/pytest_bench_multi.log
timeout -k 10 600 python -m pytest tests/test_cli.py -q -m gpu -x -k "extreme" > $O/pytest_extreme.log 2>&1; echo "pytest extreme rc=$?"; tail -n 5 $O/pytest_extreme.log
