#!/bin/bash
# bench.py N > 1 on a one-GPU box: every rank on GPU 0, collectives over gloo (logic only).  Prints one summary line per variant.
export PGT_BENCH_BACKEND=gloo PGT_BENCH_SHARE_GPU=1
S="--sites 3e6 --chroms 5 --steps 3 --warmup 1 --headline-only --no-cpu"
for args in "--gpus 3" "--gpus 3 --workload pairs --pairs 3" "--gpus 2 --scaling weak" "--gpus 2 --exchange gather" "--gpus 2 --exchange peer" "--gpus 3 --exchange both" "--gpus 2 --no-verify" "--gpus 5 --stepsize 100"; do
  python bench.py $args $S > /tmp/out.txt 2>/tmp/err.txt; rc=$?
  python - "$args" $rc <<'PY'
import json,sys
args,rc=sys.argv[1],sys.argv[2]
try:
    line=[l for l in open('/tmp/out.txt') if l.startswith('{')][-1]; j=json.loads(line)
    print(args, 'rc',rc, 'ok',j.get('ok'), 'n_gpus',j['n_gpus'],'exchange',j['config']['row_exchange'],'backend',j['config'].get('collective_backend'),'check',j['rows_check'], {k:(v.get('rows_check'),v.get('headline')) for k,v in j['extra'].items() if k.startswith('exchange')})
except Exception as e:
    print(args,'rc',rc,'NO LINE',e, open('/tmp/err.txt').read()[-600:])
PY
done
