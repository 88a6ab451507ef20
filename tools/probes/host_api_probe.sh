#!/bin/bash
# every (statistic, upload path) in a process of its own: first calls are what a command-line tool pays
for stat in het dxy fst; do
  n=1e8; [ $stat = dxy ] && n=2e7
  for mode in plain ring; do
    echo "== $stat $n PGT_UPLOAD=$mode"
    PGT_UPLOAD=$mode PGT_TRACE_API=1 python3 tools/probes/host_api_probe.py $stat $n 2>&1 | grep -v "^\[pgt-api\].*workspace\|amdgpu.ids"
  done
  echo "== $stat $n ring, prepared at open"
  PGT_TRACE_API=1 python3 tools/probes/host_api_probe.py $stat $n prepare 2>&1 | grep "probe\|upload=\|staging"
done
