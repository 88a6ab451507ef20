#!/bin/bash
# steady-state rate of the staging ring against its geometry (calls 1 and 2 of the probe), fst 1e8 = 2 GB
for w in 2 4 6 8 12; do for c in 2 4 8 16; do
  echo -n "workers $w chunk ${c}MiB: "
  PGT_UPLOAD_WORKERS=$w PGT_UPLOAD_CHUNK_MIB=$c python3 tools/probes/host_api_probe.py fst 1e8 prepare 2>/dev/null | grep -o "calls_ms=.*\]"
done; done
echo -n "plain: "; PGT_UPLOAD=plain python3 tools/probes/host_api_probe.py fst 1e8 prepare 2>/dev/null | grep -o "calls_ms=.*\]"
