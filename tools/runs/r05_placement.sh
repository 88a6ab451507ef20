set -u
O=gpurun_out/r05; mkdir -p $O
: > $O/placement_skip_probe.txt
for k in 0 40 0 40 100 0 40 100 0 160; do python tools/attic/placement_skip_probe.py $k >> $O/placement_skip_probe.txt 2>&1; done
grep skip $O/placement_skip_probe.txt
