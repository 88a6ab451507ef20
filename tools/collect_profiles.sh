#!/bin/bash
# Collects the evidence of one round on a gpurun box into gpurun_out/<round>/ (copied to profiles/<round>/ afterwards):
#   smoke.log                    python -c "import __graft_entry__ as g; g.smoke()"  (the driver's gate; runs FIRST)
#   bench_default.log            python bench.py (the driver's default line)
#   pmc_headline.json            the build kernel's HBM bytes per launch + the hash of the kernel sources (-> profiles/pmc_headline.json)
#   bench_under_rocprofv3.log    the headline-only run under rocprofv3 --kernel-trace --stats; *_kernel_stats.csv trimmed
#   pmc_counters.csv             separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 3-step headline run, trimmed
#   kernels_1e8/1e9_kernel_stats.csv   rocprofv3's own average duration of every build kernel (tools/pmc_kernels.py)
#   bench_2rank_rehearsal.log    plain `python bench.py --gpus 2` (self-launching) with two ranks sharing the one GPU
# usage: bash tools/collect_profiles.sh r04
set -u
R=${1:-r04}
OUT=gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
python bench.py > "$OUT/bench_default.log" 2>&1; echo "bench default rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o bench -- python3 bench.py --steps 20 --warmup 3 --headline-only > "$OUT/bench_under_rocprofv3.log" 2>&1; echo "rocprof stats rc=$?"
python tools/trim_rocprof.py stats "$(find "$OUT/prof_stats" -name '*kernel_stats.csv' | head -1)" > "$OUT/bench_1e9_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/prof_pmc" -o $c -- python3 bench.py --steps 3 --warmup 1 --no-cpu --headline-only --prewarm-seconds 0 > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
done
python tools/trim_rocprof.py pmc "$(dirname "$(find "$OUT/prof_pmc" -name '*counter_collection.csv' | head -1)")" > "$OUT/pmc_counters.csv" 2>/dev/null || python tools/trim_rocprof.py pmc "$OUT/prof_pmc" > "$OUT/pmc_counters.csv"
python tools/trim_rocprof.py headline "$OUT/pmc_counters.csv" "$R" "bench.py --steps 3 --warmup 1 --no-cpu --headline-only --prewarm-seconds 0" > "$OUT/pmc_headline.json"; echo "pmc headline rc=$?"
for n in 1e8 1e9; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k$n" -o k -- python3 tools/pmc_kernels.py $n > "$OUT/kernels_$n.log" 2>&1; echo "kernels $n rc=$?"
  python tools/trim_rocprof.py stats "$(find "$OUT/prof_k$n" -name '*kernel_stats.csv' | head -1)" > "$OUT/kernels_${n}_kernel_stats.csv"
done
PGT_BENCH_BACKEND=gloo PGT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --sites 2e8 --chroms 8 --steps 5 --warmup 2 > "$OUT/bench_2rank_rehearsal.log" 2>&1; echo "2-rank rehearsal rc=$?"
rm -rf "$OUT"/prof_stats "$OUT"/prof_pmc "$OUT"/prof_k1e8 "$OUT"/prof_k1e9
ls -la "$OUT"
