#!/bin/bash
# Collects the evidence of one round on a gpurun box into gpurun_out/<round>/ (copied to profiles/<round>/ afterwards):
#   smoke.log                    python -c "import __graft_entry__ as g; g.smoke()"  (the driver's gate; runs FIRST)
#   bench_default.log            python bench.py (the driver's default line)
#   pmc_headline.json            the build kernel's HBM bytes per launch + the hash of the kernel sources (-> profiles/pmc_headline.json)
#   bench_under_rocprofv3.log    the headline-only run under rocprofv3 --kernel-trace --stats; *_kernel_stats.csv trimmed
#   pmc_counters.csv             separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 3-step headline run, trimmed
#   kernels_1e8/1e9_kernel_stats.csv   rocprofv3's own average duration of every build kernel (tools/pmc_kernels.py)
#   kernels_1e8/1e9_timeline.md        the same launches in order: duration of every kernel and the idle gap in front of it
#   pmc_counters_all_1e8/1e9.csv, pmc_traffic_all_kernels_1e8/1e9.md   per build kernel: FETCH_SIZE / WRITE_SIZE / SQ / TCC counters;
#                                HBM traffic against the algorithmic bytes and the wave-cycle split (tools/trim_rocprof.py traffic)
#   bench_2rank_rehearsal.log    plain `python bench.py --gpus 2` (self-launching) with two ranks sharing the one GPU
# usage: bash tools/collect_profiles.sh r06
set -u
R=${1:-r06}
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
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k$n" -o k -- python3 tools/pmc_kernels.py $n 10 > "$OUT/kernels_$n.log" 2>&1; echo "kernels $n rc=$?"
  python tools/trim_rocprof.py stats "$(find "$OUT/prof_k$n" -name '*kernel_stats.csv' | head -1)" > "$OUT/kernels_${n}_kernel_stats.csv"
  python tools/trim_rocprof.py timeline "$(find "$OUT/prof_k$n" -name '*kernel_trace.csv' | head -1)" > "$OUT/kernels_${n}_timeline.md"
  # EVERY build kernel's counters (round 5): HBM traffic against the algorithmic bytes, and where the wave cycles go.  Separate
  # passes (FETCH_SIZE and WRITE_SIZE do not fit one; the SQ block has 8 slots), the program directly behind `--`.
  pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/prof_pmc_all_$n" -o $name -- python3 tools/pmc_kernels.py $n > "$OUT/pmc_all_${name}_$n.log" 2>&1; echo "pmc all $name $n rc=$?"; }
  pass FETCH_SIZE FETCH_SIZE
  pass WRITE_SIZE WRITE_SIZE
  pass SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM
  pass TCC TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_COUNT
  python tools/trim_rocprof.py pmc "$OUT/prof_pmc_all_$n" | grep -v "^[A-Za-z0-9_]*_counter_collection.csv" > "$OUT/pmc_counters_all_$n.csv"   # the per-kernel summary rows only
  python tools/trim_rocprof.py traffic "$OUT/pmc_counters_all_$n.csv" $n > "$OUT/pmc_traffic_all_kernels_$n.md"
  rm -f "$OUT"/pmc_all_*_$n.log
done
PGT_BENCH_BACKEND=gloo PGT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --sites 2e8 --chroms 8 --steps 5 --warmup 2 > "$OUT/bench_2rank_rehearsal.log" 2>&1; echo "2-rank rehearsal rc=$?"
rm -rf "$OUT"/prof_stats "$OUT"/prof_pmc "$OUT"/prof_k1e8 "$OUT"/prof_k1e9 "$OUT"/prof_pmc_all_1e8 "$OUT"/prof_pmc_all_1e9
ls -la "$OUT"
