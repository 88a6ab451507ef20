set -u
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L > $O/rocprofv3_counters_list.txt 2>&1; echo "list rc=$?"
grep -c . $O/rocprofv3_counters_list.txt
N=${1:-1e8}
run() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/prof_pmc_$N -o $name -- python3 tools/pmc_kernels.py $N > $O/pmc_${name}_$N.log 2>&1; echo "pmc $name rc=$?"
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run TCC TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_COUNT
python tools/trim_rocprof.py pmc $O/prof_pmc_$N > $O/pmc_counters_all_$N.csv
tail -n 80 $O/pmc_counters_all_$N.csv
rm -rf $O/prof_pmc_$N
