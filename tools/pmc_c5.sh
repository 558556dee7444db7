set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { # name counters...
  name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d gpurun_out/pmc_c5/$name -o $name --output-format csv -- python3 tools/run_workload.py c5_stress 2 1 4 bvh > gpurun_out/pmc_c5_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python3 tools/pmc_summary.py gpurun_out/pmc_c5 > gpurun_out/pmc_c5_summary.txt
