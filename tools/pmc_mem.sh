#!/bin/bash
# Texture-address / L1 counter passes of one workload: tools/pmc_mem.sh NAME [LIB] [workload] [variant] [frames]
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NAME=$1; LIB=${2:-raytracer_amd/librtmi.so}; WL=${3:-c5_stress}; VAR=${4:-4}; FR=${5:-16}
OUT=gpurun_out/pmcmem_$NAME
mkdir -p $OUT
export RTMI_LIBRARY=$PWD/$LIB
pass() { n=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d $OUT/$n -o $n --output-format csv -- python3 tools/run_workload.py $WL $VAR 2 $FR > $OUT/$n.log 2>&1
}
pass m1 TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass m2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
pass m3 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
python3 tools/pmc_summary.py $OUT > $OUT.txt
grep -A24 "k_stream\|k_paths" $OUT.txt | head -60
