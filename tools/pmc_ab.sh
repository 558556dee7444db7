#!/bin/bash
# SQ counter passes of one workload for one build of the library: tools/pmc_ab.sh NAME [LIB] [workload] [variant]
# (each --pmc pass in a run of its own; prints the per-kernel means)
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NAME=$1; LIB=${2:-raytracer_amd/librtmi.so}; WL=${3:-c2_spheres}; VAR=${4:-2}
OUT=gpurun_out/pmc_$NAME
mkdir -p $OUT
export RTMI_LIBRARY=$PWD/$LIB
pass() { n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$n -o $n --output-format csv -- python3 tools/run_workload.py $WL $VAR 2 > $OUT/$n.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass sq3 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH
python3 tools/pmc_summary.py $OUT > $OUT.txt
grep -A40 "k_paths\|k_stream" $OUT.txt | head -45
