#!/bin/bash
# rocprofv3 kernel-trace statistics of one workload: tools/kernel_stats.sh OUTNAME workload variant launches frames [accel]
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NAME=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$NAME -o $NAME --output-format csv -- python3 tools/run_workload.py "$@" > gpurun_out/ks_$NAME.log 2>&1
find gpurun_out/ks_$NAME -name "*kernel_stats.csv" -exec cat {} \;
