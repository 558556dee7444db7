#!/bin/bash
# One profiling pass for the committed evidence: kernel-trace statistics of the bench command, PMC passes
# (each in a run of its own) for the default workload and for the stress workload.
# usage (on the GPU box): bash tools/profile_all.sh
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/profile
mkdir -p $OUT
# 1. the bench command itself under the kernel trace
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_bench -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
cat $(find $OUT/ks_bench -name "*kernel_stats.csv") > $OUT/kernel_stats_c2.csv
# 2. other workloads
variant_of() { case $1 in c5_*) echo 4;; *) echo 2;; esac; }  # hierarchy scenes: the pixel-streaming kernel
for wl in c3_mesh c4_mesh_4k c5_stress; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_$wl -o $wl --output-format csv -- python3 tools/run_workload.py $wl $(variant_of $wl) 4 > $OUT/run_$wl.log 2>&1
  cat $(find $OUT/ks_$wl -name "*kernel_stats.csv") > $OUT/kernel_stats_$wl.csv
done
# 3. PMC passes (c2 default workload: 2 launches of 256 frames)
pmc() { # name workload counters...
  name=$1; wl=$2; shift; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d $OUT/pmc_$wl/$name -o $name --output-format csv -- python3 tools/run_workload.py $wl $(variant_of $wl) 2 > $OUT/pmc_${wl}_$name.log 2>&1
}
for wl in c2_spheres c5_stress; do
  pmc sq1 $wl SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
  pmc sq2 $wl SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
  pmc fetch $wl FETCH_SIZE
  pmc write $wl WRITE_SIZE
  pmc tcc $wl TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  python3 tools/pmc_summary.py $OUT/pmc_$wl > $OUT/pmc_$wl.txt
done
echo done
