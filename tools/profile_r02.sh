#!/bin/bash
# The profile set in one go (on the GPU box): bench lines of every workload, the default bench command under
# rocprofv3 --kernel-trace --stats, SQ counter passes for C2 / C3 / C5, the lane tallies and cycle stamps of
# k_paths (needs scratch/librtmi_prof.so and scratch/librtmi_stamps.so from tools/build_variant.sh).
# Results land in gpurun_out/r02/; copy what is to be kept into profiles/.
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r02
mkdir -p $OUT
for wl in c2_spheres c3_mesh c4_mesh_4k c1_plumbing c5_stress; do
  timeout -k 10 280 python3 bench.py --workload $wl $( [ $wl = c2_spheres ] || echo --no-cpu-baseline ) > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_bench -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
cat $(find $OUT/ks_bench -name "*kernel_stats.csv") > $OUT/kernel_stats_c2_spheres.csv
for wl in c3_mesh c4_mesh_4k c5_stress; do
  v=2; [ $wl = c5_stress ] && v=4
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_$wl -o $wl --output-format csv -- python3 tools/run_workload.py $wl $v 4 > $OUT/run_$wl.log 2>&1
  cat $(find $OUT/ks_$wl -name "*kernel_stats.csv") > $OUT/kernel_stats_$wl.csv
done
for wl in c2_spheres c3_mesh c5_stress; do
  v=2; [ $wl = c5_stress ] && v=4
  bash tools/pmc_ab.sh r02_$wl raytracer_amd/librtmi.so $wl $v > /dev/null
  cp gpurun_out/pmc_r02_$wl.txt $OUT/pmc_$wl.txt
done
for wl in c2_spheres c3_mesh; do
  RTMI_LIBRARY=scratch/librtmi_prof.so timeout -k 10 100 python3 tools/paths_profile.py $wl > $OUT/paths_lanes_$wl.txt
  RTMI_LIBRARY=scratch/librtmi_stamps.so timeout -k 10 100 python3 tools/paths_stamps.py $wl > $OUT/paths_stamps_$wl.txt
done
echo done
