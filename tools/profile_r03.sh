#!/bin/bash
# Profile set (on the GPU box): bench lines of every workload (and of C2-C4 in the fused arithmetic profile, and of the C5 variant with transmitting materials), the default bench command under
# rocprofv3 --kernel-trace --stats, kernel statistics of C3 / C4 (the bench's own launch sizes) / C5 / C1, lane
# tallies of k_paths (sphere and mesh parts) and cycle stamps, the frame-loop rate at the reference's defaults.
# Needs scratch/librtmi_prof.so, _prof2.so, _stamps.so from tools/build_variant.sh.  Results: gpurun_out/r03/.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r03
mkdir -p $OUT
for wl in c2_spheres c3_mesh c4_mesh_4k c1_plumbing c5_stress; do
  timeout -k 10 280 python3 bench.py --workload $wl $( [ $wl = c2_spheres ] || echo --no-cpu-baseline ) > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || exit 1
done
for wl in c2_spheres c3_mesh c4_mesh_4k; do
  timeout -k 10 280 python3 bench.py --workload $wl --profile fma --no-cpu-baseline > $OUT/bench_${wl}_fma.json 2> $OUT/bench_${wl}_fma.err || exit 1
done
timeout -k 10 280 python3 bench.py --workload c5_stress_glass --no-cpu-baseline > $OUT/bench_c5_stress_glass.json 2> $OUT/bench_c5_stress_glass.err || exit 1
echo benches done
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_bench -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err || exit 1
( echo "# python3 bench.py --no-cpu-baseline  (the default bench command: 5 warm-up + 100 timed launches of 256 frames, plus the two instrumented launches that count events)"; cat $(find $OUT/ks_bench -name "*kernel_stats.csv") ) > $OUT/kernel_stats_c2_spheres.csv
for wl in c3_mesh c4_mesh_4k c5_stress c1_plumbing; do
  v=2; [ $wl = c5_stress ] && v=4
  fr=256; [ $wl = c4_mesh_4k ] && fr=512; [ $wl = c5_stress ] && fr=16; [ $wl = c1_plumbing ] && fr=64
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/ks_$wl -o $wl --output-format csv -- python3 tools/run_workload.py $wl $v 6 $fr > $OUT/run_$wl.log 2>&1 || exit 1
  ( echo "# tools/run_workload.py $wl $v 6 $fr  (6 launches of $fr frames each)"; cat $(find $OUT/ks_$wl -name "*kernel_stats.csv") ) > $OUT/kernel_stats_$wl.csv
done
echo kernel stats done
for wl in c2_spheres c3_mesh; do
  RTMI_LIBRARY=scratch/librtmi_prof.so timeout -k 10 100 python3 tools/paths_profile.py $wl > $OUT/paths_lanes_$wl.txt || exit 1
  RTMI_LIBRARY=scratch/librtmi_stamps.so timeout -k 10 100 python3 tools/paths_stamps.py $wl > $OUT/paths_stamps_$wl.txt || exit 1
done
RTMI_LIBRARY=scratch/librtmi_prof2.so timeout -k 10 100 python3 tools/paths_profile.py c3_mesh 256 meshes > $OUT/paths_lanes_c3_mesh_meshpart.txt || exit 1
RTMI_LIBRARY=scratch/librtmi_prof2.so timeout -k 10 100 python3 tools/paths_profile.py c1_plumbing 64 meshes > $OUT/paths_lanes_c1_plumbing_meshpart.txt || exit 1
timeout -k 10 200 python3 tools/frame_loop_rate.py > $OUT/frame_loop_rate.txt 2>&1 || exit 1
echo done
