#!/bin/bash
# Bench lines of every workload, then the cost and the effect of the margins for rays along triangle planes.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r03
mkdir -p $OUT
for wl in c2_spheres c3_mesh c4_mesh_4k c1_plumbing c5_stress; do
  timeout -k 10 280 python3 bench.py --workload $wl $( [ $wl = c2_spheres ] || echo --no-cpu-baseline ) > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || exit 1
  echo bench $wl done
done
timeout -k 10 300 python3 -u tools/grazing_cap_sweep.py c5_stress 16 > $OUT/cap_sweep_c5.txt 2>&1 || exit 1
echo sweep done
timeout -k 10 300 python3 -u tools/grazing_rate.py --rays 300000 --out $OUT/grazing_rate_default.json > $OUT/grazing_rate_default.txt 2>&1 || exit 1
timeout -k 10 300 python3 -u tools/grazing_rate.py --rays 300000 --cap 1.0 --out $OUT/grazing_rate_cap1.json > $OUT/grazing_rate_cap1.txt 2>&1 || exit 1
echo done
