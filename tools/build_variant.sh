#!/bin/bash
# Build a tuning variant of librtmi.so: tools/build_variant.sh NAME "<extra hipcc flags>"
# Output: scratch/librtmi_NAME.so (select it with RTMI_LIBRARY=...).  Needs a prior normal build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1
EXTRA=$2
mkdir -p "$ROOT/scratch"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden \
    -I"$ROOT/include" -I"$ROOT/raytracer_amd/csrc" $EXTRA \
    -c "$ROOT/raytracer_amd/csrc/rt_kernels.hip" -o "$ROOT/scratch/rt_kernels_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$ROOT/scratch/librtmi_$NAME.so" \
    "$ROOT/scratch/rt_kernels_$NAME.o" $(ls "$ROOT"/build/rtmi/*.o | grep -v rt_kernels.o)
echo "built scratch/librtmi_$NAME.so"
