#!/bin/bash
# A/B builds for the image kernels: tools/build_img_variant.sh <name> [-DFLAG...]
#   -> tools/bin/<name>/libisx_hip.so + tools/bin/<name>/img_bench (linked against that library)
# ISX_DIAG_* macros change RESULTS (they remove work to see what it costs); never ship such a build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" \
   -x hip image-stitcher_amd/csrc/*.hip -o tools/bin/$name/libisx_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/img_bench.hip -Iinclude -Ltools/bin/$name -lisx_hip \
   -Wl,-rpath,'$ORIGIN' -o tools/bin/$name/img_bench
echo tools/bin/$name/img_bench
