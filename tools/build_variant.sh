#!/bin/bash
# Experimental builds of libisx_hip.so: tools/build_variant.sh <name> [-DFLAG=..]...  ->  tools/bin/libisx_<name>.so
# (run with ISX_LIB_PATH=tools/bin/libisx_<name>.so; ISX_KNN_DIAG builds give WRONG results by design)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" \
   -x hip image-stitcher_amd/csrc/*.hip -o tools/bin/libisx_$name.so
echo tools/bin/libisx_$name.so
