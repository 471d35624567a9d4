#!/bin/bash
# tools/prof_variants.sh <filter> <variant>...: rocprofv3 kernel table rows matching <filter> for tools/bin/<variant>/img_bench
R=${GRAFT_REPO_ROOT:-/root/repo}
f=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $R/gpurun_out/r02i/$v
  timeout -k 10 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02i/$v -o p -- $R/tools/bin/$v/img_bench > /dev/null 2>&1 || exit 1
done
cd $R
for v in "$@"; do
  python tools/rocpd_stats.py $(find gpurun_out/r02i/$v -name "*.db" | head -1) gpurun_out/r02i/$v.md x > /dev/null
  echo "== $v"; grep "$f" gpurun_out/r02i/$v.md | cut -c1-60,95-150
done
