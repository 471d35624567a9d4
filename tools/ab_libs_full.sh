#!/bin/bash
# tools/ab_libs_full.sh <n_images> <lib name>...: the whole pair path (pair kernel + overflow stage, tools/pairs_ab.py) with each
# tools/bin/libisx_<name>.so ("product" = the in-tree library), round robin, two rounds, on one box
n=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then unset ISX_LIB_PATH; else export ISX_LIB_PATH=tools/bin/libisx_$v.so; fi
    echo "== $v"; timeout -k 10 200 python3 tools/pairs_ab.py $n 1 4096 2>&1 | grep "pairs=" | tail -2 || exit 1
  done
done
