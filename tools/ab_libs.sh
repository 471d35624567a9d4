#!/bin/bash
# tools/ab_libs.sh <n_images> <lib name>...: the pair kernel alone (tools/pairs_diag.py) with each tools/bin/libisx_<name>.so,
# round robin, three rounds, on one box -- the same-box A/B of compile-time variants (tools/build_variant.sh)
n=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    ISX_LIB_PATH=tools/bin/libisx_$v.so timeout -k 10 120 python3 tools/pairs_diag.py $n 1 4096 2>&1 | grep "pairs=" | tail -1 || exit 1
  done
done
