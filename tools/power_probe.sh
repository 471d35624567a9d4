#!/bin/bash
# tools/power_probe.sh <lib name>...: the pair kernel (recounts off) on SIFT-like, random and 0/1 bytes -- the same instruction
# stream and the same cycles of work; what differs is the power the matrix pipe draws and with it the clock the chip holds
for v in "$@"; do
  for data in sift low random; do
    echo "== $v $data"
    ISX_DIAG_DATA=$data ISX_LIB_PATH=tools/bin/libisx_$v.so timeout -k 10 120 python3 tools/pairs_diag.py 128 1 4096 2>&1 | grep "pairs=" || exit 1
  done
done
