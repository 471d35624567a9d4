#!/bin/bash
# tools/ranks_dbg.sh [overlap=1] [bench args...]: two ranks of bench.py on one GPU over gloo, each under faulthandler with an
# ABRT time limit, so that a rank that stops shows where (gpurun_out/dbg_rank<r>.log)
cd ${GRAFT_REPO_ROOT:-/root/repo}
ov=${1:-1}; shift
export ISX_BENCH_BACKEND=gloo ISX_BENCH_OVERLAP=$ov MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT 150 python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 "$@" > gpurun_out/dbg_rank$r.log 2>&1 &
done
( for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do sleep 10; echo "t=$((i*10))s" >> gpurun_out/dbg_progress.log; done ) &
wait %1 %2
kill %3 2>/dev/null
for r in 0 1; do echo "== rank $r"; grep -v "^  File \"/usr/l" gpurun_out/dbg_rank$r.log | tail -30; done
