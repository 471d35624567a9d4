#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ISX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT 90 python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 --scenes 2 --imgs-per-scene 8 --kp 1024 > gpurun_out/dbg_rank$r.log 2>&1 &
done
wait
for r in 0 1; do echo "== rank $r"; grep -v "^  File \"/usr/l" gpurun_out/dbg_rank$r.log | tail -25; done
