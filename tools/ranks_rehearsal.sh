#!/bin/bash
# tools/ranks_rehearsal.sh: bench.py's multi-rank path on ONE GPU (ranks share cuda:0 and exchange through gloo: plumbing and
# attribution, not a scaling figure).  8 ranks on a small set, then 2 ranks on the bench set with the gather hidden behind the
# own-shard pairs (default) and without (ISX_BENCH_OVERLAP=0).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ISX_BENCH_BACKEND=gloo
show() { python3 -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
b=d['step_breakdown']
print(sys.argv[1], 'n_gpus', d['n_gpus'], 'ms_per_step %.1f' % d['ms_per_step'], 'correct', d['config']['components_correct'], '|', d['config']['parallelism'], '| allgather_ms', b['allgather_ms'], 'kernel_ms', b['kernel_ms'])
" $1; }
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 8 --steps 2 --warmup 1 --scenes 4 --imgs-per-scene 16 --kp 1024 > gpurun_out/reh8.json 2> gpurun_out/reh8.err || { tail -20 gpurun_out/reh8.err; exit 1; }
show gpurun_out/reh8.json
for ov in 1 0; do
  ISX_BENCH_OVERLAP=$ov timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2955$ov bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/reh2_$ov.json 2> gpurun_out/reh2_$ov.err || { tail -20 gpurun_out/reh2_$ov.err; exit 1; }
  show gpurun_out/reh2_$ov.json
done
