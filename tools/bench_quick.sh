#!/bin/bash
# tools/bench_quick.sh [steps]: the N=1 all-pairs bench without its CPU baseline and stitch leg, key figures only
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python3 bench.py --no-stitch --no-cpu-baseline --steps ${1:-10} > gpurun_out/bq.json || exit 1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bq.json"))
print("pairs/s %.0f  ms_per_step %.2f  kernel_ms %.2f  frac %.4f  correct %s  host_ms_rank0 %.2f" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["components_correct"], d["step_breakdown"]["host_ms_rank0"]))
PY
