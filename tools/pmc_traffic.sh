#!/bin/bash
# tools/pmc_traffic.sh: memory-side traffic of the count kernels on bench.py's workload (FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/r3_traffic/{fetch,write}/ + a summary.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_traffic
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $OUT/$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stitch > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $OUT/$c.log; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $OUT knn2_
