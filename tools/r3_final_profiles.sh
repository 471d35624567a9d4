#!/bin/bash
# tools/r3_final_profiles.sh: the round's bench line, the rocprofv3 kernel-trace summary of the same command, and the
# counter passes of the pair kernel (each pass its own run).  Output under gpurun_out/r3_final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"; head -c 600 $OUT/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $OUT/prof_bench.json 2> $OUT/prof_bench.err || { echo "profile failed"; tail -5 $OUT/prof_bench.err; exit 1; }
cd $R
db=$(ls $OUT/prof/*.db $OUT/prof/*/*.db 2>/dev/null | head -1)
echo "db: $db"
python3 tools/rocpd_stats.py $db $OUT/bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 3" | head -12
