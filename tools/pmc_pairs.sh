#!/bin/bash
# tools/pmc_pairs.sh [lib]: rocprofv3 --pmc passes over knn2_pair_filter_kernel (tools/pairs_diag.py 128 1: 8128 pairs of
# 4096 x 4096 x 128-d), one run per pass, plus a kernel trace.  Output: gpurun_out/r3_pmc_pairs/<pass>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_pmc_pairs
[ -n "$1" ] && export ISX_LIB_PATH=$R/$1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
run() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p --output-format csv -- python3 $R/tools/pairs_diag.py 128 1 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  grep "pairs=" $OUT/$name.log | tail -1
}
run sq1 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES || exit 1
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
cd $R
python3 tools/pmc_summary.py $OUT knn2_pair
for p in sq1 sq2; do python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$OUT/$p/p_kernel_trace.csv')) if 'knn2_pair' in r['Kernel_Name']]
print('$p durations ms', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,2) for r in rows])
PY
done
