#!/bin/bash
# tools/pmc_count16w.sh [flags]: rocprofv3 --pmc passes over the count kernel of get_collections on the workload of
# profiles/r01_knn2_count16_pmc.md (128 images x 4096 x 128-d, 16,256 directed items), one run per pass (the counters of a
# pass must fit the hardware: <= 8 SQ_, <= 2 GRBM_), plus one kernel trace.  ISX_COUNT_FLAGS selects the kernel
# (0 = knn2_count16w_kernel, 4 = knn2_count16_kernel, 2 = 32x32x32).  Output: gpurun_out/r3_pmc<flags>/<pass>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-0}
OUT=$R/gpurun_out/r3_pmc$F
export ISX_COUNT_FLAGS=$F
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o p --output-format csv -- python3 $R/tools/knn_pmc.py 128 2 count > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  grep "^count" $OUT/$name.log | tail -1
}
run sq1 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES || exit 1
run sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- python3 $R/tools/knn_pmc.py 128 4 count > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
grep "^count" $OUT/trace.log
cd $R
python3 tools/pmc_summary.py $OUT knn2_count
