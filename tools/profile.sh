#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then separate PMC passes.
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag>
# The counter passes are restricted to this library's kernels (--kernel-include-regex): with counters on every dispatch rocprofv3
# aborts inside rocSOLVER's syevd, which bench.py now runs to make the projectors (SURVEY 8d pipeline).
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 10 --warmup 2 --hot-path-only"
echo "trace pass" >> gpurun_out/profile_progress.log
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/trace.log
echo "trace pass (training step)" >> gpurun_out/profile_progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_e2e -o trace -- python3 bench.py --steps 10 --warmup 3 --no-extras > $OUT/trace_e2e.log 2>&1 || echo "trace_e2e failed" >> $OUT/trace_e2e.log
echo "pmc pass" >> gpurun_out/profile_progress.log
timeout -k 10 170 rocprofv3 --kernel-include-regex "nsgp|repre" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -o pmc -- $B > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed" >> $OUT/pmc_sq.log
echo "pmc pass" >> gpurun_out/profile_progress.log
timeout -k 10 170 rocprofv3 --kernel-include-regex "nsgp|repre" --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_lds -o pmc -- $B > $OUT/pmc_lds.log 2>&1 || echo "pmc_lds failed" >> $OUT/pmc_lds.log
echo "pmc pass" >> gpurun_out/profile_progress.log
timeout -k 10 170 rocprofv3 --kernel-include-regex "nsgp|repre" --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $B > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed" >> $OUT/pmc_fetch.log
echo "pmc pass" >> gpurun_out/profile_progress.log
timeout -k 10 170 rocprofv3 --kernel-include-regex "nsgp|repre" --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o pmc -- $B > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed" >> $OUT/pmc_write.log
python3 tools/summarize_prof.py $OUT gpurun_out/summary_$TAG > $OUT/summary.log 2>&1
cp $OUT/*.log gpurun_out/summary_$TAG/ 2>/dev/null
rm -rf $OUT
