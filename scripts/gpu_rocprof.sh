#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then PMC passes in their own runs (never combined
# with sys/hip/hsa tracing).  Raw outputs in gpurun_out/rocprof_*, summaries via scripts/summarize_rocprof.py.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
# tile choices come from the shipped table (no tuning launches inside the traces)
BENCH="python $R/bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
echo "== counters available"; rocprofv3 -L 2>/dev/null | grep -o -E "\b(FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU_MFMA_MOPS_F32|SQ_ACTIVE_INST_VALU|SQ_LDS_BANK_CONFLICT|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|MfmaUtil|TCP_TCC_READ_REQ_sum)\b" | sort -u | tr '\n' ' '; echo
echo "== kernel trace"; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rocprof_trace -o bench -- $BENCH --steps 3 --warmup 1 > $R/gpurun_out/rocprof_trace.log 2>&1; tail -1 $R/gpurun_out/rocprof_trace.log | cut -c1-200
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"; timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/rocprof_pmc_$tag -o bench -- $BENCH --steps 1 --warmup 1 > $R/gpurun_out/rocprof_pmc_$tag.log 2>&1; tail -1 $R/gpurun_out/rocprof_pmc_$tag.log | cut -c1-160
done
cd $R
find gpurun_out -name "*.csv" -size +30M -exec sh -c 'echo "dropping large $1"; rm "$1"' _ {} \;
python scripts/summarize_rocprof.py gpurun_out 2>&1 | tail -60
