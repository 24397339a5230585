#!/bin/bash
# round 4, call 2: where the time of the row-block GEMM goes -- ablation forms + PMC counters of the isolated launches
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; export TMPDIR=/tmp
rm -f $R/gpurun_out/tune_rb.txt
{
for A in 0 1 2 3 4 5 7; do echo "== ABL $A"; PF_RB_ABL=$A RB_ONLY=1 timeout 120 python $R/scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids; done
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  RB_ONLY=1 TUNE_OUT=/dev/null timeout 120 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "rb_linear" --output-format csv -d $R/gpurun_out/pmcrb_$i -o rb -- python $R/scripts/tune_rb.py > $R/gpurun_out/pmcrb_$i.log 2>&1; tail -1 $R/gpurun_out/pmcrb_$i.log | cut -c1-100
done
cd $R
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/pmcrb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"])[:60] + " grid" + r.get("Grid_Size", "?") + " lds" + r.get("LDS_Block_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, v in sorted(agg.items()):
    print(k)
    print("   " + "  ".join(f"{c}={v[c]/max(n[k][c],1):.4g}" for c in sorted(v)) + f"  (per launch, {max(n[k].values())} launches)")
PY
} > $R/gpurun_out/rb2.log 2>&1
rm -rf $R/gpurun_out/pmcrb_*/
tail -80 $R/gpurun_out/rb2.log
