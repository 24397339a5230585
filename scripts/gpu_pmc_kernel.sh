#!/bin/bash
# PMC counters for kernels matching $PMC_KERNEL (regex), in separate passes (--kernel-trace only).
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --no-cpu-baseline --events-in-timed 0 --steps 1 --warmup 1 > /dev/null 2>&1
BENCH="python $R/bench.py --no-cpu-baseline --events-in-timed 0"
K=${PMC_KERNEL:-dwconv7x7}
rm -rf $R/gpurun_out/pmck_*
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_GATE_EN1_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/pmck_$i -o bench -- $BENCH --steps 1 --warmup 0 > $R/gpurun_out/pmck_$i.log 2>&1; tail -1 $R/gpurun_out/pmck_$i.log | cut -c1-120
done
cd $R
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k)[:40] + " grid" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, v in sorted(agg.items()):
    print(k)
    print("   " + "  ".join(f"{c}={v[c]/max(n[k][c],1):.4g}" for c in sorted(v)) + f"  (per launch, {max(n[k].values())} launches)")
PY
