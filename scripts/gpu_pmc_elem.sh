#!/bin/bash
# PMC counters for the HBM-bound kernels (own runs, --kernel-trace only).
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
export PF_TUNE_CACHE=$R/gpurun_out/tune_cache.txt
python $R/bench.py --no-cpu-baseline --events-in-timed 0 --steps 1 --warmup 1 > /dev/null 2>&1
BENCH="python $R/bench.py --no-cpu-baseline --events-in-timed 0"
echo "== counters available"; rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_WAVES|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INST_CYCLES_VMEM|SQ_INSTS_VMEM_RD|SQ_WAIT_ANY|SQ_THREAD_CYCLES_VALU|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TCC_READ_REQ_sum|TCP_PENDING_STALL_CYCLES_sum|TCP_TA_TCP_STATE_READ_sum|TCP_GATE_EN1_sum|TCP_TCC_READ_REQ_LATENCY_sum|TA_BUSY_avr|TA_TA_BUSY_sum|TCP_READ_TAGCONFLICT_STALL_CYCLES_sum|SQ_INSTS_VALU_ADD_F32|SQ_INSTS_VALU_FMA_F32)\b" | sort -u | tr '\n' ' '; echo
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "== pmc $C"; timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcelem_$i -o bench -- $BENCH --steps 1 --warmup 1 > $R/gpurun_out/pmcelem_$i.log 2>&1; tail -1 $R/gpurun_out/pmcelem_$i.log | cut -c1-160
done
cd $R
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob("gpurun_out/pmcelem_*/**/*counter_collection.csv", recursive=True):
    seen = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k)[:48]
        if "igemm" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
print("kernel," + ",".join(names))
for k, v in sorted(agg.items()):
    print(k + "," + ",".join(f"{v.get(c, 0):.4g}" for c in names))
PY
find gpurun_out -name "*.csv" -size +30M -delete
