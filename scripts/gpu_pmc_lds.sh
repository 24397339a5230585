#!/bin/bash
# LDS bank-conflict counters of the implicit-GEMM kernels (one --pmc pass, --kernel-trace only).
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
export PF_TUNE_CACHE=$R/gpurun_out/tune_cache.txt
python $R/bench.py --no-cpu-baseline --events-in-timed 0 --steps 1 --warmup 1 > /dev/null 2>&1
rm -rf $R/gpurun_out/pmclds
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --kernel-include-regex "igemm" --output-format csv -d $R/gpurun_out/pmclds -o bench -- python $R/bench.py --no-cpu-baseline --events-in-timed 0 --steps 1 --warmup 0 > $R/gpurun_out/pmclds.log 2>&1
tail -1 $R/gpurun_out/pmclds.log | cut -c1-120
cd $R
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/pmclds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k)
        m = re.search(r"igemm(?:_sb|_sbh)?_kernel<(\d+), (\d+), (\d+), (\d+)", k)
        k = re.match(r"pf::\w+", k).group(0) + (f"<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}>" if m else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
lines = ["| kernel | launches | SQ_LDS_BANK_CONFLICT | SQ_LDS_IDX_ACTIVE | conflict / active | SQ_INSTS_LDS |", "|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    bc, act = v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0)
    lines.append(f"| {k} | {max(n[k].values())} | {bc:.4g} | {act:.4g} | {bc / max(act, 1):.4f} | {v.get('SQ_INSTS_LDS', 0):.4g} |")
open("gpurun_out/pmc_lds.md", "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
find gpurun_out/pmclds -name "*.csv" -size +30M -delete
