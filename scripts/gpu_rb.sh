#!/bin/bash
# Row-block layers of MiT stage 3 (rb_gemm.hip / rb_chain.hip) on one box.  RB_DO = space-separated list of:
#   tests   op parity of rb_linear / rb_srkv (tests/test_gpu_ops.py)
#   time    isolated launch times against the best LDS tile per shape (scripts/tune_rb.py)
#   stamps  s_memtime timeline of one block (PF_RB_STAMPS=1)
#   ablate  timing-only ablation forms (PF_RB_ABL: 1 no weight refills, 2 no MFMAs, 3 neither, 7 neither and no fragment reads)
#   pmc     SQ / TCP / TCC counters of the isolated launches (separate --pmc passes)
#   mask    same-box A/B of PF_RB_CHAIN masks in bench.py (MASKS="0 28 60")
#   layers  per-shape layer table of the forward with mask 0 and 60
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; export TMPDIR=/tmp
rm -f $R/gpurun_out/tune_rb.txt
{
for what in ${RB_DO:-tests time mask}; do
  echo "==== $what"
  case $what in
    tests) timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "rb_linear or rb_srkv" 2>&1 | tail -5 ;;
    time) timeout 300 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids ;;
    stamps) PF_RB_STAMPS=1 RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids | grep -v "wave [123]" ;;
    ablate) for A in 0 1 2 3 7; do echo "== ABL $A"; PF_RB_ABL=$A RB_ONLY=1 timeout 120 python scripts/tune_rb.py 2>&1 | grep -v amdgpu.ids; done ;;
    pmc)
      cd /tmp; i=0
      for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
        i=$((i+1))
        RB_ONLY=1 TUNE_OUT=/dev/null timeout 120 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "rb_" --output-format csv -d $R/gpurun_out/pmcrb_$i -o rb -- python $R/scripts/tune_rb.py > $R/gpurun_out/pmcrb_$i.log 2>&1
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
      rm -rf $R/gpurun_out/pmcrb_*/ ;;
    mask) for rep in 1 2 3; do for rb in ${MASKS:-0 28 60}; do PF_RB_CHAIN=$rb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb', $rb, d['value'], d['ms_per_step'])"; done; done ;;
    layers) for rb in 0 60; do echo "== layers PF_RB_CHAIN=$rb"; PF_RB_CHAIN=$rb timeout 300 python scripts/profile_layers.py --batch 32 --out gpurun_out/layers_rb$rb.txt 2>&1 | grep "M=   12800 N=\|M=    3200 N=\|total\|layernorm  \|attention  "; done ;;
  esac
done
} > $R/gpurun_out/rb.log 2>&1
tail -60 $R/gpurun_out/rb.log
