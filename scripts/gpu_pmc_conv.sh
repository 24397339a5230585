#!/bin/bash
# SQ counters of single conv / GEMM shapes (isolation, random data): where do the waves spend their cycles?
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
SHAPES="${PMC_SHAPES:-s3_qproj:1:12800:1:320:320:1:1:0:sb64x64f2 s3_fc2:1:12800:1:1280:320:1:1:0:sb128x64f2 s3_fc1:1:12800:1:320:1280:1:1:0:sb128x256w8 cnx0_pw1:1:204800:1:96:384:1:1:0:sb256x128w8 s4_fc2:1:3200:1:2048:512:1:1:0:sb128x64f2 rcu80:32:80:80:256:256:3:1:1:sbh256x64w8}"
rm -rf $R/gpurun_out/pmc_conv_*
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | cut -c1-20 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv_$tag -o c -- python $R/scripts/pmc_conv.py $SHAPES > $R/gpurun_out/pmc_conv_$tag.log 2>&1
  tail -2 $R/gpurun_out/pmc_conv_$tag.log | cut -c1-200
done
cd $R; python - <<'PY'
import csv, glob, collections, re
by = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmc_conv_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'igemm' not in r['Kernel_Name'] and 'wino' not in r['Kernel_Name']: continue
        k = re.sub(r'^void pf::', '', re.sub(r'\(.*$', '', r['Kernel_Name'])) + ' grid' + r['Grid_Size']
        by[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for k, d in by.items():
    a = {c: sum(v) / len(v) for c, v in d.items()}
    wc = a.get('SQ_WAVE_CYCLES', 0) or 1
    line = f"{k}: {sum(dur[k])/len(dur[k]):.1f} us | of wave cycles: wait_any {a.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {a.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active {a.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} (valu {a.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} lds {a.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} vmem {a.get('SQ_ACTIVE_INST_VMEM',0)/wc:.2f}) | "
    if a.get('GRBM_GUI_ACTIVE'): line += f"mfma_busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(a['GRBM_GUI_ACTIVE']/8*1024):.3f} waves {a.get('SQ_WAVES',0):.0f} L2hit {a.get('TCC_HIT_sum',0)/max(a.get('TCC_HIT_sum',0)+a.get('TCC_MISS_sum',0),1):.2f} "
    line += f"| insts valu {a.get('SQ_INSTS_VALU',0):.3g} lds {a.get('SQ_INSTS_LDS',0):.3g} vmem_rd {a.get('SQ_INSTS_VMEM_RD',0):.3g} lds_conflict/idx {a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}"
    out.append(line)
open('gpurun_out/pmc_conv_summary.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
