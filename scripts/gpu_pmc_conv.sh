#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $C | cut -c1-20 | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv_$tag -o c -- python $R/scripts/pmc_conv.py sb128x128 128x128p sb64x64 > $R/gpurun_out/pmc_conv_$tag.log 2>&1
done
cd $R; python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_conv_*/**/*counter_collection.csv', recursive=True)):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = 'sb' if 'igemm_sb' in r['Kernel_Name'] else ('f32' if 'igemm_kernel' in r['Kernel_Name'] else None)
        if not k: continue
        k += '_' + r['Kernel_Name'].split('<')[1].split(',')[0] + 'x' + r['Kernel_Name'].split(',')[1].strip()
        by[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in by.items():
        print(k, {c: f"{sum(v)/len(v):.3e}" for c, v in d.items()})
PY
