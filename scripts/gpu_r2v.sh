#!/bin/bash
# attention: query tiles per block (PF_ATTN_QT) sweep + tests + bench
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for QT in 1 2 4 0; do
echo "== PF_ATTN_QT=$QT"; PF_ATTN_QT=$QT TUNE_OUT=gpurun_out/r2v_tune_attn_qt$QT.txt timeout 200 python scripts/tune_attn.py 2>&1 | sed 's/fp32 MFMA.*| split/split/' | cut -c1-120
done
echo "== pytest attention"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "attention" 2>&1 | tail -3
echo "== bench"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | tee gpurun_out/r2v_bench.json | cut -c1-160
echo "== bench PF_ATTN_QT=1"; PF_ATTN_QT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
