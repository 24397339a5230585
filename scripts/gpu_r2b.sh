#!/bin/bash
# Round 2, second GPU call: full suite with the new kernels (column-blocked dw7, split-f16 attention, double-buffered halo tiles),
# kernel sweeps, per-layer tables with / without tuning, batch-8 table (cache residency of the elementwise kernels).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -E "^\[attention|^\[split|passed|failed|Error|error|assert|FAILED" | tail -40 | tee gpurun_out/r2b_test_gpu.log
echo "== tune dw7"; TUNE_OUT=gpurun_out/r2b_tune_dw7.txt timeout 600 python scripts/tune_dw7.py 2>&1 | cut -c1-700 | tail -6
echo "== tune attn"; TUNE_OUT=gpurun_out/r2b_tune_attn.txt timeout 300 python scripts/tune_attn.py 2>&1 | tail -6
echo "== tune_conv 3x3 shapes (new halo tiles)"; TUNE_ONLY=rcu,fold_c1,fold_c2,conv TUNE_PREC=0 TUNE_OUT=gpurun_out/r2b_tune_conv_3x3.txt timeout 600 python scripts/tune_conv.py 2>&1 | sed 's/.*| auto/auto/' | cut -c1-900 | tail -12
echo "== layers table tiles"; timeout 300 python scripts/profile_layers.py --out gpurun_out/r2b_layers_table.txt 2>&1 | head -24
echo "== layers autotuned"; timeout 300 python scripts/profile_layers.py --autotune 1 --out gpurun_out/r2b_layers_tuned.txt 2>&1 | head -60
echo "== layers B=8"; timeout 300 python scripts/profile_layers.py --batch 8 --out gpurun_out/r2b_layers_b8.txt 2>&1 | grep -E "batch|layernorm|dwconv|upsample|attention" | head -40
echo "== bench autotuned"; timeout 600 python bench.py --steps 10 --warmup 3 --autotune 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r2b_bench_tuned.json | cut -c1-400
echo "== bench attn fp32 kernel"; PF_ATTN_VARIANT=0 PF_DW7_VARIANT=2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-200
