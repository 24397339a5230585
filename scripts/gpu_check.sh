#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, short bench.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt
echo "== ops" ; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/test_ops.log
echo "== e2e" ; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/test_e2e.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 2>&1 | tail -5 | tee gpurun_out/bench.log
