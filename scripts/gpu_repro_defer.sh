#!/bin/bash
# Diagnosis / regression of the packed depthwise kernels beside this library's forward (profiles/r04_dw7_packed.md): the deferred-branch repro at B = 16 / 32 and the
# two GPU tests that pin it.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for cfg in "REPRO_B=16,16,5,16" "REPRO_B=32,32,32 REPRO_TRIALS=4"; do
  echo "== $cfg"; env REPRO_TRIALS=6 $cfg timeout 200 python scripts/repro_defer.py 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -x -q -k "deferred or beside or dwconv7x7" -p no:cacheprovider 2>&1 | tail -5
} > gpurun_out/repro_defer.log 2>&1
cat gpurun_out/repro_defer.log
