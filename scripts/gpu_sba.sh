#!/bin/bash
# Split-plane activation format: correctness of the new operand paths, tile sweep fp32-operand vs plane-operand, end-to-end A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ops (planes)"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "split or plane or all_tiles or epilogues" -p no:cacheprovider -x 2>&1 | tail -5
echo "== e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -12
echo "== tune sb"; TUNE_SB=1 TUNE_ONLY=${TUNE_ONLY:-rcu80,rcu20,fold_c4,conv0,conv1,pe3,s1_fc2,s1_sr,s2_fc1,s3_,s4_fc2,cnx0_pw2,cnx2,cnx3_pw2} timeout 900 python scripts/tune_conv.py 2>&1 | cut -c1-900
cp gpurun_out/tune_conv.txt gpurun_out/tune_conv_sba.txt
echo "== bench SBA=1"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | cut -c1-200
echo "== bench SBA=0"; PF_SBA=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-in-timed 0 2>&1 | tail -1 | cut -c1-200
echo "== layers SBA=1"; timeout 600 python scripts/profile_layers.py --out gpurun_out/layers_sba.txt 2>&1 | head -12
