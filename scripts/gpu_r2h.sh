#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest ops"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
echo "== tune_conv (split tiles)"; TUNE_ONLY=s1_,s2_,s3_,s4_,cnx,conv,rcu,fold,pe TUNE_PREC=0 TUNE_OUT=gpurun_out/r2h_tune_conv.txt timeout 900 python scripts/tune_conv.py 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15}' | tail -45
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
echo "== bench autotuned"; timeout 600 python bench.py --steps 10 --warmup 3 --autotune 1 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c1-160
