#!/bin/bash
# r05 call 1: packed-fp32 src1-high forms.  (a) microbenchmark with form 7 (v_pk_add_f32 d, x, x -- hipcc's horizontal-sum form) alone and beside
# the forward; (b) same-box A/B of the library built WITHOUT the packed-fp32 feature in cnx_mlp / mit_mlp / rb_gemm / rb_chain (lib/) against the r04 library (lib_prev/);
# (c) the deferred-branch test looped 50x; (d) the full GPU suite.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== microbench"; timeout 400 python scripts/microbench/pk_opsel_beside.py 2>&1 | grep "form" | tee gpurun_out/r05_pk_opsel_beside.txt | cut -c1-220
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0 --steps 10 --warmup 3"
for rep in 1 2; do
echo "== B=32 r04 library"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B 2>&1 | tail -1 | cut -c60-100
echo "== B=32 no-packed-fp32 units"; $B 2>&1 | tail -1 | cut -c60-100
done 2>&1 | tee gpurun_out/r05_nopk_ab.log
echo "== deferred branch x50 (one process, both ParamNet architectures)"; timeout 600 python - <<'PY' 2>&1 | tail -3 | tee gpurun_out/r05_deferred_x50.log
import tests.test_gpu_e2e as t
for tag in ("centered", "uncentered"):
    for i in range(50):
        t.test_deferred_paramnet_branch_equals_joined_forward(tag)
    print(f"{tag}: 50 x test_deferred_paramnet_branch_equals_joined_forward passed (bit-identical to joined forwards)")
PY
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee gpurun_out/r05_test_gpu_call1.log
