#!/bin/bash
# Same-box A/B of the library in lib_prev/ (built from the previous sources: cp -r lib lib_prev before rebuilding) against lib/ at B = 32 (twice, alternating),
# 64 (the forward of the mixed-resolution workload), 16, 8 and 1; then the full GPU suite on lib/.  LABEL_PREV / LABEL_NEW name the two.
export TMPDIR=/tmp
mkdir -p gpurun_out
B="timeout 200 python bench.py --no-cpu-baseline --no-extras --events-in-timed 0"
for rep in 1 2; do
echo "== B=32 prev (${LABEL_PREV:-previous sources})"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100
echo "== B=32 new (${LABEL_NEW:-working tree})"; $B --steps 10 --warmup 3 2>&1 | tail -1 | cut -c60-100
done
for b in 64 16 8 1; do
  st=$((b == 1 ? 200 : (b == 64 ? 8 : 20)))
  echo "== B=$b prev"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100
  echo "== B=$b new"; $B --batch $b --steps $st --warmup 5 2>&1 | tail -1 | cut -c60-100
done
if [ -n "${AB_X6:-}" ]; then
echo "== bf16x6 prev"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B --steps 10 --warmup 3 --precision fp32_bf16x6 2>&1 | tail -1 | cut -c60-100
echo "== bf16x6 new"; $B --steps 10 --warmup 3 --precision fp32_bf16x6 2>&1 | tail -1 | cut -c60-100
fi
if [ -z "${AB_NO_TESTS:-}" ]; then timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3; fi
