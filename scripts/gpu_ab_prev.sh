#!/bin/bash
# Same-box A/B of the library in lib_prev/ (built from the previous sources: cp -r lib lib_prev before rebuilding) against lib/.  LABEL_PREV / LABEL_NEW name the two.
export TMPDIR=/tmp
mkdir -p gpurun_out
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for rep in 1 2; do
echo "== prev (${LABEL_PREV:-previous sources})"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B 2>&1 | tail -1 | cut -c60-100
echo "== new (${LABEL_NEW:-working tree})"; $B 2>&1 | tail -1 | cut -c60-100
done
echo "== B=1 prev"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 timeout 200 python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c60-130
echo "== B=1 new"; timeout 200 python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --events-in-timed 0 2>&1 | tail -1 | cut -c60-130
if [ -n "${AB_X6:-}" ]; then
echo "== bf16x6 prev"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B --precision fp32_bf16x6 2>&1 | tail -1 | cut -c60-100
echo "== bf16x6 new"; $B --precision fp32_bf16x6 2>&1 | tail -1 | cut -c60-100
fi
timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_new.txt 2>&1 | sed -n 8,14p
PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_prev.txt 2>&1 | sed -n 8,14p
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3
