export TMPDIR=/tmp
B="timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for rep in 1 2; do
echo "== prev (swizzled A rows)"; PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 $B 2>&1 | tail -1 | cut -c60-100
echo "== new (linear 80-byte A rows)"; $B 2>&1 | tail -1 | cut -c60-100
done
timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_lin.txt 2>&1 | sed -n 8,14p
PF_LIB_SUFFIX=_prev PF_SKIP_DIGEST_CHECK=1 timeout 200 python scripts/profile_layers.py --out gpurun_out/layers_prev.txt 2>&1 | sed -n 8,14p
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3
