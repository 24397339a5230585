#!/bin/bash
# r06 call D: Winograd half-patch geometry on the 40 x 40 maps -- op tests (fp64 parity, bit identity with the square-patch kernel), per-launch time, bench A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== wino op tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -s -k "winograd" 2>&1 | grep -E "^\[|passed|failed|FAILED|Error|error" | tail -40
echo "== e2e goldens + split"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_r06.py -q -m gpu -p no:cacheprovider -k "golden or stage3_batch_split or fused_block_mlps" 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/r06_d_wino_half.log
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
from perspectivefields_amd import ops
names = ops.conv_tiles(); t = names.index("wino256x64d")
for (B, H) in ((64, 40), (64, 80), (64, 24), (64, 56)):
    ms = min(ops.conv2d_bench(B, H, H, 256, 256, 3, 1, 1, tile=t, iters=20) for _ in range(3))
    print(f"PF_WINO_HALF={os.environ.get('PF_WINO_HALF','1')} B={B} {H}x{H} 256->256: {1e3*ms:.1f} us  {2.0*B*H*H*256*2304/ms/1e9:.1f} TF")
'''
for half in ("0", "1", "0", "1"):
    env = dict(os.environ, PF_WINO_HALF=half)
    subprocess.run([sys.executable, "-c", code], env=env)
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --events-in-timed 0"
for i in 1 2 3; do
  for m in 0 1; do echo -n "PF_WINO_HALF=$m: "; PF_WINO_HALF=$m timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee -a gpurun_out/r06_d_wino_half.log
