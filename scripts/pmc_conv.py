"""Runs a few conv / GEMM shapes with given tiles (pf_op_conv2d_bench) -- the target of scripts/gpu_pmc_conv.sh (rocprofv3 --pmc).
usage: pmc_conv.py  name:B:H:W:Cin:Cout:K:stride:pad:tile  ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops
tiles = ops.conv_tiles()
for spec in sys.argv[1:]:
    name, b, h, w, cin, cout, k, st, pd, tile = spec.split(":")
    ms = ops.conv2d_bench(int(b), int(h), int(w), int(cin), int(cout), int(k), int(st), int(pd), tile=tiles.index(tile), iters=3)
    print(name, tile, ms)
