import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import ops
tiles = ops.conv_tiles()
for name in sys.argv[1:]:
    t = tiles.index(name)
    ms = ops.conv2d_bench(32, 80, 80, 256, 256, 3, 1, 1, tile=t, iters=3)
    print(name, ms)
