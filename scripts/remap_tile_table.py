"""Writes a copy of the shipped tile table with tile names replaced (tuning builds: run the whole forward on variant kernels).
  python scripts/remap_tile_table.py gpurun_out/tiles_v2.txt sbh128x64=sbhV2_128x64 sbh128x32=sbhV2_128x32 sbh128x128=sbhV2_128x128
then  PF_TUNING_BUILD=1 PF_TILE_TABLE=$PWD/gpurun_out/tiles_v2.txt python -m pytest tests -m gpu ...  /  python bench.py ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd.engine import TILE_TABLE

out = sys.argv[1]
mp = dict(a.split("=") for a in sys.argv[2:])
n = 0
lines = []
for line in open(TILE_TABLE):
    parts = line.rstrip("\n").split(" ")
    if parts and parts[-1] in mp:
        parts[-1] = mp[parts[-1]]
        n += 1
    lines.append(" ".join(parts))
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
print(f"{out}: {n} of {len(lines)} entries remapped")
