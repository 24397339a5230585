"""Generates the shipped tile table (perspectivefields_amd/tuned/gfx950_tiles.txt) on a real MI355X: pf_autotune for the
common batch sizes of every architecture and parity scheme, all choices written to one file (keyed by launch shape).

    python scripts/gen_tile_table.py --out gpurun_out/gfx950_tiles.txt      # on the GPU box; then copy into tuned/
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PF_TILE_TABLE"] = ""  # start from an empty table: every entry is measured in this run
from perspectivefields_amd import PerspectiveFields

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/gfx950_tiles.txt")
ap.add_argument("--batches", default="1,2,4,8,16,32,64")
ap.add_argument("--versions", default="Paramnet-360Cities-edina-centered,PersNet-360Cities,Paramnet-360Cities-edina-uncentered")
ap.add_argument("--precisions", default="fp32,fp32_bf16x6")
a = ap.parse_args()
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
lines = set()
for version in a.versions.split(","):
    for prec in a.precisions.split(","):
        if prec != "fp32" and version != "Paramnet-360Cities-edina-centered":
            continue
        m = PerspectiveFields(version, weights="synthetic:0", precision=prec).eval().cuda()
        eng = m._get_engine()
        for b in [int(v) for v in a.batches.split(",")]:
            t0 = time.time()
            eng.autotune(b, save_to=a.out + ".part")
            print(f"{version} {prec} B={b}: tuned in {time.time() - t0:.1f} s", flush=True)
        lines.update(open(a.out + ".part").read().splitlines())
        del m, eng
os.remove(a.out + ".part")
open(a.out, "w").write("\n".join(sorted(lines, key=lambda l: [int(v) for v in l.split()[:12]])) + "\n")
print(f"wrote {a.out}: {len(lines)} entries")
