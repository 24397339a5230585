import torch, math, sys
sys.path.insert(0, "/root/repo")
from perspectivefields_amd import ops
names = ops.conv_tiles()
tc, tw = names.index("wino256x64c"), names.index("wino256x64w4")
def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * scale
for (B, H, W, Cin, Cout) in [(1, 16, 16, 32, 64), (1, 16, 16, 64, 64), (1, 16, 16, 256, 64), (1, 80, 80, 256, 256), (2, 40, 40, 256, 256)]:
    x = rnd((B, H, W, Cin), 1).cuda(); w = rnd((Cout, Cin, 3, 3), 2, 1.0 / math.sqrt(Cin * 9)); b = rnd((Cout,), 3, 0.1)
    for rep in range(2):
        yc = ops.conv2d(x, w, b, pad=1, tile=tc, splitk=False).cpu(); yw = ops.conv2d(x, w, b, pad=1, tile=tw, splitk=False).cpu()
        d = (yc - yw).abs()
        bad = d > 1e-3
        print(f"B{B} {H}x{W} Cin{Cin} Cout{Cout} rep{rep}: max|d| {d.max():.3e}  bad frac {bad.float().mean():.4f}")
        if bad.any():
            idx = bad.nonzero()
            print("   bad by (y%16):", torch.bincount(idx[:, 1] % 16, minlength=16).tolist())
            print("   bad by (x%16):", torch.bincount(idx[:, 2] % 16, minlength=16).tolist())
            print("   bad by (c%64)//4:", torch.bincount((idx[:, 3] % 64) // 4, minlength=16).tolist())
            print("   bad by patch:", torch.bincount((idx[:, 1] // 16) * 8 + idx[:, 2] // 16).tolist())
            break
