"""Static resource table of every gfx950 kernel in libpf_hip.so (no GPU needed): VGPRs, spills, scratch, LDS.
Reads the clang offload bundles of the .hip_fatbin section and the AMDGPU metadata notes of each code object."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "copy.so")], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(lib):
    rows = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: int((re.search(k + r":\s+(\d+)", blk) or [0, "0"])[1])
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            dem = re.sub(r"\(.*$", "", dem).replace("void ", "")
            rows.append({"kernel": dem, "vgpr": g(r"\.vgpr_count"), "agpr": int(blk.split()[0]), "spill": g(r"\.vgpr_spill_count"),
                         "scratch": g(r"\.private_segment_fixed_size"), "lds": g(r"\.group_segment_fixed_size"), "max_threads": g(r"\.max_flat_workgroup_size")})
    return sorted(rows, key=lambda r: r["kernel"])


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "perspectivefields_amd", "lib", "libpf_hip.so")
    rows = kernels(lib)
    print(f"{len(rows)} gfx950 kernels in {os.path.relpath(lib, ROOT)}")
    print(f"{'vgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7} {'thr':>5}  kernel")
    for r in rows:
        print(f"{r['vgpr']:5d} {r['spill']:5d} {r['scratch']:7d} {r['lds']:7d} {r['max_threads']:5d}  {r['kernel']}")
