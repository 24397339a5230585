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


_PK_SRC1_HI = re.compile(r"v_pk_(?:fma|mul|add)_f32 (v\[\d+:\d+\]), (\S+), (\S+?)(?:,| ).*op_sel:\[(\d),(\d)")


def packed_src1_high_forms(lib):
    """Packed-fp32 instructions whose LOW lane reads the HIGH half of src1 (op_sel[1] = 1), per kernel: [(mangled kernel name, instruction)] -- ANY such form,
    including the compiler's horizontal reductions `v_pk_add_f32 d, x, x op_sel:[0,1] op_sel_hi:[1,0]` (the same register pair twice).
    On MI355X these forms returned wrong low-lane results in the upper lanes of a wave while kernels of this library ran on the same CUs from another stream -- exact
    alone (profiles/r04_dw7_packed.md, scripts/microbench/pk_opsel_beside.hip: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; form 7 = the x, x reduction).  Since r05
    the kernels hipcc packed this way are compiled with PF_NO_PK_F32 (pf_kernels.h) and the library is expected to contain none."""
    hits = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            p = subprocess.Popen([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, text=True)
            cur = None
            for line in p.stdout:
                if line.endswith(">:\n"):
                    cur = line.split("<")[1][:-3]
                elif "v_pk_" in line and "op_sel:[" in line:
                    m = _PK_SRC1_HI.search(line)
                    if m and m.group(5) == "1":
                        hits.append((cur, line.split("//")[0].strip()))
            p.wait()
    return hits


_MFMA = re.compile(r"(v_mfma\S+|v_smfmac\S+)\s+([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],")
_VDEF = re.compile(r"^(v_\S+)\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def scan_valu_write_then_mfma_read(lines, states_needed=2):
    """The scan of valu_write_then_mfma_read() over the lines of one llvm-objdump -d listing."""
    hits = []
    cur, hist = None, []
    for line in lines:
        line = line.rstrip("\n")
        if line.endswith(">:"):
            cur, hist = line.split("<")[1][:-2], []
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        m = _MFMA.search(ins)
        if m:
            srcs = [(m.group(5), int(m.group(6)), int(m.group(7))), (m.group(8), int(m.group(9)), int(m.group(10)))]
            states = 0
            for prev in reversed(hist):
                if states >= states_needed:
                    break
                d = _VDEF.match(prev)
                if d and not prev.startswith(("v_mfma", "v_smfmac", "v_cmp", "v_nop")):
                    lo = int(d.group(2) or d.group(4)); hi = int(d.group(3) or d.group(4))
                    if any(s[0] == "v" and not (hi < s[1] or s[2] < lo) for s in srcs):
                        hits.append((cur, prev, ins, states))
                n = re.match(r"s_nop (\d+)", prev)
                states += int(n.group(1)) + 1 if n else 1
        hist.append(ins)
        del hist[:-8]
    return hits


def valu_write_then_mfma_read(lib, states_needed=2):
    """MFMAs that read, as their A or B operand, a VGPR a VALU instruction wrote fewer than `states_needed` wait states earlier, per kernel:
    [(mangled kernel name, the VALU instruction, the MFMA, states in between)].  hipcc pads this hazard for its own instructions and not for what an inline-asm
    statement writes (sb_split.h split_f16_mfma_pad; profiles/r06_asm_mfma_hazard.md: one state -- an s_waitcnt -- between a v_fma_mixhi_f16 and the MFMA gave the
    matrix core the register's old contents).  Every instruction counts one state, `s_nop N` counts N + 1; the library is expected to contain none."""
    hits = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            p = subprocess.Popen([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, text=True)
            hits += scan_valu_write_then_mfma_read(p.stdout, states_needed)
            p.wait()
    return hits


def blocks_per_cu(r):
    """Resident blocks per CU of a kernel: 512 VGPRs per SIMD lane in granules of 8 (at most 8 waves per SIMD), 4 SIMDs, 160 KB of LDS."""
    waves_per_simd = min(8, 512 // max(8, (r["vgpr"] + 7) // 8 * 8))
    by_regs = waves_per_simd * 4 // max(1, r["max_threads"] // 64)
    return min(by_regs, (160 * 1024) // r["lds"] if r["lds"] else by_regs)


def compare(old_lib, new_lib):
    """Kernels whose spills or resident blocks per CU differ between two builds (run after every change to a shared device function)."""
    a = {r["kernel"]: r for r in kernels(old_lib)}
    out = []
    for r in kernels(new_lib):
        o = a.get(r["kernel"])
        if o and (blocks_per_cu(o) != blocks_per_cu(r) or o["spill"] != r["spill"]):
            out.append(f"{o['vgpr']:4d} -> {r['vgpr']:4d} VGPRs  blocks/CU {blocks_per_cu(o)} -> {blocks_per_cu(r)}  spills {o['spill']} -> {r['spill']}  {r['kernel']}")
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--compare":  # kernel_resources.py --compare OLD.so [NEW.so]
        new = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "perspectivefields_amd", "lib", "libpf_hip.so")
        diff = compare(sys.argv[2], new)
        print("\n".join(diff) if diff else "no kernel changed its spills or its resident blocks per CU")
        sys.exit(0)
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "perspectivefields_amd", "lib", "libpf_hip.so")
    rows = kernels(lib)
    print(f"{len(rows)} gfx950 kernels in {os.path.relpath(lib, ROOT)}")
    print(f"{'vgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>7} {'thr':>5} {'blk/CU':>6}  kernel")
    for r in rows:
        print(f"{r['vgpr']:5d} {r['spill']:5d} {r['scratch']:7d} {r['lds']:7d} {r['max_threads']:5d} {blocks_per_cu(r):6d}  {r['kernel']}")
