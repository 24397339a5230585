"""Aggregates rocprofv3 CSVs (kernel trace + PMC passes) per kernel name -> gpurun_out/rocprof_summary.{json,md}."""
import csv, glob, json, math, os, re, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.search(r"pf::(\w+)", name)
    if m:
        t = re.search(r"igemm(?:_sb)?_kernel<(\d+), (\d+), (\d+), (\d+)", name)
        if t:
            return f"pf::{m.group(1)}<{t.group(1)}x{t.group(2)},w{int(t.group(3))*int(t.group(4))}>"
        h = re.search(r"igemm_sbh_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+)(?:, (\w+))?>", name)  # trailing ASB (split-plane input) flag since r02
        if h:  # patch, BN, waves, mode (2 = concat), fused up-sampling
            return f"pf::igemm_sbh_kernel<{h.group(1)}x{h.group(2)},n{h.group(3)},w{int(h.group(4))*int(h.group(5))}" + (",cat" if h.group(6) == "2" else "") + (",ups" if h.group(10) == "true" else "") + (",planes" if h.group(11) == "true" else "") + ">"
        return f"pf::{m.group(1)}"
    return name[:60]

out = {"kernels": {}}
# ---- kernel trace: durations
for f in glob.glob(os.path.join(root, "rocprof_trace", "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[k][0] += 1; agg[k][1] += d
    tot = sum(v[1] for v in agg.values())
    for k, (n, us) in agg.items():
        out["kernels"].setdefault(k, {})["trace"] = {"calls": n, "total_us": round(us, 1), "avg_us": round(us / n, 2), "pct": round(100 * us / tot, 2)}
    out["trace_total_us"] = round(tot, 1)
# ---- PMC passes
for d in glob.glob(os.path.join(root, "rocprof_pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]; agg[k][c] += float(r["Counter_Value"]); cnt[k][c] += 1
        for k in agg:
            for c in agg[k]:
                out["kernels"].setdefault(k, {}).setdefault("pmc", {})[c] = {"sum": agg[k][c], "dispatches": cnt[k][c], "avg": agg[k][c] / max(cnt[k][c], 1)}
# derived: HBM traffic per launch (guide: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950; FETCH_SIZE under-reports wide reads by 2x)
for k, v in out["kernels"].items():
    p = v.get("pmc", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        v["hbm_bytes_per_launch"] = (2 * p["FETCH_SIZE"]["avg"] + p["WRITE_SIZE"]["avg"]) * 1024
    if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
        h, m = p["TCC_HIT_sum"]["sum"], p["TCC_MISS_sum"]["sum"]
        v["l2_hit_rate"] = h / max(h + m, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in p and p["GRBM_GUI_ACTIVE"]["sum"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES: 64 per v_mfma_f32_32x32x2_f32, summed over all SIMDs
        v["mfma_busy_frac"] = p["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (p["GRBM_GUI_ACTIVE"]["sum"] / 8.0 * 1024)  # GRBM counter is summed over the 8 XCDs; 1024 SIMDs
# class aggregate matching bench.py's `roofline` object (all split-bf16 implicit-GEMM launches: linear + halo tiles)
cls = [v["trace"] for k, v in out["kernels"].items() if (k.startswith("pf::igemm_sb") or k.startswith("pf::cnx_mlp") or k.startswith("pf::wino")) and "trace" in v]  # igemm_sb_kernel<*>, igemm_sbh_kernel<*>, cnx_mlp_kernel, the Winograd kernels
if cls:
    n, us = sum(t["calls"] for t in cls), sum(t["total_us"] for t in cls)
    out["split_bf16_igemm_class"] = {"calls": n, "total_us": round(us, 1), "avg_us": round(us / n, 2),
                                     "pct": round(100 * us / out.get("trace_total_us", us), 2)}
# ---- per launch SHAPE of the 3x3 halo kernels (kernel name + grid size): the dominant shape of the forward gets its own HBM-traffic row
shapes = {}
for f in glob.glob(os.path.join(root, "rocprof_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm_sbh" not in r["Kernel_Name"] and "wino" not in r["Kernel_Name"]: continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        # launches of one kernel + grid can still differ in K (e.g. 256 -> 256 and the folded 64 -> 256 conv at 80^2): split by duration octave
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(round(math.log2(max(us, 1.0)))))
        e = shapes.setdefault(k, {"calls": 0, "us": 0.0})
        e["calls"] += 1; e["us"] += us
for d in glob.glob(os.path.join(root, "rocprof_pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm_sbh" not in r["Kernel_Name"] and "wino" not in r["Kernel_Name"]: continue
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(round(math.log2(max(us, 1.0)))))
            e = shapes.setdefault(k, {"calls": 0, "us": 0.0}).setdefault("pmc", {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
rows_s = []
for (k, grid, octave), v in shapes.items():
    if not v["calls"]: continue  # a duration octave seen only under the counters (profiled runs clock a little lower)
    p = {c: s / max(n, 1) for c, (s, n) in v.get("pmc", {}).items()}
    row = {"kernel": k, "grid": grid, "duration_octave_us": 2 ** octave, "calls": v["calls"], "avg_us": round(v["us"] / max(v["calls"], 1), 2), "total_us": round(v["us"], 1)}
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p: row["hbm_bytes_per_launch"] = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p: row["l2_hit_rate"] = p["TCC_HIT_sum"] / max(p["TCC_HIT_sum"] + p["TCC_MISS_sum"], 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in p and p.get("GRBM_GUI_ACTIVE", 0) > 0: row["mfma_busy_frac"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    rows_s.append(row)
rows_s.sort(key=lambda r: -r["total_us"])
out["halo_kernel_by_shape"] = rows_s
if rows_s and "hbm_bytes_per_launch" in rows_s[0]:
    d0 = rows_s[0]
    json.dump({"kernel": d0["kernel"], "grid": d0["grid"], "calls_in_trace": d0["calls"], "avg_us": d0["avg_us"],
               "hbm_bytes_per_launch": d0["hbm_bytes_per_launch"], "l2_hit_rate": d0.get("l2_hit_rate"), "mfma_busy_frac": d0.get("mfma_busy_frac"),
               "shape": "dominant launch shape of the forward by total time (3x3 256->256 @80^2, both decoder heads in one grouped launch, B=32; since r05: the Winograd kernel)",
               "algorithmic_bytes_per_launch": {"input": 2 * 32 * 80 * 80 * 256 * 4, "output": 2 * 32 * 80 * 80 * 256 * 4,
                                                "residual_operands": "0, 1 or 2 x the output size (4 launches per step: none / res1+res2 / none / res1): 0.84 - 1.68 GB, mean 1.15 GB"},
               "commit": os.environ.get("PF_EVIDENCE_COMMIT"),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py (B=32, steady state, shipped tile table), rows of this kernel + grid size only; "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request)"},
              open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "rocprof_summary.json"), "w"), indent=1)
rows = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("trace", {}).get("total_us", 0))
md = ["| kernel | calls | total us | avg us | % | HBM MB/launch | L2 hit | MFMA busy |", "|---|---|---|---|---|---|---|---|"]
for k, v in rows[:40]:
    t = v.get("trace", {})
    md.append(f"| {k} | {t.get('calls','')} | {t.get('total_us','')} | {t.get('avg_us','')} | {t.get('pct','')} | "
              f"{v.get('hbm_bytes_per_launch', 0)/1e6:.1f} | {v.get('l2_hit_rate', float('nan')):.3f} | {v.get('mfma_busy_frac', float('nan')):.3f} |")
if "split_bf16_igemm_class" in out:
    c = out["split_bf16_igemm_class"]
    md.append("")
    md.append(f"split (fp16 / bf16) implicit GEMM as one class (pf::igemm_sb_kernel<*> + pf::igemm_sbh_kernel + pf::cnx_mlp_kernel = bench.py's `split_gemm_class`): "
              f"{c['calls']} calls, {c['total_us']} us, avg {c['avg_us']} us per launch, {c['pct']} % of kernel time")
md.append("")
md.append("3x3 halo / Winograd kernels by launch shape (kernel, grid size):")
md.append("| kernel | grid | calls | avg us | HBM MB/launch | L2 hit | MFMA busy |")
md.append("|---|---|---|---|---|---|---|")
for r in rows_s[:12]:
    md.append(f"| {r['kernel']} | {r['grid']} | {r['calls']} | {r['avg_us']} | {r.get('hbm_bytes_per_launch', 0)/1e6:.1f} | {r.get('l2_hit_rate', float('nan')):.3f} | {r.get('mfma_busy_frac', float('nan')):.3f} |")
open(os.path.join(root, "rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
