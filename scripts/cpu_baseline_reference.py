"""bench.py's cpu_baseline() where the unmodified reference exists (the build container: kind = "reference"); the GPU box has no /root/reference and reports the port.
Output: profiles/r05_cpu_baseline_reference.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = {}
for kind_env in ("reference", "port"):
    if kind_env == "port":
        os.environ["PF_REFERENCE_ROOT"] = "/nonexistent"
        import importlib, oracle.ref_shim as rs
        importlib.reload(rs)
    out[kind_env] = bench.cpu_baseline("Paramnet-360Cities-edina-centered", 640, budget_s=20.0, max_images=8)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_cpu_baseline_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
