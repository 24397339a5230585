"""Host-inclusive rate of PerspectiveFields.inference_batch (numpy uint8 640x640 images in host memory -> result dicts),
single process, with the PIL resize on a host core vs the bit-identical resize on the GPU (N1).  Not the bench `value`."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perspectivefields_amd import PerspectiveFields
from perspectivefields_amd.synth import synthetic_image
m = PerspectiveFields("Paramnet-360Cities-edina-centered", weights="synthetic:0").eval().cuda()
B = 32
imgs = [synthetic_image(640, 640, 300 + i) for i in range(B)]
out = {}
for mode in (False, True):
    m.device_resize = mode
    for _ in range(3): m.inference_batch(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = m.inference_batch(imgs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    out["device_resize" if mode else "host_pil_resize"] = {"images_per_sec": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2)}
# ---- N3: three-stream pipeline, results delivered to pinned host memory (what the reference's callers do with .cpu())
def to_host_sync(res):
    return [{k: (v.cpu() if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in r.items()} for r in res]
for mode in (False, True):
    m.device_resize = mode
    nb = 6
    list(m.inference_stream([imgs] * 2))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nb): to_host_sync(m.inference_batch(imgs))
    torch.cuda.synchronize(); dt_sync = (time.perf_counter() - t0) / nb
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in m.inference_stream([imgs] * nb, to_host=True): pass
    torch.cuda.synchronize(); dt_pipe = (time.perf_counter() - t0) / nb
    out[("device_resize" if mode else "host_pil_resize") + "_fields_to_host"] = {
        "inference_batch_then_cpu_images_per_sec": round(B / dt_sync, 1), "inference_stream_images_per_sec": round(B / dt_pipe, 1)}
m.device_resize = False
out["note"] = "one Python process, pageable host memory, synchronous H2D; batch 32 of 640x640 uint8"
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/bench_e2e_host.json", "w"), indent=1); print(json.dumps(out, indent=1))
