"""bench.py -- images/sec of the PerspectiveFields hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch: 320x320 uint8 network inputs already resident
in HBM -> MiT-B3 + both decoders + ParamNet (pf_forward_u8) -> post-process of every image to its
original 640x640 size (pf_postprocess_batch) -> per-image ParamNet scalars (+ all-gather of the
scalars over RCCL when N > 1).  Workload = BASELINE.json configs[2]: batch 32, 640x640,
Paramnet-360Cities-edina-centered, random-init (seeded synthetic) weights, synthetic images.
The reference's host-side PIL resize (perspectivefields.py:201) happens before the timed region.
Reported next to `value` (never as `value`):
  * `with_device_resize`: the same step starting from the ORIGINAL 640x640 uint8 images resident in HBM
    (bit-exact device PIL resize, pf_resize_batch_u8, inside the timed region);
  * `latency_ms`: batch-1 / batch-8 latency of the hot path (device resident) and of inference() / inference_batch()
    from host numpy images;
  * `parity`: one image of the LAST timed step checked against the CPU oracle after the timed loop.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: re-launches itself as N ranks, one per GPU, under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --workload mixed ...  (BASELINE configs[4]: mixed-resolution stream, bucketed resize-in / post-process-out, per-bucket images/sec)
    (--dry-run-cpu: the same control flow on CPU with a stub engine and the gloo backend -- tests/test_bench_dist_gloo.py)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 dense MFMA peak (spec; 2:1 sparsity NOT counted)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
GFLOP_PER_IMAGE_REF = 213.44   # contraction FLOPs of the reference graph, Paramnet-centered
MFMA_PER_PRODUCT = {"fp32": 3, "fp32_bf16x6": 6}  # (the 3-term bf16 split "bf16x3" is no bench mode any more: reduced precision AND slower than the fp32-class default, profiles/r03_bench_configs.json)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640, help="original image height = width")
    ap.add_argument("--version", default="Paramnet-360Cities-edina-centered")
    ap.add_argument("--precision", default="fp32", choices=list(MFMA_PER_PRODUCT),
                    help="arithmetic of the dense contractions; fp32 (split-f16, fp32-class accuracy) is the parity mode and the headline; "
                         "fp32_bf16x6 = exact bf16 split (no range window); no reduced-precision mode is offered")
    ap.add_argument("--autotune", type=int, default=0, help="1: time every tile configuration for this batch size before the warm-up (default: shipped tile table)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--defer-params", type=int, default=1, help="1 (default): the ParamNet branch of step i runs on the engine's own stream next to step i + 1's backbone (pf_set_defer_params); "
                    "every step's work still completes inside the timed region (join + device synchronize before the clock stops)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the device-resize figure, the latency numbers and the parity check")
    ap.add_argument("--events-in-timed", type=int, default=1, help="bracket the launches of >= 200 GFLOP (the dominant decoder convs) with HIP events inside the timed region (every timed step; "
                                                                   "negligible overhead) and take the per-class profile in one extra step after it; 0: no events at all")
    ap.add_argument("--workload", default="fixed", choices=("fixed", "mixed"),
                    help="fixed: BASELINE configs[2] (the headline: --batch images of --size x --size per GPU); mixed: configs[4], a stream of 384x512 / 640x640 / 1024x1365 "
                         "originals (1:2:1) resident in HBM, sharded round-robin within each (H, W) bucket, device resize + forward + post-process per step, per-bucket images/sec")
    ap.add_argument("--dry-run-cpu", action="store_true", help="control-flow test: stub engine on CPU, gloo backend (no GPU, no numbers)")
    return ap.parse_args(argv)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_traffic.json, produced by scripts/gpu_rocprof.sh + scripts/summarize_rocprof.py)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT) + (f" (taken at commit {d['commit']})" if d.get("commit") else "")


def cpu_baseline(version, size, budget_s=12.0, max_images=12):
    """The reference's CPU path timed on this host's cores on a bounded sample of the same workload.  kind = "reference": the UNMODIFIED reference
    (/root/reference through oracle/ref_shim.py: stubs for its import-only dependencies) whenever that tree exists -- the build container; kind = "port": the oracle
    (oracle/pf_oracle.py, pinned to the reference by tests/test_oracle_vs_reference.py and the goldens) on the GPU box, where the reference cannot travel."""
    from oracle import pf_oracle, ref_shim
    from perspectivefields_amd.config import arch_of, get_cfg
    from perspectivefields_amd.synth import synthetic_image, synthetic_state_dict, to_torch

    sd = to_torch(synthetic_state_dict(version, 0))
    arch = arch_of(get_cfg(version))
    ncpu = os.cpu_count() or 1
    kind = "port"
    run = lambda imgs: pf_oracle.inference_batch(sd, arch, imgs)
    what = "oracle/pf_oracle.py inference_batch"
    if ref_shim.reference_available():
        try:
            ref_model = ref_shim.build_reference(version, sd)
            run = lambda imgs: ref_model.inference_batch(imgs)
            kind, what = "reference", "the unmodified reference's PerspectiveFields.inference_batch (CPU, through oracle/ref_shim.py)"
        except Exception:  # the shim failing must not hide the measurement: fall back to the port
            pass
    with torch.no_grad():
        # torch's intra-op pool does not scale to hundreds of threads on these small maps: pick the
        # fastest of a few thread counts on one image each (reported as `cores`)
        best, threads = None, 1
        for t in sorted({min(ncpu, c) for c in (16, 32, 64)}):
            torch.set_num_threads(t)
            run([synthetic_image(size, size, 899)])  # warm-up at this setting
            t1 = time.perf_counter()
            run([synthetic_image(size, size, 900)])
            d = time.perf_counter() - t1
            if best is None or d < best:
                best, threads = d, t
        torch.set_num_threads(threads)
        n, t0 = 0, time.perf_counter()
        while n < max_images and time.perf_counter() - t0 < budget_s:
            run([synthetic_image(size, size, 901 + n + i) for i in range(2)])
            n += 2
        dt = time.perf_counter() - t0
    return {
        "value": round(n / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "host_logical_cpus": ncpu,
        "cores_note": f"`cores` = torch intra-op threads actually used (the fastest of 16 / 32 / 64 on one image each); the box has {ncpu} logical CPUs",
        "kind": kind,
        "sample": f"{n} synthetic {size}x{size} images through {what} (PIL resize + fp32 forward + post-process), batches of 2, {dt:.1f} s",
    }


def parity_check(version, resized_u8, hw, out, index):
    """One image of the last timed step against the CPU oracle (same 320x320 uint8 input, same synthetic checkpoint)."""
    from oracle import pf_oracle
    from perspectivefields_amd.config import arch_of, get_cfg
    from perspectivefields_amd.synth import synthetic_state_dict, to_torch

    pg, pl, outs, params = out
    sd = to_torch(synthetic_state_dict(version, 0))
    arch = arch_of(get_cfg(version))
    with torch.no_grad():
        ref = pf_oracle.forward(sd, arch, resized_u8[index:index + 1], [tuple(hw)])[0]
    up, lat = outs[index]
    g, go = up.double().cpu(), ref["pred_gravity_original"].double()
    cosv = float((1.0 - (g * go).sum(0) / torch.sqrt((g * g).sum(0) * (go * go).sum(0))).max())
    lat_l1 = float((lat.double().cpu() - ref["pred_latitude_original"].double()).abs().mean())
    res = {"image_index": index, "up_1_minus_cos_max": cosv, "latitude_l1_deg": lat_l1, "tolerances": {"up_1_minus_cos": 1e-3, "latitude_l1": 1e-3, "paramnet": 1e-4}}
    ok = cosv <= 1e-3 and lat_l1 <= 1e-3
    if params is not None:
        keys = ("pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal")
        dpar = max(abs(float(params[index, j]) - float(ref[k])) for j, k in enumerate(keys))
        res["paramnet_max_abs_delta"] = dpar
        ok = ok and dpar <= 1e-4
    res["ok"] = bool(ok)
    return res


class _StubEngine:
    """--dry-run-cpu: stands in for the HIP engine so that the rank logic (shards, barrier, MAX over ranks, gather of
    the scalars, rank-0-only JSON) can run under 2-process gloo on CPU.  Produces no numbers worth reading."""

    def __init__(self, rank):
        self.rank = rank

    def forward(self, batch):
        B = batch.shape[0]
        time.sleep(0.002 * (1 + self.rank))  # ranks finish at different times: exercises the MAX reduction
        params = (batch.reshape(B, -1)[:, :8].to(torch.float32) + self.rank).contiguous()
        return torch.zeros((B, 2, 4, 4)), torch.zeros((B, 1, 4, 4)), params

    def postprocess_batch(self, pg, pl, sizes):
        return [(torch.zeros((2, 2, 2)), torch.zeros((2, 2))) for _ in sizes]

    def set_defer_params(self, on):
        pass

    def join_params(self):
        pass

    def resize_batch_into(self, imgs, out):
        return out

    # the profiler calls of the measured path, so that the dry run walks the same control flow (a collective inside the rank-0-only extra step would hang it)
    PROFILE_CLASSES = ("igemm", "attention", "layernorm", "dwconv3x3_gelu", "dwconv7x7", "upsample2x", "other", "igemm_sb")

    def profile_begin(self, classes=None, large_only=False):
        pass

    def profile_end(self):
        return {n: {"ms": 0.0, "work": 0.0, "launches": 0} for n in self.PROFILE_CLASSES}

    def profile_records(self):
        return []

    def profile_phases(self):
        return {}


MIXED_PATTERN = [(384, 512), (640, 640), (640, 640), (1024, 1365)]  # BASELINE configs[4]: short edge 384 / 640 / 1024, mix 1:2:1


def self_launch(args, argv):
    """`python bench.py --gpus N` started as a plain process (no WORLD_SIZE): run the N ranks ourselves, one per GPU, under
    torch.distributed.run on 127.0.0.1 -- the JSON line then really describes N ranks.  Returns the launcher's exit code."""
    import socket
    import subprocess

    if not args.dry_run_cpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to report a {args.gpus}-GPU number from fewer devices", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *(sys.argv[1:] if argv is None else list(argv))]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # one rank per GPU is the contract: a line that says n_gpus = WORLD_SIZE while the caller asked for --gpus N would be a false curve
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}; launch with --nproc-per-node {args.gpus} (or without torchrun: bench.py launches the ranks itself)", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run_cpu
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)

    from perspectivefields_amd.dist import shard_round_robin_by_bucket
    from perspectivefields_amd.synth import synthetic_image

    precision = args.precision
    B, S = args.batch, args.size
    mixed = args.workload == "mixed"
    counts = None
    if mixed:
        # configs[4]: a global stream of B x world images over three (H, W) buckets; round-robin WITHIN each bucket gives every rank the same mix of
        # resize-in / post-process-out work (the network batch itself is resolution independent: everything becomes 320x320)
        sizes_global = [MIXED_PATTERN[i % len(MIXED_PATTERN)] for i in range(B * world)]
        shards = [shard_round_robin_by_bucket(sizes_global, r, world) for r in range(world)]
        counts = [len(sh) for sh in shards]
        sizes = [sizes_global[i] for i in shards[rank]]
        imgs = [synthetic_image(h, w, seed=7 + k) for k, (h, w) in enumerate(sorted(set(MIXED_PATTERN)))]
        by_size = dict(zip(sorted(set(MIXED_PATTERN)), imgs))
    else:
        # per-rank shard of the global batch: synthetic images (4 distinct ones tiled over the batch: the network is data independent), host resize outside the timed region
        imgs = [synthetic_image(S, S, seed=1000 + rank * B + i) for i in range(min(B, 4))]
        sizes = [(S, S)] * B
    Bl = len(sizes)  # images of this rank per step
    orig = None
    if dry:
        model, eng = None, _StubEngine(rank)
        t_resize = 0.0
        resized = np.stack([np.full((8, 8, 3), (rank * B + i) % 251, dtype=np.uint8) for i in range(Bl)])
    else:
        from perspectivefields_amd import PerspectiveFields

        model = PerspectiveFields(args.version, weights="synthetic:0", precision=precision).eval().to(dev)
        t_resize = time.perf_counter()
        resized4 = [model.aug.apply_image(im) for im in imgs]
        t_resize = (time.perf_counter() - t_resize) / len(imgs)
        if mixed:
            rs = dict(zip(sorted(set(MIXED_PATTERN)), resized4))
            resized = np.stack([rs[hw] for hw in sizes])
            dev_img = {hw: torch.from_numpy(im).to(dev) for hw, im in by_size.items()}
            orig = [dev_img[hw] for hw in sizes]  # the ORIGINAL uint8 images, resident in HBM
        else:
            resized = np.stack([resized4[i % len(resized4)] for i in range(B)])
        eng = model._get_engine()
        if args.autotune:
            eng.autotune(Bl)
    batch = torch.from_numpy(resized).to(dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        sync()

    # The step is the PRODUCT's sharded entry point (perspectivefields_amd.dist.ShardedPerspectiveFields.forward_step: device resize for the mixed stream, forward,
    # post-process, all-gather of the ParamNet rows).  Pipeline (--defer-params 1): forward i returns with its camera-parameter tensor still being computed on the
    # engine's stream, next to forward i + 1's backbone; the class gathers the rows of step i one step late and drain() joins the last branch and gathers the last
    # step's rows.  Nothing is skipped: K steps' work is inside the timed region.
    from perspectivefields_amd.dist import ShardedPerspectiveFields

    spf = ShardedPerspectiveFields(model, engine=eng)
    defer = bool(args.defer_params)  # also in the CPU dry run: the one-step-late gather of the scalars is rank logic the gloo test walks
    spf.set_pipeline(defer)
    local = {}   # references only (no GPU work): this rank's rows of the last two steps, for the bit-identity check after the timed region

    def step():
        o = spf.forward_step(batch, sizes, counts, originals=orig)
        if o.params is not None:
            local["overlapped"], local["last"] = local.get("last"), o.params
        return o.pred_gravity, o.pred_latitude, o.fields, o.gathered

    def drain(out):
        """the tail of the pipeline: the last step's rows (joined) replace the one-step-late ones"""
        last = spf.drain()
        return out if last is None else out[:3] + (last,)

    for _ in range(args.warmup):
        step()
    drain((None, None, None, None))
    barrier()
    # Per-launch HIP events INSIDE the timed region on the launches of >= 200 GFLOP only (the dominant decoder convs: ~6 per step), in every timed step: a dozen event
    # records per step cost nothing, so `value` and `roofline` come from the same steps.  The per-class figures (all split GEMMs, depthwise, LayerNorm, attention) need an
    # event pair around each of the ~330 launches of a step (+15 % step time, one stream): they are taken in ONE EXTRA step after the timed region.
    use_events = bool(args.events_in_timed) and not args.no_roofline
    if use_events:
        eng.profile_begin(classes=("igemm", "igemm_sb"), large_only=True)
    ev_steps = args.steps
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = step()
    out = drain(out)
    barrier()
    dt = time.perf_counter() - t0
    # Every step runs the same batch, so the scalars of a step whose ParamNet branch ran BESIDE the next step's backbone must equal, bit for bit, those of the last
    # step, whose branch ran alone after the join -- the parity check below only sees the last step (round 4: a packed-FMA operand form that was only wrong beside
    # other kernels went through exactly that gap, tests/test_gpu_e2e.py::test_deferred_paramnet_branch_equals_joined_forward caught it)
    deferred_identical = None
    if defer and not dry and local.get("overlapped") is not None and local.get("last") is not None and args.steps >= 2:
        deferred_identical = bool(torch.equal(local["overlapped"], local["last"]))
    if defer:
        spf.set_pipeline(False)  # everything after the timed region (extra profiled step, latency figures, parity check) reads its results right away
    if use_events:
        eng.profile_end()
    recs = eng.profile_records() if use_events else []
    prof = None
    if use_events and rank == 0:
        # rank 0 only, and WITHOUT the all-gather of the scalars: the other ranks are already past the timed region, a collective here would never complete
        eng.profile_begin(classes=("igemm", "igemm_sb", "dwconv3x3_gelu", "dwconv7x7", "upsample2x", "layernorm", "attention"))
        if orig is not None:
            eng.resize_batch_into(orig, batch)
        pg_x, pl_x, _ = eng.forward(batch)
        eng.postprocess_batch(pg_x, pl_x, sizes)
        sync()
        prof = eng.profile_end()
        cls_step_ms = sum(v["ms"] for v in prof.values())
        phases = eng.profile_phases()   # component split of the same profiled step (pf_profile_phases: event marks at the component boundaries)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total_images = (sum(counts) if mixed else B * world) * args.steps
    value = total_images / dt
    gathered_rows = int(out[3].shape[0]) if out[3] is not None else 0

    per_bucket = None
    if mixed:
        # per-bucket throughput: the forward is shared by all buckets (cost per image = forward / images of the rank); resize-in and post-process-out are timed
        # per (H, W) bucket on every rank, MAX over ranks
        def timed(fn, n=5):
            fn(); sync()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            sync()
            return (time.perf_counter() - t1) / n

        keys = sorted(set(MIXED_PATTERN))
        tv = [timed(lambda: eng.forward(batch))]
        nloc = []
        for hw in keys:
            idx = [i for i, t in enumerate(sizes) if t == hw]
            nloc.append(len(idx))
            if not idx:
                tv.append(0.0)
                continue
            ii = torch.tensor(idx, device=dev)
            pgb, plb, szb = out[0].index_select(0, ii), out[1].index_select(0, ii), [hw] * len(idx)
            if orig is not None:
                ob, u8b = [orig[i] for i in idx], torch.empty((len(idx),) + tuple(batch.shape[1:]), dtype=torch.uint8, device=dev)
                tv.append(timed(lambda: (eng.resize_batch_into(ob, u8b), eng.postprocess_batch(pgb, plb, szb))))
            else:
                tv.append(timed(lambda: eng.postprocess_batch(pgb, plb, szb)))
        tt = torch.tensor(tv, dtype=torch.float64, device=dev)
        if world > 1:
            import torch.distributed as dist

            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        fwd_img = float(tt[0]) / max(Bl, 1)
        per_bucket = {}
        for j, hw in enumerate(keys):
            n_glob = sum(1 for t in sizes_global if t == hw)
            io_img = float(tt[1 + j]) / max(nloc[j], 1)
            per_bucket[f"{hw[0]}x{hw[1]}"] = {
                "images_per_step_all_ranks": n_glob, "images_per_step_this_rank": nloc[j],
                "resize_plus_postprocess_us_per_image": round(io_img * 1e6, 1), "forward_us_per_image": round(fwd_img * 1e6, 1),
                "images_per_sec_all_ranks": round(world / (fwd_img + io_img), 1),
            }

    if rank != 0:
        if world > 1:
            import torch.distributed as dist

            dist.barrier()  # rank 0 may still be running its extras
            dist.destroy_process_group()
        return

    nt = MFMA_PER_PRODUCT[precision]
    line = {
        "metric": "images/sec (640x640, Paramnet-360Cities)",
        "value": round(value, 2),
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"fp32": "f32 (operands fp32 in HBM; contractions as 2-way fp16 split on the MFMA, fp32 accumulate, fp32-class accuracy)",
                  "fp32_bf16x6": "f32 (exact 3-way bf16 split on the MFMA, fp32 accumulate)"}.get(precision, precision),
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[4]: mixed-resolution stream, {B} images/GPU per step (384x512 : 640x640 : 1024x1365 = 1:2:1, one synthetic image per bucket), {args.version}; ORIGINAL "
                         "uint8 images resident in HBM -> bucketed bit-exact device resize -> forward at 320x320 -> post-process to each original size; sharded round-robin within each bucket"
                         ) if mixed else (
                         f"BASELINE configs[2]: batch {B}/GPU {S}x{S} {args.version}, fields + ParamNet, 320x320 network inputs resident in HBM (4 distinct synthetic images tiled over the batch; "
                         f"the network is data independent), post-process to {S}x{S}"),
            "global_batch": sum(counts) if mixed else B * world, "image_size": "mixed" if mixed else [S, S], "parallelism": f"dp{world} (images sharded, all-gather of ParamNet scalars)",
            "weights": "seeded synthetic checkpoint (no network for the trained .pth)", "precision": precision,
            "tiles": "autotuned on this device before the warm-up" if args.autotune else "shipped tile table + static heuristic (no tuning)",
            "gathered_param_rows": gathered_rows,
            "step_pipeline": ("ParamNet branch of step i on the engine's own stream beside step i + 1's backbone (pf_set_defer_params); all K steps complete inside the timed region"
                              if defer else "none: every forward joins its ParamNet branch before it returns"),
        },
    }
    if per_bucket is not None:
        line["per_bucket"] = per_bucket
    if dry:
        line["data"] = "dry run (stub engine on CPU, gloo): control flow only, numbers are meaningless"
    if prof is not None:
        traffic, traffic_src = pmc_traffic()
        ig, sb = prof["igemm"], prof["igemm_sb"]

        def mfma_obj(pr, kernel, peak, basis, executed_factor):
            ach = pr["work"] / (pr["ms"] * 1e-3) / 1e12
            return {
                "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": None, "kernel": kernel, "peak_basis": basis,
                "executed_mfma_tflops": round(ach * executed_factor, 1),
                "launches_per_step": pr["launches"],
                "avg_launch_us": round(1000.0 * pr["ms"] / max(pr["launches"], 1), 2),
                "algorithmic_gflop_per_step": round(pr["work"] / 1e9, 2),
                "share_of_profiled_step": round(pr["ms"] / cls_step_ms, 4),
                "measured": "one extra step after the timed region with an event pair around every launch of the classes (such a step is ~15 % slower than a timed one)",
            }

        objs = []
        if sb["ms"] > 0:
            what = {"fp32": "split-f16: 3 x v_mfma_f32_32x32x16_f16 per product, fp32-class accuracy",
                    "fp32_bf16x6": "split-bf16: 6 x v_mfma_f32_32x32x16_bf16 per product, fp32-accurate"}.get(precision, f"split-bf16: {nt} x v_mfma_f32_32x32x16_bf16 per product, reduced precision")
            objs.append((sb["ms"], mfma_obj(
                sb, f"pf::igemm_sb_kernel + pf::igemm_sbh_kernel + pf::cnx_mlp_kernel (implicit-GEMM conv/GEMM: linear tiles, 3x3 halo tiles, fused ConvNeXt block MLP; {what})",
                BF16_MFMA_PEAK_TFLOPS, "achieved = algorithmic FLOPs (2*M*N*K) / time, priced against the DENSE 16-bit MFMA peak; the kernel "
                f"executes {nt} MFMA FLOPs per algorithmic FLOP (executed_mfma_tflops), i.e. its ceiling is 2500/{nt} = {2500.0 / nt:.1f} TFLOP/s", float(nt))))
        if ig["ms"] > 0:
            objs.append((ig["ms"], mfma_obj(
                ig, "pf::igemm_kernel (implicit-GEMM conv/GEMM, exact fp32: v_mfma_f32_32x32x2_f32)",
                FP32_MFMA_PEAK_TFLOPS, "dense fp32-input MFMA peak", 1.0)))
        objs.sort(key=lambda t: -t[0])
        if objs:
            # `roofline` = the DOMINANT LAUNCH SHAPE of the dominant kernel class (most time per step): its own algorithmic FLOPs per
            # launch / its own average HIP-event duration; the whole class follows as `split_gemm_class`
            by_shape = {}
            for cls, work, ms, mnk in recs:
                if cls == "igemm_sb":
                    e = by_shape.setdefault(mnk, [0, 0.0, 0.0])
                    e[0] += 1; e[1] += ms; e[2] += work
            dom = max(by_shape.items(), key=lambda kv: kv[1][1]) if by_shape else None
            if dom is not None:
                (M_, N_, K_, KH_), (n_, ms_, work_) = dom
                ach = work_ / (ms_ * 1e-3) / 1e12
                # r05: 3x3 / stride-1 convs with 64-multiple channels on maps >= PF_WINO^2 run as Winograd F(2x2, 3x3) (wino.hip): 16 position products per 2x2 outputs
                # instead of 36 tap products -> 3 * 16 / 36 executed MFMA FLOPs per algorithmic FLOP
                wino_min = int(os.environ.get("PF_WINO", "40"))
                is_wino = bool(KH_ == 3 and precision == "fp32" and wino_min > 0 and N_ % 64 == 0 and (K_ // 9) % 64 == 0 and M_ // max(2 * Bl, 1) >= wino_min * wino_min)
                nt_k = nt * 16.0 / 36.0 if is_wino else float(nt)
                line["roofline"] = {
                    "bound": "mfma", "achieved": round(ach, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic if KH_ == 3 else None,
                    "traffic_unit": "HBM bytes per launch of the 3x3 256->256 @80x80 two-head shape (rocprofv3 PMC: (2*FETCH_SIZE + WRITE_SIZE)*1024); NOT measured in this run: read from the newest committed PMC pass",
                    "traffic_source": traffic_src,
                    "kernel": (f"pf::wino4d_f2x2_kernel (Winograd F(2x2,3x3), tile {os.environ.get('PF_WINO_TILE', 'wino256x64d')}: 16 position GEMMs on the split-f16 MFMA, fused input / output transforms) on the dominant launch shape: conv as GEMM M={M_} N={N_} K={K_} (KH={KH_})" if is_wino else
                               f"pf::igemm_sbh_kernel (3x3 halo-tile implicit GEMM, split scheme of --precision {precision}) on the dominant launch shape: GEMM M={M_} N={N_} K={K_} (KH={KH_})" if KH_ == 3 else
                               f"pf::igemm_sb_kernel / fused block MLP (linear tile, split scheme of --precision {precision}) on the dominant launch shape: GEMM M={M_} N={N_} K={K_} (KH={KH_})"),
                    "peak_basis": f"achieved = algorithmic FLOPs of the launch (2*M*N*K = {2.0 * M_ * N_ * K_ / 1e9:.1f} GFLOP) / its average HIP-event duration, priced against the DENSE 16-bit MFMA peak; "
                                  f"the kernel executes {nt_k:.3g} MFMA FLOPs per algorithmic FLOP (executed_mfma_tflops): its ceiling is 2500/{nt_k:.3g} = {2500.0 / nt_k:.1f} TFLOP/s"
                                  + (" (Winograd: the direct conv's FLOPs stay the numerator -- the figure the r04 halo kernel was priced with -- while 2.25x fewer MFMAs are executed; the kernel is bound by the VALU work of the transforms issued in the same stream, profiles/r05_winograd.md)" if is_wino else ""),
                    "executed_mfma_tflops": round(ach * nt_k, 1), "frac_of_scheme_ceiling": round(ach * nt_k / BF16_MFMA_PEAK_TFLOPS, 4),
                    "mfma_only_ceiling_tflops": 1900.0,
                    "mfma_only_ceiling_note": "NOT measured in this run: scripts/microbench/valu_mfma_interleave.hip on MI355X (profiles/r03_candidates.md): back-to-back "
                                              "v_mfma_f32_32x32x16_f16 retire one per 42 nominal (2.4 GHz) cycles instead of 32 (the part clocks ~1.8 GHz under MFMA load), and a VALU "
                                              "instruction next to them costs 55-75 % of its stand-alone issue time wherever it is placed; `peak` stays the guide's 2500",
                    "launches_per_step": n_ // ev_steps, "avg_launch_us": round(1000.0 * ms_ / n_, 2),
                    "algorithmic_gflop_per_launch": round(work_ / n_ / 1e9, 2),
                    "share_of_step_time": round(ms_ / ev_steps / (1000.0 * dt / args.steps), 4), "event_steps": ev_steps,
                    "measured": f"HIP events on the launch stream around every launch of >= 200 GFLOP in ALL {ev_steps} timed steps (PF_PROFILE_LARGE_ONLY: ~6 event pairs per step)",
                }
            line["split_gemm_class"] = objs[0][1]
            if len(objs) > 1:
                line["roofline_second"] = objs[1][1]
            tot_ms, tot_work = ig["ms"] + sb["ms"], ig["work"] + sb["work"]
            line["implicit_gemm_all"] = {"achieved_tflops_fp32_equiv": round(tot_work / (tot_ms * 1e-3) / 1e12, 2),
                                         "vs_fp32_mfma_peak": round(tot_work / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                         "share_of_profiled_step": round(tot_ms / cls_step_ms, 4)}
        # HBM-bound classes (north_star: ">= 60 % of the HBM roofline on the depthwise stages"): algorithmic bytes / event time
        for cls, key, kernel in (("dwconv3x3_gelu", "roofline_dwconv3x3", "pf::dwconv3x3_gelu kernels"),
                                 ("dwconv7x7", "roofline_dwconv7x7", "pf::dwconv7x7 kernels (+ fused LayerNorm where enabled)"),
                                 ("layernorm", "roofline_layernorm", "pf::layernorm_kernel"),
                                 ("upsample2x", "roofline_upsample2x", "pf::upsample2x_cell_kernel")):
            dw = prof[cls]
            if dw["ms"] > 0:
                gbps = dw["work"] / (dw["ms"] * 1e-3) / 1e9
                line[key] = {
                    "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                    "traffic": None, "kernel": kernel, "launches_per_step": dw["launches"],
                    "algorithmic_mb_per_step": round(dw["work"] / 1e6, 1), "ms_per_step": round(dw["ms"], 3), "measured": "extra profiled step after the timed region",
                }
        d3, d7 = prof["dwconv3x3_gelu"], prof["dwconv7x7"]
        if d3["ms"] + d7["ms"] > 0:  # both depthwise classes together (north_star's "depthwise stages")
            gbps = (d3["work"] + d7["work"]) / ((d3["ms"] + d7["ms"]) * 1e-3) / 1e9
            line["roofline_depthwise_all"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None,
                                              "launches_per_step": d3["launches"] + d7["launches"], "algorithmic_mb_per_step": round((d3["work"] + d7["work"]) / 1e6, 1),
                                              "ms_per_step": round(d3["ms"] + d7["ms"], 3), "measured": "extra profiled step after the timed region",
                                              "note": "in the timed steps the depthwise 7x7 launches belong to the ParamNet branch, which runs beside the next step's backbone"}
        at = prof["attention"]
        if at["ms"] > 0:
            line["attention"] = {"achieved_tflops": round(at["work"] / (at["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(at["ms"], 3),
                                 "launches_per_step": at["launches"], "measured": "extra profiled step after the timed region"}
        line["achieved_tflops_ref_graph"] = round(value / world * GFLOP_PER_IMAGE_REF / 1e3, 2)
        if phases and sum(phases.values()) > 0:
            # SURVEY 8d's component split: elapsed stream time between event marks at the component boundaries of the extra profiled step (joined forward, one stream,
            # an event pair around every launch: ~15 % slower than a timed step; in the timed steps ParamNet runs beside the next step's backbone)
            tot = sum(phases.values())
            line["step_split_ms"] = {**{k: round(v, 3) for k, v in phases.items()}, "total": round(tot, 3),
                                     "share": {k: round(v / tot, 4) for k, v in phases.items()},
                                     "measured": "extra profiled step after the timed region (pf_profile_phases); includes launch gaps and the profile's own event pairs"}
    line["host_resize_ms_per_image"] = round(1000.0 * t_resize, 3)
    if deferred_identical is not None:
        line["deferred_branch_bit_identical"] = deferred_identical  # scalars of the step before the last (branch beside the last step's backbone) == the last step's (branch alone)

    if not dry and not args.no_extras:
        # ---- (1) parity of the timed configuration: one image of the LAST timed step against the CPU oracle
        try:
            pi = min(Bl - 1, 3)
            line["parity"] = parity_check(args.version, resized, sizes[pi], out, index=pi)
            if Bl - 2 > pi:   # ... and one from the END of the batch (another block / wave of every launch; Bl - 2: with four distinct images tiled over the batch it is not
                               # image `pi` again); the GPU suite holds the full-batch evidence
                last = parity_check(args.version, resized, sizes[Bl - 2], out, index=Bl - 2)
                line["parity"]["second_image"] = {k: last[k] for k in ("image_index", "up_1_minus_cos_max", "latitude_l1_deg", "paramnet_max_abs_delta", "ok") if k in last}
                line["parity"]["ok"] = bool(line["parity"]["ok"] and last["ok"])
            line["parity_checked"] = bool(line["parity"]["ok"])
        except Exception as e:  # the checker failing must not hide the measurement
            line["parity"] = {"error": repr(e)}
            line["parity_checked"] = False
        # ---- (2) the same step starting from the original 640x640 uint8 images in HBM (device PIL resize inside the timed region)
        try:
            if mixed:
                raise RuntimeError("not applicable: the mixed stream already starts from the original images")
            orig640 = [torch.from_numpy(imgs[i % len(imgs)]).to(dev) for i in range(B)]
            u8 = torch.empty((B, 320, 320, 3), dtype=torch.uint8, device=dev)

            def step_resize():
                eng.resize_batch_into(orig640, u8)
                pg, pl, params = eng.forward(u8)
                return eng.postprocess_batch(pg, pl, sizes)

            for _ in range(2):
                step_resize()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_resize()
            sync()
            d1 = time.perf_counter() - t1
            same = bool(torch.equal(u8, batch))
            line["with_device_resize"] = {
                "value": round(B * args.steps / d1, 2), "unit": "images/sec", "ms_per_step": round(1000.0 * d1 / args.steps, 3), "n_gpus": 1,
                "workload": f"batch {B} ORIGINAL {S}x{S} uint8 images resident in HBM -> bit-exact PIL resize on the device (pf_resize_batch_u8) -> forward -> post-process",
                "resized_bytes_identical_to_host_pil": same,
            }
        except Exception as e:
            line["with_device_resize"] = {"error": repr(e)}
        # ---- (2b) the window-free exact mode on the same step, driver-visible: fp32_bf16x6 (every operand split exactly into three bf16 values, six MFMAs per product)
        try:
            if precision != "fp32":
                raise RuntimeError("only reported next to the default mode")
            eng.set_precision("fp32_bf16x6")
            for _ in range(2):
                pg, pl, _p = eng.forward(batch)
                eng.postprocess_batch(pg, pl, sizes)
            sync()
            n6 = max(3, min(5, args.steps))
            t1 = time.perf_counter()
            for _ in range(n6):
                pg, pl, _p = eng.forward(batch)
                eng.postprocess_batch(pg, pl, sizes)
            sync()
            d6 = time.perf_counter() - t1
            line["precision_exact"] = {"mode": "fp32_bf16x6", "value": round(Bl * n6 / d6, 2), "unit": "images/sec", "steps": n6, "ms_per_step": round(1000.0 * d6 / n6, 3), "n_gpus": 1,
                                       "note": "same step, exact 3-way bf16 split of every operand (fp32-accurate for any fp32 input, no fp16 range window), forwards joined (no deferred branch)"}
        except Exception as e:
            line["precision_exact"] = {"error": repr(e)}
        finally:
            try:
                eng.set_precision(precision)
            except Exception:
                pass
        line["precision_modes"] = ("fp32 (2-way fp16 split, 3 MFMAs per product) is the fastest mode that holds the parity tolerances; fp32_bf16x6 is the exact alternative; "
                                   "no reduced-precision mode is offered (r01-r04's bf16 switches cost accuracy for +0.6 ... 4 %; a bf16 throughput path was not built: include/pf_hip.h)")
        # ---- (3) latency of the reference's primary call pattern: inference(img) / small batches
        try:
            lat = {}
            for b in (1, 8):
                xb = batch[:b].contiguous()
                sz = sizes[:b]
                for _ in range(3):
                    pg, pl, _p = eng.forward(xb)
                    eng.postprocess_batch(pg, pl, sz)
                sync()
                n = 20
                t1 = time.perf_counter()
                for _ in range(n):
                    pg, pl, _p = eng.forward(xb)
                    eng.postprocess_batch(pg, pl, sz)
                    sync()
                lat[f"b{b}_device_resident"] = round(1000.0 * (time.perf_counter() - t1) / n, 3)
                host_imgs = [imgs[i % len(imgs)] for i in range(b)]
                for _ in range(2):
                    model.inference_batch(host_imgs)
                sync()
                t1 = time.perf_counter()
                for _ in range(8):
                    model.inference_batch(host_imgs)
                    sync()
                lat[f"b{b}_host_images_pil_resize"] = round(1000.0 * (time.perf_counter() - t1) / 8, 3)
            lat["note"] = "ms per call, synchronised after every call; device_resident = pf_forward_u8 + pf_postprocess_batch on 320x320 uint8 in HBM; host_images = inference_batch(list of 640x640 numpy images): host PIL resize + H2D + forward + post-process"
            line["latency_ms"] = lat
        except Exception as e:
            line["latency_ms"] = {"error": repr(e)}
    if not dry and not args.no_extras and world == 1 and not mixed:
        # ---- (4) the host edge, one process: host numpy 640x640 images -> pinned staging -> H2D -> bit-exact device resize -> forward -> post-process -> pinned D2H of the
        #      four field tensors, through PerspectiveFields.inference_stream (three streams, deferred ParamNet branch); what a caller that starts from decoded images gets
        try:
            nb = 12
            host_batches = [[imgs[(j + i) % len(imgs)] for i in range(B)] for j in range(nb)]
            model.device_resize = True
            try:
                list(model.inference_stream(host_batches[:2], to_host=True, depth=2))  # warm-up (allocator, pinned buffers)
                sync()
                t1 = time.perf_counter()
                nimg = 0
                for res in model.inference_stream(host_batches, to_host=True, depth=2):
                    nimg += len(res)
                sync()
                d_e2e = time.perf_counter() - t1
                # the same without the D2H of the fields (results stay on the device)
                t1 = time.perf_counter()
                for res in model.inference_stream(host_batches, to_host=False, depth=2):
                    pass
                sync()
                d_dev = time.perf_counter() - t1
            finally:
                model.device_resize = False
            mb_in, mb_out = B * S * S * 3 / 1e6, B * 3 * S * S * 4 / 1e6 + B * 3 * 320 * 320 * 4 / 1e6
            line["e2e_host_stream"] = {
                "value": round(nimg / d_e2e, 2), "unit": "images/sec", "n_gpus": 1, "batches": nb, "batch": B, "ms_per_batch": round(1000.0 * d_e2e / nb, 3),
                "fields_left_on_device": {"value": round(nb * B / d_dev, 2), "ms_per_batch": round(1000.0 * d_dev / nb, 3)},
                "pcie_mb_per_batch": {"h2d": round(mb_in, 1), "d2h": round(mb_out, 1)},
                "workload": f"{nb} batches of {B} host numpy {S}x{S} uint8 images, ONE process / one Python thread: np.copyto into pinned staging, one H2D per batch, device resize, "
                            "forward, post-process, async D2H of pred_gravity(_original) / pred_latitude(_original) into pinned host tensors (PerspectiveFields.inference_stream, depth 2)",
                "note": "`value` of the headline excludes this edge by the bench contract (inputs resident in HBM); this is the PCIe- and host-inclusive rate",
            }
        except Exception as e:
            line["e2e_host_stream"] = {"error": repr(e)}
        # ---- (5) BASELINE configs[1]: batch 8, 640x640, PersNet-360Cities (73 / 180-way logits + argmax decode), 320x320 uint8 inputs resident in HBM
        try:
            m1 = PerspectiveFields("PersNet-360Cities", weights="synthetic:0", precision=precision).eval().to(dev)
            e1 = m1._get_engine()
            x1 = torch.from_numpy(np.stack([m1.aug.apply_image(imgs[i % len(imgs)]) for i in range(8)])).to(dev)
            sz1 = [(S, S)] * 8
            for _ in range(3):
                pg1, pl1, _p = e1.forward(x1)
                e1.postprocess_batch(pg1, pl1, sz1)
            sync()
            n1 = 20
            t1 = time.perf_counter()
            for _ in range(n1):
                pg1, pl1, _p = e1.forward(x1)
                e1.postprocess_batch(pg1, pl1, sz1)
            sync()
            d1c = time.perf_counter() - t1
            line["configs_1"] = {"value": round(8 * n1 / d1c, 2), "unit": "images/sec", "n_gpus": 1, "steps": n1, "ms_per_step": round(1000.0 * d1c / n1, 3),
                                 "workload": f"batch 8, {S}x{S}, PersNet-360Cities: backbone + decoders + 73 / 180-way classification heads (NCHW logits) + argmax decode + post-process",
                                 "dtype": line.get("dtype"), "note": "BASELINE names this config 'bf16': no reduced-precision mode is offered (include/pf_hip.h pf_set_precision); "
                                 "this is the fp32-class parity mode"}
            del m1, e1
        except Exception as e:
            line["configs_1"] = {"error": repr(e)}
    if not dry and not args.no_extras and world == 1 and not mixed:
        # ---- (6) the DEFAULT-constructed model: PerspectiveFields(version) has precision="auto", which reads the saturation counter behind every forward (a 4-byte copy
        #      and a host synchronisation) and builds the reference's result dicts -- the headline drives the engine directly with the fast mode pinned.  Same device-resident
        #      batch through the engine path of inference_batch (PerspectiveFields._run: forward, counter read, post-process, dicts), forwards joined.
        try:
            m_auto = PerspectiveFields(args.version, weights="synthetic:0").eval().to(dev)
            assert m_auto.precision == "auto"
            for _ in range(3):
                m_auto._run(batch, sizes)   # the first call is the range probe that settles the mode
            sync()
            na = max(5, min(10, args.steps))
            t1 = time.perf_counter()
            for _ in range(na):
                res_a = m_auto._run(batch, sizes)
            sync()
            d_a = time.perf_counter() - t1
            line["default_auto"] = {"value": round(B * na / d_a, 2), "unit": "images/sec", "steps": na, "ms_per_step": round(1000.0 * d_a / na, 3), "n_gpus": 1,
                                    "precision_settled_on": m_auto.precision, "result_keys": len(res_a[0]),
                                    "workload": "the headline batch through the default-constructed model (precision='auto'): PerspectiveFields._run = forward + read of the saturation "
                                                "counter (host sync per batch) + post-process + the reference's result dicts; joined forwards (no deferred ParamNet branch)"}
            del m_auto
        except Exception as e:
            line["default_auto"] = {"error": repr(e)}
        # ---- (7) BASELINE configs[4]: the mixed-resolution stream (384x512 : 640x640 : 1024x1365 = 1:2:1), 64 images per step, ORIGINAL images resident in HBM ->
        #      bucketed bit-exact device resize -> forward at 320x320 -> post-process to each original size, through the product's sharded step (world 1 here)
        try:
            Bm = 64
            sizes_m = [MIXED_PATTERN[i % len(MIXED_PATTERN)] for i in range(Bm)]
            img_m = {hw: torch.from_numpy(synthetic_image(hw[0], hw[1], seed=7 + k)).to(dev) for k, hw in enumerate(sorted(set(MIXED_PATTERN)))}
            orig_m = [img_m[hw] for hw in sizes_m]
            batch_m = torch.empty((Bm, 320, 320, 3), dtype=torch.uint8, device=dev)
            spf.set_pipeline(True)
            for _ in range(2):
                spf.forward_step(batch_m, sizes_m, originals=orig_m)
            spf.drain(); sync()
            nm = 6
            t1 = time.perf_counter()
            for _ in range(nm):
                spf.forward_step(batch_m, sizes_m, originals=orig_m)
            spf.drain(); sync()
            d_m = time.perf_counter() - t1
            spf.set_pipeline(False)
            line["configs_4"] = {"value": round(Bm * nm / d_m, 2), "unit": "images/sec", "n_gpus": 1, "steps": nm, "ms_per_step": round(1000.0 * d_m / nm, 3), "batch": Bm,
                                 "workload": "BASELINE configs[4] on one GPU: 64 images per step, 384x512 : 640x640 : 1024x1365 = 1:2:1 (one synthetic image per bucket), original uint8 "
                                             "images resident in HBM -> bucketed bit-exact device resize -> forward -> post-process to each original size (ShardedPerspectiveFields.forward_step); "
                                             "per-bucket figures: `python bench.py --workload mixed --batch 64`"}
            del orig_m, batch_m, img_m
        except Exception as e:
            line["configs_4"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline and not dry:
        try:
            line["cpu_baseline"] = cpu_baseline(args.version, S)
        except Exception as e:  # the checker failing must not hide the measurement
            line["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
