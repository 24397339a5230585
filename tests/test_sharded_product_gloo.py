"""The PRODUCT's multi-GPU entry point (perspectivefields_amd.dist.ShardedPerspectiveFields) under world-size-2 / 3 gloo on CPU.

The real `PerspectiveFields` object runs its real host path (PIL resize, chunking, dict assembly, ParamNet entries) on CPU tensors; only the HIP engine is replaced
by a stub whose camera-parameter rows encode WHICH image they were computed from (its mean pixel value), so that sharding, ragged shards, the bucket round-robin of
mixed-resolution lists, the all-gather and the restoring of the global order can be checked exactly.  No numbers are produced here -- the GPU suite owns parity.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StubEngine:
    """Stands in for perspectivefields_amd.engine.Engine on CPU: same methods, rows that identify their input image."""

    max_batch = 4          # smaller than the shards below: the model's chunking runs too
    defer_params = False

    def __init__(self):
        self.forwards = 0
        self.joined = 0

    def forward(self, batch):
        B = batch.shape[0]
        self.forwards += 1
        x = batch.reshape(B, -1).to(torch.float32)
        params = torch.zeros((B, 8), dtype=torch.float32)
        params[:, 0] = x.mean(1)           # the image's identity (images below are constant-valued)
        params[:, 2] = 0.5                 # a vfov that keeps 1 / tan finite in the centered scalar formulas
        return torch.zeros((B, 2, 320, 320)), torch.zeros((B, 1, 320, 320)), params

    def postprocess_batch(self, pg, pl, sizes):
        return [(torch.zeros((2, h, w)), torch.zeros((h, w))) for h, w in sizes]

    def set_defer_params(self, on):
        self.defer_params = bool(on)

    def join_params(self):
        self.joined += 1

    def resize_batch_into(self, imgs, out):
        for i, im in enumerate(imgs):
            out[i] = im.reshape(-1)[0]
        return out


def _model():
    from perspectivefields_amd import PerspectiveFields

    m = PerspectiveFields("Paramnet-360Cities-edina-centered", weights="synthetic:0", precision="fp32")
    eng = StubEngine()
    m._get_engine = lambda: eng   # the one seam: everything above it is the product's own code
    return m, eng


def _images(sizes):
    return [np.full((h, w, 3), 10 + i, dtype=np.uint8) for i, (h, w) in enumerate(sizes)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from perspectivefields_amd.dist import ShardedPerspectiveFields, shard_range, shard_round_robin_by_bucket

        m, eng = _model()
        spf = ShardedPerspectiveFields(m)
        ok = {}
        # ---- (1) contiguous ragged shards: 11 images (> max_batch per shard: chunked) over `world` ranks
        sizes = [(48, 64)] * 11
        imgs = _images(sizes)
        out = spf.inference_batch(imgs)
        lo, hi = shard_range(11, rank, world)
        ok["contig_indices"] = out.indices == list(range(lo, hi))
        ok["contig_rows_global_order"] = bool(torch.equal(out.params[:, 0], torch.arange(10, 21, dtype=torch.float32)))
        ok["contig_dicts"] = len(out.results) == hi - lo and all(float(r["pred_roll"]) == 10.0 + i for r, i in zip(out.results, out.indices))
        ok["contig_keys"] = list(out.results[0]) == ["pred_gravity", "pred_gravity_original", "pred_latitude", "pred_latitude_original", "pred_latitude_original_mode",
                                                      "pred_roll", "pred_pitch", "pred_vfov", "pred_rel_focal", "pred_general_vfov", "pred_rel_cx", "pred_rel_cy"]
        ok["contig_field_shape"] = tuple(out.results[0]["pred_gravity_original"].shape) == (2, 48, 64)
        # ---- (2) mixed-resolution list, bucket round-robin, other ranks' images withheld (None) with sizes= given: rows come back in the GLOBAL order
        pattern = [(24, 32), (40, 40), (40, 40), (64, 86)]
        sizes = [pattern[i % 4] for i in range(10)]
        imgs = _images(sizes)
        mine = shard_round_robin_by_bucket(sizes, rank, world)
        sparse = [im if i in mine else None for i, im in enumerate(imgs)]
        out = spf.inference_batch(sparse, bucketed=True, sizes=sizes)
        ok["bucket_indices"] = out.indices == mine
        ok["bucket_rows_global_order"] = bool(torch.equal(out.params[:, 0], torch.arange(10, 20, dtype=torch.float32)))
        ok["bucket_sizes"] = all(tuple(r["pred_latitude_original"].shape) == sizes[i] for r, i in zip(out.results, out.indices))
        per_bucket = {hw: sum(1 for i in mine if sizes[i] == hw) for hw in set(sizes)}
        ok["bucket_balance"] = all(abs(per_bucket[hw] - sum(1 for s in sizes if s == hw) / world) < 1.0 for hw in per_bucket)
        # ---- (3) fewer images than ranks: the empty shard contributes zero rows, every rank still gets every row
        few = _images([(16, 16)] * (world - 1))
        out = spf.inference_batch(few)
        ok["empty_shard"] = tuple(out.params.shape) == (world - 1, 8) and bool(torch.equal(out.params[:, 0], torch.arange(10, 10 + world - 1, dtype=torch.float32)))
        ok["empty_shard_results"] = len(out.results) == (1 if rank < world - 1 else 0)
        # ---- (4) the device-resident step of bench.py: pipeline on -> rows one step late, drain() joins and gathers the last step's
        spf2 = ShardedPerspectiveFields(None, engine=StubEngine())
        spf2.set_pipeline(True)
        counts = [3 + (r == 0) for r in range(world)]   # ragged: rank 0 has one image more
        n = counts[rank]
        got = []
        for step in range(3):
            batch = torch.full((n, 320, 320, 3), 50 * step + rank, dtype=torch.uint8)
            o = spf2.forward_step(batch, [(8, 8)] * n, counts)
            got.append(o.gathered)
        last = spf2.drain()
        want = lambda step: torch.cat([torch.full((counts[r],), 50.0 * step + r) for r in range(world)])
        ok["step_first_is_none"] = got[0] is None
        ok["step_late_rows"] = bool(torch.equal(got[1][:, 0], want(0))) and bool(torch.equal(got[2][:, 0], want(1)))
        ok["step_drain_rows"] = bool(torch.equal(last[:, 0], want(2))) and spf2.engine.joined == 1 and spf2.drain() is None
        spf2.set_pipeline(False)
        o = spf2.forward_step(torch.full((n, 320, 320, 3), 7, dtype=torch.uint8), [(8, 8)] * n, counts)
        ok["step_joined_rows"] = bool(torch.equal(o.gathered[:, 0], torch.full((sum(counts),), 7.0)))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_product_class_under_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        bad = [k for k, v in ok.items() if not v]
        assert not bad, f"rank {rank}: {bad}"


def test_sharded_single_process_is_plain_inference_batch():
    """Without a process group the class is inference_batch + the raw rows."""
    from perspectivefields_amd.dist import ShardedPerspectiveFields

    m, eng = _model()
    spf = ShardedPerspectiveFields(m)
    assert (spf.rank, spf.world) == (0, 1)
    out = spf.inference_batch(_images([(20, 30)] * 5))
    assert out.indices == [0, 1, 2, 3, 4] and tuple(out.params.shape) == (5, 8) and len(out.results) == 5
    assert eng.forwards == 2   # 5 images, 4 per engine forward
    with pytest.raises(ValueError):
        spf.inference_batch([None, None], bucketed=True)
