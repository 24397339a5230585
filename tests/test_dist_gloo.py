"""world_size-2 gloo tests (CPU) of the N>1 host path: image sharding + all-gather of the ParamNet scalars."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from perspectivefields_amd.dist import gather_params, shard_range, shard_round_robin_by_bucket


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_items, rank, world)
        # stand-in for the engine output: row i of the global (n_items, 8) matrix is i * [1..8]
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.arange(1, 9, dtype=torch.float32)[None]
        counts = [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]
        out = gather_params(local, counts if ragged else None)
        want = torch.arange(n_items, dtype=torch.float32)[:, None] * torch.arange(1, 9, dtype=torch.float32)[None]
        q.put((rank, bool(torch.equal(out, want)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items,ragged", [(8, False), (7, True)])
def test_gather_params_world2(n_items, ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_items, 8) for _, _, shape in res)


def test_single_process_gather_is_identity():
    x = torch.randn(5, 8)
    assert gather_params(x) is x
