"""Image-level data parallelism: one process per GPU, images are independent units
(no cross-image op on the path; BatchNorm is eval-mode), weights replicated.  The only
exchange step is an all-gather of the per-image ParamNet scalars ((B_local, 8) fp32,
<= 8 KB at B=256) over RCCL/xGMI (`torch.distributed` backend "nccl"), or gloo on CPU in
tests.  Dense fields stay on the GPU that produced them, as in the reference (outputs
stay on device, perspectivefields.py:255-272)."""
from __future__ import annotations

from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_round_robin_by_bucket(sizes: Sequence[Tuple[int, int]], rank: int, world: int) -> List[int]:
    """Mixed-resolution streams: round-robin within each (H, W) bucket so every rank gets the same
    mix of post-process work (the network batch itself is resolution independent: all -> 320x320)."""
    buckets = {}
    for i, hw in enumerate(sizes):
        buckets.setdefault(tuple(hw), []).append(i)
    mine, k = [], 0
    for hw in sorted(buckets):
        for i in buckets[hw]:
            if k % world == rank:
                mine.append(i)
            k += 1
    return sorted(mine)


def gather_params(local: torch.Tensor, counts: Sequence[int] = None, group=None) -> torch.Tensor:
    """All-gather (B_local, P) -> (sum B_local, P) in rank order.  `counts` = per-rank row counts
    (needed only when shards are ragged); single-process runs return `local` unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if counts is None:
        counts = [local.shape[0]] * world
    mx = max(counts)
    padded = local
    if local.shape[0] < mx:
        padded = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    chunks = [out[r * mx : r * mx + counts[r]] for r in range(world)]
    return torch.cat(chunks)


class StepOutput(NamedTuple):
    """One device-resident step of ShardedPerspectiveFields.forward_step."""
    pred_gravity: torch.Tensor            # (B_local, Cg, 320, 320)
    pred_latitude: torch.Tensor           # (B_local, Cl, 320, 320)
    fields: list                          # [(up (2, H, W), latitude (H, W))] per local image
    params: Optional[torch.Tensor]        # this rank's raw (B_local, 8) rows of THIS step (with the pipeline on: complete once the next step has been issued / after drain())
    gathered: Optional[torch.Tensor]      # (B_global, 8) in rank order: this step's rows, or -- pipeline on -- the PREVIOUS step's (None on the first step)


class ShardedOutput(NamedTuple):
    """Result of ShardedPerspectiveFields.inference_batch / one item of .inference_stream."""
    indices: List[int]                    # positions of this rank's images in the global list
    results: List[dict]                   # the reference's per-image dicts for those images (dense fields stay on this rank's GPU, as in the reference)
    params: Optional[torch.Tensor]        # (B_global, 8) raw engine rows of EVERY image, in the global list's order, on this rank's device (None without a ParamNet)


class ShardedPerspectiveFields:
    """`PerspectiveFields.inference_batch` over the GPUs of one node, SPMD (reference call being scaled: perspective2d/perspectivefields.py:207-221).

    One process per GPU, `torch.distributed` initialised by the launcher (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests); every rank constructs the same
    model (weights replicated: 0.4 GB) and makes the same calls with the same GLOBAL image list.  Images are independent units: a rank runs the network on its
    shard only -- contiguous, or round-robin within each (H, W) bucket for mixed-resolution streams so that every rank gets the same mix of resize / post-process
    work -- and the one exchange step is an all-gather of the raw (B_local, 8) camera-parameter rows (<= 8 KB at B = 256).  Dense fields stay where they were made.

    Three call shapes:
      * inference_batch(global_list)            -> ShardedOutput (this rank's dicts + all images' parameter rows in global order);
      * inference_stream(iterable of lists)     -> ShardedOutput per global batch, through PerspectiveFields.inference_stream (upload / compute / download overlapped,
                                                   bit-exact device resize: a host core resizes ~330 images / s, eight GPUs want > 10 000) -- the gather of batch i is
                                                   issued when batch i is finished, on a stream that does not wait for batch i + 1's compute;
      * forward_step(...) / drain()             -> the device-resident step of bench.py (inputs already in HBM): forward + post-process (+ device resize) + gather.
        With set_pipeline(True) the ParamNet branch of step i runs beside step i + 1's backbone (Engine.set_defer_params), so the rows of step i are complete in stream
        order only once step i + 1 has been issued: they are gathered ONE STEP LATE, and drain() joins the last branch and gathers the last step's rows.
    """

    def __init__(self, model, group=None, engine=None):
        self.model = model
        self.group = group
        self._engine = engine   # injected by tests (a stub on CPU); the product path uses the model's engine
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.pipeline = False
        self._late = None       # (params, counts) of the step whose rows are not gathered yet

    @property
    def engine(self):
        return self._engine if self._engine is not None else self.model._get_engine()

    # ------------------------------------------------------------------ partition
    def partition(self, sizes_global: Sequence, bucketed: bool = False) -> List[List[int]]:
        """Index lists of all ranks for a global list of (H, W) sizes (or of anything else when not bucketed: only its length is used)."""
        n = len(sizes_global)
        if bucketed:
            return [shard_round_robin_by_bucket([tuple(int(v) for v in s) for s in sizes_global], r, self.world) for r in range(self.world)]
        return [list(range(*shard_range(n, r, self.world))) for r in range(self.world)]

    def _to_global_order(self, gathered: torch.Tensor, parts: List[List[int]]) -> torch.Tensor:
        """rank-major rows (what the all-gather returns) -> the order of the global list"""
        order = [i for part in parts for i in part]
        if order == sorted(order):
            return gathered
        inv = torch.empty(len(order), dtype=torch.long)
        inv[torch.tensor(order, dtype=torch.long)] = torch.arange(len(order), dtype=torch.long)
        return gathered.index_select(0, inv.to(gathered.device))

    # ------------------------------------------------------------------ host images
    def inference_batch(self, img_bgr_list: Sequence, bucketed: bool = False, sizes: Sequence = None) -> ShardedOutput:
        """Every rank passes the same global list (entries of other ranks' shards may be None when `sizes` gives all (H, W) -- needed only for bucketed sharding)."""
        if sizes is None:
            if bucketed and any(im is None for im in img_bgr_list):
                raise ValueError("bucketed sharding needs the (H, W) of every image: pass sizes= when other ranks' images are None")
            sizes = [tuple(int(v) for v in im.shape[:2]) if im is not None else None for im in img_bgr_list]
        parts = self.partition(sizes, bucketed)
        mine = parts[self.rank]
        results, params = self.model.inference_batch_with_params([img_bgr_list[i] for i in mine])
        allp = None
        if params is not None:
            allp = self._to_global_order(gather_params(params, [len(p) for p in parts], self.group), parts)
        return ShardedOutput(mine, results, allp)

    def inference_stream(self, batches, bucketed: bool = False, to_host: bool = True, depth: int = 2, device_resize: bool = True):
        """`batches`: an iterable of GLOBAL image lists, the same on every rank.  Every rank must own at least one image of every batch (len(batch) >= world size)."""
        parts_q: list = []

        def local():
            for imgs in batches:
                sz = [tuple(int(v) for v in im.shape[:2]) for im in imgs]
                parts = self.partition(sz, bucketed)
                if not parts[self.rank]:
                    raise ValueError(f"a batch of {len(imgs)} images leaves rank {self.rank} of {self.world} without work: stream batches of at least `world` images")
                parts_q.append(parts)
                yield [imgs[i] for i in parts[self.rank]]

        prev = self.model.device_resize
        self.model.device_resize = bool(device_resize) and self.model.device.type == "cuda"
        try:
            for results, params in self.model.inference_stream(local(), to_host=to_host, depth=depth, with_params=True):
                parts = parts_q.pop(0)
                allp = None
                if params is not None:   # the batch is finished: its rows are complete; the collective does not wait for the batches still in flight
                    allp = self._to_global_order(gather_params(params, [len(p) for p in parts], self.group), parts)
                yield ShardedOutput(parts[self.rank], results, allp)
        finally:
            self.model.device_resize = prev

    # ------------------------------------------------------------------ device-resident steps (bench.py)
    def set_pipeline(self, on: bool):
        """Deferred ParamNet branch for loops that issue step after step (Engine.set_defer_params); off: joins a pending branch.  Call drain() before switching off."""
        self.engine.set_defer_params(bool(on))
        self.pipeline = bool(on)

    def forward_step(self, batch_u8, sizes: Sequence, counts: Sequence[int] = None, originals: Sequence = None) -> StepOutput:
        """batch_u8: this rank's (B_local, 320, 320, 3) network input in HBM (filled from `originals` -- uint8 (H, W, 3) device tensors -- by the bit-exact device resize
        when given); sizes: the local images' (H, W); counts: per-rank row counts when the shards are ragged."""
        eng = self.engine
        if originals is not None:
            eng.resize_batch_into(originals, batch_u8)
        pg, pl, params = eng.forward(batch_u8)
        fields = eng.postprocess_batch(pg, pl, sizes)
        gathered = None
        if params is not None:
            if self.pipeline:
                prev, self._late = self._late, (params, counts)
                if prev is not None:   # complete in stream order: this step's forward has joined the previous step's branch
                    gathered = gather_params(prev[0], prev[1], self.group)
            else:
                gathered = gather_params(params, counts, self.group)
        return StepOutput(pg, pl, fields, params, gathered)

    def drain(self) -> Optional[torch.Tensor]:
        """The tail of the pipeline: joins the last step's ParamNet branch and gathers its rows (None when nothing is pending)."""
        if not self.pipeline:
            return None
        self.engine.join_params()
        if self._late is None:
            return None
        (params, counts), self._late = self._late, None
        return gather_params(params, counts, self.group)
