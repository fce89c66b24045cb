"""Multi-GPU sharding of the render path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code at all; its "scale-out" is SLURM array jobs.
The unit of work here -- a pose (one DRR), or a whole volume -- is independent, so the path shards
with NO collective inside the render:

* one volume, many poses (configs C2/C3): the 512 MiB volume is replicated on every GPU
  (0.2 % of 288 GB) and the pose batch is split; the only exchange is an all-gather of the rendered
  DRRs (30 MB per rank at B=116, 256^2) -- ``all_gather_drrs``;
* the same volume optimised on several GPUs: the voxel gradients are summed -- ``allreduce_volume_grad``;
* multi-start registration (C4): one independent optimisation per rank, then ``multistart_best``
  all-gathers (score, 4x4 pose) = 68 B per rank and every rank takes the arg-max;
* one volume per GPU (C5): nothing render-side to exchange (the regressor's gradients are DDP's job).

Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over gloo with world_size 2.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .pose import RigidTransform


def init_distributed(backend: str | None = None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if use_gpu else "gloo")
        kwargs = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n: int, rank: int | None = None, world: int | None = None):
    """Contiguous, balanced split of range(n): the first n % world ranks get one extra item."""
    rank = _rank() if rank is None else rank
    world = _world() if world is None else world
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_counts(n: int, world: int | None = None):
    world = _world() if world is None else world
    return [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]


def shard_poses(pose: RigidTransform, rank: int | None = None, world: int | None = None) -> RigidTransform:
    lo, hi = shard_bounds(len(pose), rank, world)
    return RigidTransform(pose.matrix[lo:hi])


def all_gather_drrs(img: torch.Tensor, total: int | None = None, async_op: bool = False):
    """All-gather per-rank DRR batches [B_r, C, H, W] into [sum B_r, C, H, W] on every rank.

    ``total`` (the global batch size) lets ranks hold ragged shards (``shard_bounds`` split): shards
    are padded to the largest one for a single fixed-size collective and trimmed afterwards.
    Returns the gathered tensor, or (tensor_getter, work) when ``async_op``.
    """
    world = _world()
    if world == 1:
        return img if not async_op else ((lambda: img), None)
    counts = shard_counts(total, world) if total is not None else [img.shape[0]] * world
    bmax = max(counts)
    send = img.detach()
    if send.shape[0] < bmax:
        pad = torch.zeros(bmax - send.shape[0], *send.shape[1:], dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad])
    send = send.contiguous()
    out = torch.empty(world * bmax, *send.shape[1:], dtype=send.dtype, device=send.device)
    work = dist.all_gather_into_tensor(out, send, async_op=async_op)

    def finish():
        if all(c == bmax for c in counts):
            return out
        return torch.cat([out[r * bmax: r * bmax + c] for r, c in enumerate(counts)])

    if async_op:
        return finish, work
    return finish()


def render_sharded(render_fn, pose: RigidTransform) -> torch.Tensor:
    """Render this rank's contiguous slice of ``pose`` with ``render_fn(pose_shard) -> [B_r,C,H,W]``
    and return the full batch on every rank (forward only; no collective inside the render)."""
    local = render_fn(shard_poses(pose))
    return all_gather_drrs(local, total=len(pose))


def allreduce_volume_grad(grad: torch.Tensor) -> torch.Tensor:
    """Sum the voxel gradients of ranks that rendered different poses of the SAME volume (in place)."""
    if _world() > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return grad


def allreduce_volume_grad_bucketed(grad: torch.Tensor, n_buckets: int = 8, force: bool = False):
    """The same sum as ``allreduce_volume_grad`` issued as ``n_buckets`` ASYNC all-reduces over contiguous slabs of the volume's
    first axis (a [512, 512, 512] float32 gradient is 512 MiB: eight 64 MiB collectives are in flight together, which is what
    lets RCCL use all seven xGMI links of a GPU instead of one ring -- SURVEY.md section 8e).  Returns the list of work handles
    (empty on one rank unless ``force``: the one-rank groups of bench.py's dry run); ``wait_all`` blocks on them.  The slabs are
    views of ``grad``: the sum lands in place."""
    if (_world() == 1 and not force) or not (dist.is_available() and dist.is_initialized()):
        return []
    flat = grad.view(grad.shape[0], -1) if grad.dim() > 1 and grad.is_contiguous() else grad.reshape(-1, 1)
    n = flat.shape[0]
    n_buckets = max(1, min(int(n_buckets), n))
    works = []
    for k in range(n_buckets):
        lo, hi = shard_bounds(n, k, n_buckets)
        if hi > lo:
            works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
    return works


class SlabAllReduce:
    """The voxel-gradient sum of a fwd+bwd(pose+voxel) step, OVERLAPPED with the backward that produces it: while installed, the
    renderer computes the voxel gradient in ``count`` x slabs (xvr_amd.renderers.VOXEL_GRAD_SLABS, option gather_slab of the library)
    and every finished slab goes into an async all-reduce at once -- RCCL moves slab i over xGMI while the splat works on slab i + 1;
    only the last slab's collective is exposed.

    Contract (checked): ONE render with a voxel gradient per backward, and the volume's ``.grad`` is ``None`` when the
    backward starts.  Autograd would otherwise ADD the backward's tensor into the existing ``.grad`` (or a second render's tensor
    into the first's) on the compute stream while the collectives are still writing it, and the sum would hold whatever the race left.
    ``install(leaf)`` raises on a ``.grad`` that is not ``None``; slabs of a second gradient tensor make ``finish`` raise (after
    waiting for what is in flight).  ``finish(grad)`` waits and makes sure the sums are what the caller reads: the slab views
    (and the pending collectives) hold references to the backward's tensor, so autograd never adopts it as ``.grad`` but clones it
    -- mid-reduction; the reduced tensor is therefore put in the clone's place (``leaf.grad`` is re-pointed when the leaf was given
    to ``install``: no copy back; otherwise the reduced slabs are copied over ``grad``)."""

    def __init__(self, count: int = 4, force: bool = False):
        self.count, self.force = int(count), force
        self.slabs, self.works = [], []
        self.leaf, self._base, self._second = None, None, False

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (_world() > 1 or self.force)

    def _hook(self, i, slab):
        base = slab._base if slab._base is not None else slab
        if self._base is None:
            self._base = base
        elif base.data_ptr() != self._base.data_ptr():
            self._second = True          # a second render's gradient in the same backward: not reduced, reported by finish()
            return
        self.slabs.append(slab)
        if self._active():
            self.works.append(dist.all_reduce(slab, op=dist.ReduceOp.SUM, async_op=True))

    def install(self, leaf=None):
        """``leaf``: the volume tensor whose ``.grad`` the backward will fill (optional; lets ``finish`` adopt the reduced tensor
        instead of copying it back, and lets the ``.grad is None`` half of the contract be checked here)."""
        from . import renderers
        if leaf is not None and leaf.grad is not None:
            raise RuntimeError("SlabAllReduce: the volume's .grad must be None when the backward starts (autograd would accumulate into it "
                               "while the slab all-reduces are in flight); set it to None, or reduce after the backward with "
                               "allreduce_volume_grad_bucketed")
        self.leaf = leaf
        renderers.VOXEL_GRAD_SLABS = (self.count, self._hook)
        return self

    def remove(self):
        from . import renderers
        renderers.VOXEL_GRAD_SLABS = None

    __enter__ = install

    def __exit__(self, *exc):
        self.remove()

    def fired(self) -> bool:
        return bool(self.slabs)

    def finish(self, grad) -> None:
        wait_all(self.works)
        slabs, self.slabs, self.works = self.slabs, [], []
        base, second, leaf = self._base, self._second, self.leaf
        self._base, self._second, self.leaf = None, False, None
        if second:
            raise RuntimeError("SlabAllReduce: two renders with a voxel gradient in one backward -- only the first one's slabs were "
                               "reduced and autograd added the second into a tensor the collectives were writing; render once per "
                               "backward, or reduce after it with allreduce_volume_grad_bucketed")
        if grad is None or not slabs:
            return
        if base.data_ptr() != grad.data_ptr():
            if leaf is not None and leaf.grad is grad and base.shape == grad.shape and base.is_contiguous():
                leaf.grad = base         # autograd cloned the backward's tensor mid-reduction: the reduced tensor takes its place
                return
            row = base.stride(0)
            for sl in slabs:
                x0 = (sl.storage_offset() - base.storage_offset()) // row
                grad[x0:x0 + sl.shape[0]].copy_(sl)


def wait_all(works) -> None:
    for w in works:
        if w is not None:
            w.wait()


def multistart_best(score: torch.Tensor, pose_matrix: torch.Tensor):
    """Every rank ran its own registration; return (best_score, best_pose[4,4], best_rank) everywhere.
    ``score`` is a scalar tensor (higher is better), ``pose_matrix`` is [4,4] or [1,4,4]."""
    world = _world()
    rec = torch.cat([score.detach().reshape(1).to(torch.float32), pose_matrix.detach().reshape(16).to(torch.float32)])
    if world == 1:
        return rec[0], rec[1:].reshape(4, 4), 0
    flat = torch.empty(world * 17, dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(flat, rec.contiguous())
    out = flat.view(world, 17)
    best = int(torch.argmax(out[:, 0]).item())
    return out[best, 0], out[best, 1:].reshape(4, 4), best
