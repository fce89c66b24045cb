"""world_size-2 gloo tests of the pose-sharded render path (CPU; the render itself is stubbed with the
oracle because the HIP renderer needs a GPU -- what is under test is the sharding + collectives)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out_q):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle.diffdrr_restated import RenderSpec, drr_from_pose
    from xvr_amd import distributed as xd
    from xvr_amd.pose import convert

    r, w, dev = xd.init_distributed("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    g = torch.Generator().manual_seed(0)
    vol = torch.rand(12, 12, 12, generator=g)
    affine = torch.eye(4)
    affine[:3, 3] = -5.5
    rot = (torch.rand(B, 3, generator=g) - 0.5) * 0.6
    xyz = torch.tensor([[0.0, 120.0, 0.0]]).repeat(B, 1) + (torch.rand(B, 3, generator=g) - 0.5) * 10
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    spec = RenderSpec(renderer="trilinear", n_points=40)

    def render_fn(p):
        return drr_from_pose(vol, affine, p.matrix, 6, 5, 200.0, 3.0, 3.0, 0.0, 0.0, spec)

    full = render_fn(pose)                      # what one process would render
    gathered = xd.render_sharded(render_fn, pose)  # each rank renders its slice, then all-gather
    lo, hi = xd.shard_bounds(B)
    getter, work = xd.all_gather_drrs(render_fn(xd.shard_poses(pose)), total=B, async_op=True)
    work.wait()
    ok_async = torch.allclose(getter(), full, atol=1e-6)

    grad = torch.full((4, 4, 4), float(rank + 1))
    xd.allreduce_volume_grad(grad)
    # the bucketed, asynchronous form bench.py's N > 1 step uses: slabs of the first axis, summed in place
    g2 = (torch.arange(5 * 3 * 2, dtype=torch.float32).reshape(5, 3, 2) + 100.0 * rank)
    works = xd.allreduce_volume_grad_bucketed(g2, n_buckets=3)
    xd.wait_all(works)
    bucket_ok = len(works) == 3 and torch.equal(g2, 2 * torch.arange(30, dtype=torch.float32).reshape(5, 3, 2) + 100.0)
    # the overlapped form: the renderer hands every finished x slab of the gradient to the hook (here: by hand), the sums land in
    # place -- and in a COPY of the gradient too, which is what .grad is if autograd did not adopt the backward's tensor
    g3 = (torch.arange(5 * 3 * 2, dtype=torch.float32).reshape(5, 3, 2) + 100.0 * rank)
    sl = xd.SlabAllReduce(count=3)
    for i, (a, b) in enumerate(((0, 2), (2, 3), (3, 5))):
        sl._hook(i, g3[a:b])
    fired, copy = sl.fired(), torch.zeros_like(g3)
    sl.finish(copy)
    want3 = 2 * torch.arange(30, dtype=torch.float32).reshape(5, 3, 2) + 100.0
    slab_ok = fired and not sl.fired() and torch.equal(g3, want3) and torch.equal(copy, want3)
    # the contract: a leaf whose .grad is a clone gets the REDUCED tensor put in its place, no copy back; a .grad that is
    # not None at install time raises; the slabs of a second gradient tensor in one backward make finish raise
    leaf = torch.zeros(5, 3, 2, requires_grad=True)
    g4 = (torch.arange(30, dtype=torch.float32).reshape(5, 3, 2) + 100.0 * rank)
    sl.install(leaf)
    sl.remove()
    for i, (a, b) in enumerate(((0, 2), (2, 3), (3, 5))):
        sl._hook(i, g4[a:b])
    leaf.grad = g4.clone()                      # what AccumulateGrad does while the slab views hold the tensor
    sl.finish(leaf.grad)
    slab_ok = slab_ok and leaf.grad.data_ptr() == g4.data_ptr() and torch.equal(leaf.grad, want3)
    try:
        sl.install(leaf)
        slab_ok = False
    except RuntimeError as e:
        slab_ok = slab_ok and ".grad must be None" in str(e)
    finally:
        sl.remove()
    g5, g6 = torch.ones(4, 2, 2) * (rank + 1), torch.ones(4, 2, 2)
    sl._hook(0, g5[0:2]); sl._hook(1, g5[2:4]); sl._hook(0, g6[0:2])
    try:
        sl.finish(g5)
        slab_ok = False
    except RuntimeError as e:
        slab_ok = slab_ok and "two renders" in str(e) and torch.equal(g5, torch.full((4, 2, 2), 3.0)) and not sl.fired()
    score = torch.tensor(0.5 + 0.1 * ((rank * 7) % 3))
    best_score, best_pose, best_rank = xd.multistart_best(score, pose.matrix[lo])
    out_q.put(dict(rank=rank, bounds=(lo, hi), same=torch.allclose(gathered, full, atol=1e-6), ok_async=ok_async,
                   grad=grad[0, 0, 0].item(), bucket_ok=bucket_ok, slab_ok=slab_ok, best_rank=best_rank, best_score=best_score.item(),
                   best_pose_ok=torch.allclose(best_pose, pose.matrix[xd.shard_bounds(B, best_rank, world)[0]])))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_pose_sharded_render_world2_gloo(B):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in procs), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [d["bounds"] for d in res] == ([(0, 2), (2, 4)] if B == 4 else [(0, 3), (3, 5)])
    for d in res:
        assert d["same"] and d["ok_async"], "all-gathered shards must equal the single-process render"
        assert d["grad"] == 3.0  # 1 + 2
        assert d["bucket_ok"], "bucketed async all-reduce of the volume gradient: the in-place sum over both ranks"
        assert d["slab_ok"], "slab-by-slab all-reduce of the volume gradient (SlabAllReduce): in place, and into a copied .grad"
        assert d["best_rank"] == 1 and abs(d["best_score"] - 0.6) < 1e-6 and d["best_pose_ok"]


def test_shard_bounds_cover_and_balance():
    from xvr_amd.distributed import shard_bounds, shard_counts

    for n in (0, 1, 7, 8, 116, 117):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            counts = shard_counts(n, world)
            assert sum(counts) == n and max(counts) - min(counts) <= 1
