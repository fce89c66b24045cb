"""BASELINE.json configs[4] across ranks: pose-regressor training with ONE CT PER GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_multigpu.py
        [--size 512 --det 256 --batch 116 --steps 8]           (on one GPU: --backend gloo --single-device)

Every rank owns its own pre-registered volume (here: a seeded phantom, seed = rank, with 8 label channels) and runs the
render side of xvr's training step on it (/root/reference/src/xvr/model/trainer.py:185-230): HU -> density with a random
bone multiplier, render #1 at sampled poses (no grad, mask -> channels), the regressor's prediction, render #2 at the
predicted poses with the pose gradient, PoseRegressionLoss, backward.  Volume-parallel data parallelism (SURVEY.md
section 8e, C5): no collective inside a step; the ONLY exchange is the all-reduce of the regressor's gradients every
`n_grad_accum_itrs` = 4 steps (/root/reference/src/xvr/config/trainer.py:31) as one flat bucket over RCCL.

The timm ResNet is out of scope: the stand-in regressor is a small conv stem + a wide linear
head sized to ResNet-18's 11.7 M parameters (47 MB of fp32 gradients -- the bucket the reference would all-reduce), so
the collective moves the right number of bytes; its accuracy is not the point.
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read, transform_hu_to_density  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.loss import PoseRegressionLoss  # noqa: E402
from xvr_amd.metrics import XrayTransforms  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402
from xvr_amd.training import get_random_pose, render_samples  # noqa: E402


class StandInRegressor(torch.nn.Module):
    """image [B,1,H,W] -> (quaternion-adjugate rotation [B,10], translation [B,3]) as a residual on a given pose."""

    def __init__(self, n_params: int):
        super().__init__()
        self.stem = torch.nn.Sequential(torch.nn.Conv2d(1, 16, 7, stride=4, padding=3), torch.nn.ReLU(),
                                        torch.nn.Conv2d(16, 32, 3, stride=2, padding=1), torch.nn.ReLU(),
                                        torch.nn.AdaptiveAvgPool2d(4))
        width = max((n_params - 6000) // (512 + 13), 8)
        self.wide = torch.nn.Linear(512, width)
        self.head = torch.nn.Linear(width, 13)
        torch.nn.init.zeros_(self.head.weight)
        torch.nn.init.zeros_(self.head.bias)

    def forward(self, img, rot0, xyz0):
        h = self.head(torch.relu(self.wide(self.stem(img).flatten(1))))
        return rot0 + 0.01 * h[:, :10], xyz0 + h[:, 10:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--batch", type=int, default=116)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--accum", type=int, default=4, help="n_grad_accum_itrs")
    ap.add_argument("--params", type=int, default=11_700_000, help="parameters of the stand-in regressor (ResNet-18: 11.7 M)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))

    B, H = args.batch, args.det
    vol, lab = make_phantom(args.size, n_ellipsoids=32, n_labels=8, seed=rank, device=dev)   # this rank's own CT
    hu = vol * 1400 - 1000
    drr = DRR(read(hu, lab, spacing=(512.0 / args.size,) * 3, orientation="AP", hu=True), 1020.0, H, 1.08821875 * 256 / H,
              renderer="trilinear", reverse_x_axis=False).to(dev)
    drr.register_buffer("volume", hu)
    transforms = XrayTransforms(H)
    lossfn = PoseRegressionLoss(1020.0).to(dev)
    torch.manual_seed(0)                        # identical initial weights on every rank
    net = StandInRegressor(args.params).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    params = [p for p in net.parameters()]
    nparam = sum(p.numel() for p in params)
    bucket = torch.zeros(nparam, device=dev)    # one flat bucket: a single collective per accumulation window
    g = torch.Generator().manual_seed(1000 + rank)
    t_coll = []

    def step(itr):
        pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B, generator=g).to(dev)
        tmp = transform_hu_to_density(drr.volume, float(torch.empty(1).uniform_(1.0, 10.0, generator=g)))
        with torch.no_grad():
            img, mask, keep = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pose)
        rot0, xyz0 = pose.convert("quaternion_adjugate")
        rot, xyz = net(transforms(img), rot0, xyz0)
        pred_pose = convert(rot, xyz, parameterization="quaternion_adjugate")
        pred_img, pred_mask, _ = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pred_pose)
        loss, *_ = lossfn(transforms(img), mask, pose, transforms(pred_img), pred_mask, pred_pose)
        (loss.mean() / args.accum).backward()
        if (itr + 1) % args.accum == 0:
            if world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch._foreach_copy_(list(bucket.split([p.numel() for p in params])), [p.grad.reshape(-1) for p in params])
                e0.record()
                dist.all_reduce(bucket)          # RCCL over xGMI: nparam * 4 bytes, once per `accum` steps
                e1.record()
                t_coll.append((e0, e1))
                bucket.div_(world)
                for p, chunk in zip(params, bucket.split([p.numel() for p in params])):
                    p.grad.copy_(chunk.view_as(p))
            opt.step()
            opt.zero_grad(set_to_none=False)
        return loss.detach().mean()

    for i in range(args.warmup * args.accum):
        step(i)
    t_coll.clear()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    coll_ms = sum(a.elapsed_time(b) for a, b in t_coll) / max(len(t_coll), 1) if t_coll else 0.0
    # every rank holds the same weights after the all-reduce: check it (one scalar per rank)
    digest = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1)
    same = True
    if world > 1:
        alld = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(alld, digest)
        same = all(torch.allclose(alld[0], d, rtol=1e-9, atol=0.0) for d in alld)
    print(f"rank {rank}/{world}: {args.steps} steps of batch {B} on its own {args.size}^3 CT -> {H}^2, {dt / args.steps * 1e3:.2f} ms/step, "
          f"{2 * B * args.steps / dt:.0f} DRRs/s on this rank ({2 * B * args.steps * world / dt:.0f} over {world} ranks); regressor {nparam / 1e6:.2f} M "
          f"parameters, gradient all-reduce every {args.accum} steps: {coll_ms:.2f} ms for {nparam * 4 / 1e6:.1f} MB; "
          f"weights identical across ranks: {same}; loss {last.item():.4f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
