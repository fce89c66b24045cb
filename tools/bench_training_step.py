"""The render-side of one xvr training step (configs[4]; /root/reference/src/xvr/model/trainer.py:185-230) at full
size: HU -> density with a random bone multiplier, render #1 (no grad, 8 label channels), render #2 at the predicted
poses with the pose gradient, PoseRegressionLoss and its backward.  The timm regressor is out of scope: the
"network output" is a leaf tensor of pose parameters.  Run on the GPU box."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read, transform_hu_to_density  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.loss import PoseRegressionLoss  # noqa: E402
from xvr_amd.metrics import XrayTransforms  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402
from xvr_amd.training import get_random_pose, render_samples  # noqa: E402

dev = torch.device("cuda")
B, H, size = 116, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol, lab = make_phantom(size, n_ellipsoids=64, n_labels=8, seed=0, device=dev)
hu = vol * 1400 - 1000
drr = DRR(read(hu, lab, spacing=(512.0 / size,) * 3, orientation="AP", hu=True), 1020.0, H, 1.08821875, renderer="trilinear",
          reverse_x_axis=False).to(dev)
drr.register_buffer("volume", hu)
transforms = XrayTransforms(H)
lossfn = PoseRegressionLoss(1020.0).to(dev)
g = torch.Generator().manual_seed(0)
marks = {}


def lap(name, e_prev):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.setdefault(name, []).append((e_prev, e))
    return e


def step():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B, generator=g).to(dev)
    e = lap("sample poses", e)
    tmp = transform_hu_to_density(drr.volume, float(torch.empty(1).uniform_(1.0, 10.0, generator=g)))
    e = lap("HU -> density", e)
    with torch.no_grad():
        img, mask, keep = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pose)
    e = lap("render #1 (no grad, 8 channels)", e)
    rot, xyz = pose.convert("quaternion_adjugate")
    rot = (rot + 0.01 * torch.randn(rot.shape, generator=g).to(dev)).requires_grad_()
    xyz = (xyz + 5.0 * torch.randn(xyz.shape, generator=g).to(dev)).requires_grad_()
    pred_pose = convert(rot, xyz, parameterization="quaternion_adjugate")
    e = lap("stand-in for the regressor", e)
    pred_img, pred_mask, _ = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pred_pose)
    e = lap("render #2 (grad)", e)
    loss, *_ = lossfn(transforms(img), mask, pose, transforms(pred_img), pred_mask, pred_pose)
    e = lap("transforms + PoseRegressionLoss", e)
    loss.mean().backward()
    e = lap("backward (loss + render)", e)
    return loss


import time  # noqa: E402

g.manual_seed(0)
for _ in range(3):
    step()
marks.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n * 1e3
print(f"training step (render side), {size}^3 -> {H}^2, batch {B}, 8 labels: {total:.2f} ms")
for k, v in marks.items():
    print(f"  {k:38s} {sum(a.elapsed_time(b) for a, b in v) / len(v):7.2f} ms")
