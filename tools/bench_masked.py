"""Timing of the masked (mask_to_channels) trilinear render at C2/C5 size: forward and backward to the
pose, C = 8 channels.  Run on the GPU box."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.training import get_random_pose, render_samples  # noqa: E402

dev = torch.device("cuda")
B, H = 116, 256
vol, lab = make_phantom(512, n_ellipsoids=64, n_labels=8, seed=0, device=dev)
drr = DRR(read(vol, lab, orientation="AP"), 1020.0, H, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
g = torch.Generator().manual_seed(0)
pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B, generator=g)
rot, xyz = pose.convert("euler_angles", "ZXY")
rot, xyz = rot.to(dev).requires_grad_(True), xyz.to(dev).requires_grad_(True)
from xvr_amd.pose import convert  # noqa: E402


def run(masked, grad):
    p = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    seg = drr.mask if masked else None
    if not grad:
        with torch.no_grad():
            return render_samples(drr, drr.density, seg, drr.affine_inverse, p)
    img, mask, keep = render_samples(drr, drr.density, seg, drr.affine_inverse, p)
    rot.grad = xyz.grad = None
    img.sum().backward()
    return img, mask, keep


for masked in (False, True):
    for grad in (False, True):
        for _ in range(2):
            run(masked, grad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = run(masked, grad)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"masked={masked!s:5} grad={grad!s:5}: {ms:7.2f} ms / batch of {B}  ({B / ms * 1e3:8.0f} DRR/s)  channels={out[1].shape[1]} kept={int(out[2].sum())}")
