"""Does the ORDER of the poses in a batch matter to the forward (C2: 116 poses, 512^3 -> 256^2)?  Blocks are numbered tile-strip major
with all poses of a strip in flight together; poses that are neighbours in the batch are neighbours in the block order.  The same
116 poses in the sampler's order, sorted by yaw, and in a snake through (pitch bins, yaw): forward + jacobian by HIP events.
Run on the GPU box:  python tools/exp_pose_order.py"""
import sys
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
from bench import deepfluoro_poses  # noqa: E402
from xvr_amd import renderers  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
drr = DRR(read(vol, orientation="AP"), 1020.0, 256, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
rot, xyz = deepfluoro_poses(116, seed=0).convert("euler_angles", "ZXY")
orders = {"sampler's order": torch.arange(116)}
orders["sorted by yaw"] = torch.argsort(rot[:, 0])
bins = ((rot[:, 1] - rot[:, 1].min()) / (rot[:, 1].max() - rot[:, 1].min() + 1e-6) * 6).floor()
key = bins * 1000 + torch.where(bins % 2 == 0, rot[:, 0], -rot[:, 0]) * 10
orders["snake through (pitch bins, yaw)"] = torch.argsort(key)
depth_bins = ((xyz[:, 1] - xyz[:, 1].min()) / (xyz[:, 1].max() - xyz[:, 1].min() + 1e-6) * 4).floor()
orders["sorted by (depth bin, yaw)"] = torch.argsort(depth_bins * 1000 + rot[:, 0] * 10)
orders["interleaved: yaw-sorted, stride 8"] = torch.argsort(rot[:, 0])[torch.cat([torch.arange(k, 116, 8) for k in range(8)])]
for rep in range(2):
    for name, idx in orders.items():
        r, x = rot[idx].to(dev).requires_grad_(), xyz[idx].to(dev).requires_grad_()
        for _ in range(4):
            drr(r, x, parameterization="euler_angles", convention="ZXY", n_points=500)
        torch.cuda.synchronize()
        renderers.PROFILER = []
        for _ in range(10):
            drr(r, x, parameterization="euler_angles", convention="ZXY", n_points=500)
        torch.cuda.synchronize()
        ev, renderers.PROFILER = renderers.PROFILER, None
        ts = sorted(a.elapsed_time(b) for n_, a, b in ev if n_.startswith("trilinear_forward"))
        print(f"{name:38s} forward + jacobian: median {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f})", flush=True)
