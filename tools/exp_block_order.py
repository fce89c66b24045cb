"""Forward time of the benchmark batch under the block -> (pose, tile) orders of map_ray (options block_order /
order_group).  Run on the GPU box."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import deepfluoro_poses  # noqa: E402
from xvr_amd import _lib, renderers  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402

dev = torch.device("cuda")
B, H = 116, 256
rot, xyz = (t.to(dev) for t in deepfluoro_poses(B, seed=0).convert("euler_angles", "ZXY"))
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
for renderer in ("trilinear", "siddon"):
    drr = DRR(read(vol, orientation="AP"), 1020.0, H, 1.08821875, renderer=renderer, reverse_x_axis=False).to(dev)
    kw = {"n_points": 500} if renderer == "trilinear" else {}
    r = rot.clone().requires_grad_(True)
    for order, group in ((0, "16x4"), (2, "16x4"), (2, "4x16"), (2, "2x16"), (2, "1x16"), (2, "16x1"), (2, "8x16"), (2, "8x2")):
        gx, gy = map(int, group.split("x"))
        _lib.set_option("block_order", order)
        _lib.set_option("order_group", gx | gy << 8)
        for _ in range(2):
            drr(r, xyz, parameterization="euler_angles", convention="ZXY", **kw)
        renderers.PROFILER = []
        for _ in range(4):
            drr(r, xyz, parameterization="euler_angles", convention="ZXY", **kw)
        torch.cuda.synchronize()
        ev, renderers.PROFILER = renderers.PROFILER, None
        t = [a.elapsed_time(b) for k, a, b in ev if k.startswith(renderer)]
        print(f"{renderer} order {order} group {group:>5s}: forward+jac {sum(t) / len(t):.3f} ms", flush=True)
