"""One registration pose (512^3 CT, 256^2 detector, B = 1) through the forward on the natural layout and on the tiled y-pair copy,
split and unsplit: does the layout that pays for large launches (fewer lines per sample) pay in the latency regime too?
Run on the GPU box:  python tools/exp_small_batch_tiles.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd import _lib, renderers  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=16, seed=0, device=dev)
for det in (256, 512):
    drr = DRR(read(vol, orientation="AP"), 1020.0, det, 0.1360 * 8 * 256 / det, renderer="trilinear", reverse_x_axis=False, voxel_shift=0.0).to(dev)
    rot, xyz = torch.tensor([[3.1, 0.05, -0.02]], device=dev).requires_grad_(), torch.tensor([[5.0, 750.0, -8.0]], device=dev).requires_grad_()

    def timed(tag, n=40):
        for _ in range(4):
            drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        torch.cuda.synchronize()
        renderers.PROFILER = []
        for _ in range(n):
            drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        torch.cuda.synchronize()
        ev, renderers.PROFILER = renderers.PROFILER, None
        ts = sorted(a.elapsed_time(b) * 1e3 for name, a, b in ev if name.startswith("trilinear_forward"))
        print(f"{det}^2  {tag:44s} forward + jacobian: median {ts[len(ts) // 2]:7.1f} us  (min {ts[0]:.1f})", flush=True)

    for rep in range(2):      # (twice: the order of the variants must not be what is measured)
        renderers.YPAIR_MIN_WAVEFRONTS = 1 << 30
        timed("auto split factor, natural layout")
        with _lib.option("fwd_split", 1):
            timed("unsplit kernel, natural layout")
            renderers.YPAIR_MIN_WAVEFRONTS = 0
            timed("unsplit kernel, tiled y-pair copy")
        renderers.YPAIR_MIN_WAVEFRONTS = 2048
        timed("product (tiles from 2048 wavefronts)")
