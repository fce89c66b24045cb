"""Forward(+jacobian) render time for SMALL batches (one registration pose) at several detector sizes, with the
sample-split factor forced (the option fwd_split) vs chosen automatically.  Run on the GPU box."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402

dev = torch.device("cuda")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vol, _ = make_phantom(size, n_ellipsoids=16, seed=0, device=dev)
sub = read(vol, spacing=(256.0 / size,) * 3, orientation="AP")
rot, xyz = torch.tensor([[3.1, 0.05, -0.02]]), torch.tensor([[5.0, 750.0, -8.0]])


from xvr_amd import _lib, renderers  # noqa: E402


def timed(fn, n=30):
    """Mean duration (us) of the render kernel alone: HIP events around the C-ABI launch (renderers.PROFILER)."""
    for _ in range(5):
        fn()
    renderers.PROFILER = []
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for name, a, b in renderers.PROFILER if name.endswith("_forward+jac")]
    renderers.PROFILER = None
    return sum(t) / len(t) * 1e3


for renderer in (sys.argv[2:] or ["trilinear", "siddon"]):
    for B in (1, 2, 4, 8):
        for det in (64, 128, 256, 512):
            drr = DRR(sub, 1020.0, det, 1.4 * 256 / det, renderer=renderer, reverse_x_axis=False,
                      voxel_shift=0.0 if renderer == "trilinear" else 0.5).to(dev)
            r = rot.repeat(B, 1).cuda().requires_grad_()
            pose_args = (r, xyz.repeat(B, 1).cuda())
            row = []
            for ns in ("1", "2", "4", "8", "16", "102", "104", "auto"):
                _lib.set_option("fwd_split", 0 if ns == "auto" else int(ns))
                row.append(f"{ns}:{timed(lambda: drr(*pose_args, parameterization='euler_angles', convention='ZXY')):7.1f}")
            print(f"{renderer:9s} B={B} det={det:3d}  fwd+jac us  " + "  ".join(row), flush=True)
