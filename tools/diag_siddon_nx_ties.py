"""Diagnostic (GPU): per ray, how much length the slab-march forward and the splat backward credit to the two voxel planes either
side of the structural tie of dims = shape + 1 on an even-sized axis."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import make_case  # noqa: E402
from xvr_amd import _lib, renderers  # noqa: E402
from xvr_amd.renderers import render  # noqa: E402
from xvr_amd.spec import RenderSpec  # noqa: E402

shape = (40, 36, 44)
spec = RenderSpec(renderer="siddon", norm_dims_offset=1)
H, W = 24, 20
case = make_case(seed=5, shape=shape, height=H, width=W, delx=0.9 * max(shape) / H, xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
src, tgt, img = (case[k].cuda() for k in ("source", "target", "img"))
ylo = shape[1] // 2 - 1
probe = torch.zeros(shape, device="cuda")
probe[:, ylo, :] = 1.0


def fwd(slab):
    with _lib.option("siddon_slab", slab), _lib.option("fwd_split", 1):
        return render(probe, src, tgt, img, spec, None, ray_grid_w=W).detach()[:, 0]


def bwd(r, b, splat, gather=True):
    renderers.VOXEL_GATHER = gather
    vol = torch.rand(shape, device="cuda", requires_grad=True)
    w = torch.zeros(2, 1, H * W, device="cuda")
    w[b, 0, r] = 1.0
    with _lib.option("siddon_splat", splat), _lib.option("fwd_split", 1):
        out = render(vol, src, tgt, img, spec, None, ray_grid_w=W)
        (out * w).sum().backward()
    renderers.VOXEL_GATHER = True
    return vol.grad[:, ylo, :].sum().item()


fs, fm = fwd(1), fwd(2)
n_bad = 0
for b in range(2):
    for r in range(0, H * W, 3):
        s, sc = bwd(r, b, 1), bwd(r, b, 1, gather=False)
        a, m = fs[b, r].item(), fm[b, r].item()
        if abs(s - a) > 1e-3 or abs(sc - m) > 1e-3:
            n_bad += 1
            if n_bad <= 25:
                d = (tgt[b, r] - src[b, 0]).tolist()
                print(f"pose {b} ray {r}: fwd slab {a:.4f} bwd splat {s:.4f} | fwd merge {m:.4f} bwd scatter {sc:.4f} | d = {[round(x, 3) for x in d]}")
print("rays with a forward / backward mismatch:", n_bad)
