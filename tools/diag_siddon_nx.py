"""Diagnostic (GPU): forward / voxel-gradient consistency of the Siddon paths under a non-exact index map.
Adjoint identity <A v, w> = <v, A^T w> for every (forward, backward) pairing, and where splat and scatter differ."""
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

shape = tuple(int(x) for x in (sys.argv[1:4] or (41, 37, 45)))
kw = dict(norm_dims_offset=1)
spec = RenderSpec(renderer="siddon", **kw)
case = make_case(seed=5, shape=shape, height=96, width=80, delx=0.9 * max(shape) / 96, xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
case["volume"] = torch.rand(shape, generator=torch.Generator().manual_seed(8))
w = torch.rand(2, 1, 96 * 80, generator=torch.Generator().manual_seed(9)).cuda()


def run(slab, splat, gather):
    renderers.VOXEL_GATHER = gather
    vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
    vol.requires_grad_(True)
    with _lib.option("siddon_slab", slab), _lib.option("siddon_splat", splat), _lib.option("fwd_split", 1):
        out = render(vol, src, tgt, img, spec, None, ray_grid_w=80)
        (out * w).sum().backward()
    renderers.VOXEL_GATHER = True
    return out.detach().double(), vol.grad.double()


v = case["volume"].cuda().double()
res = {}
for name, (slab, splat, gather) in {"slab+splat": (1, 1, True), "slab+scatter": (1, 1, False), "merge+cells": (2, 0, True),
                                    "merge+scatter": (2, 1, False), "merge+splat": (2, 1, True)}.items():
    out, g = run(slab, splat, gather)
    lhs, rhs = (out * w.double()).sum().item(), (g * v).sum().item()
    res[name] = (out, g)
    print(f"{name:14s} <Av,w> = {lhs:.6f}  <v,A^T w> = {rhs:.6f}  rel {abs(lhs - rhs) / abs(lhs):.2e}")
a, b = res["slab+splat"][1], res["slab+scatter"][1]
d = (a - b).abs() > 1e-4 * b.abs().max()
idx = d.nonzero()
print("splat vs scatter: differing voxels", int(d.sum()), "of", d.numel())
print(idx[:40].tolist())
print("values", [(round(a[tuple(i)].item(), 4), round(b[tuple(i)].item(), 4)) for i in idx[:12]])
fo = (res["slab+splat"][0] - res["merge+scatter"][0]).abs()
print("forward slab vs merge: rays differing > 1e-4", int((fo > 1e-4 * res["merge+scatter"][0].abs().max()).sum()), "of", fo.numel())
