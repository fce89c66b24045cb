"""A detector far beyond the benchmark's (1000 x 1100 pixels, not multiples of any tile) on a 201 x 181 x 221 volume, two poses: the
lattice kernels against the general ones (ray_grid_w = 0), the voxel gradient's gather / splat against the atomic scatter, and the
adjoint identity <A v, w> = <v, A^T w> -- HIP against HIP (the oracle needs minutes for 2.2e6 rays), every renderer variant.
    python tools/check_big_detector.py        (on the GPU box)"""
import sys
from pathlib import Path

import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from conftest import make_case  # noqa: E402
from xvr_amd import renderers  # noqa: E402
from xvr_amd.renderers import render  # noqa: E402
from xvr_amd.spec import RenderSpec  # noqa: E402

# optional: H W depth_mm [only-siddon].  `python tools/check_big_detector.py 3500 3400 60 1`: ONE source INSIDE the volume and 1.19e7 pixels --
# every brick's footprint is then the whole detector, 2.97e6 pixels per wavefront quarter, beyond what k_siddon_splat's float
# estimate of (pixel / columns) resolves without its integer correction (fixed in round 6)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1100)
DEPTH = float(sys.argv[3]) if len(sys.argv) > 3 else None
ONLY_SIDDON = len(sys.argv) > 4
SHAPE = (201, 181, 221)     # (odd sizes: no structural tie under dims = shape + 1, conftest.has_structural_tie -- the splat carries the march's
                            #  plane alphas, the scatter the merge walk's: comparable voxel by voxel only without one)
if DEPTH is None:
    case = make_case(seed=3, shape=SHAPE, height=H, width=W, delx=0.25, xyz=((3.0, 420.0, -2.0), (-8.0, 300.0, 5.0)))
else:
    case = make_case(seed=3, shape=SHAPE, height=H, width=W, delx=275.0 / max(H, W), rot=((170.0, 10.0, 5.0),), xyz=((3.0, DEPTH, -2.0),))
NB = case["source"].shape[0]
g = torch.Generator().manual_seed(1)
case["volume"] = torch.rand(SHAPE, generator=g)
w = torch.rand(NB, 1, H * W, generator=g).cuda()
ok = True
KWS = (dict(renderer="trilinear", n_points=300), dict(renderer="trilinear", n_points=300, clip_to_volume=True), dict(renderer="siddon"),
       dict(renderer="siddon", norm_dims_offset=1))
for kw in (KWS[2:] if ONLY_SIDDON else KWS):
    spec = RenderSpec(**kw)
    res = {}
    for name, grid_w, gather in (("lattice", W, True), ("general", 0, True), ("scatter", W, False)):
        renderers.VOXEL_GATHER = gather
        vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
        vol.requires_grad_(True)
        out = render(vol, src, tgt, img, spec, None, ray_grid_w=grid_w)
        (out * w).sum().backward()
        res[name] = (out.detach(), vol.grad.detach())
        renderers.VOXEL_GATHER = True
    o, gv = res["lattice"]
    lhs, rhs = (o.double() * w.double()).sum().item(), (gv.double() * case["volume"].cuda().double()).sum().item()
    e_img = ((o - res["general"][0]).abs().max() / o.abs().max()).item()
    e_g1 = ((gv - res["scatter"][1]).abs().max() / gv.abs().max()).item()
    bad_g = int(((gv - res["scatter"][1]).abs() > 1e-4 * gv.abs().max()).sum())
    # (dims + 1: the splat carries the march's plane alphas, the scatter the merge walk's -- ~5e-5 of the 6.6e8 lookups land on the other
    #  side of a threshold and move a segment between two neighbouring voxels: the totals agree, a few 1e-3 of the voxels differ)
    sums = abs(gv.double().sum().item() - res["scatter"][1].double().sum().item()) <= 1e-5 * res["scatter"][1].double().abs().sum().item()
    good = abs(lhs - rhs) <= 3e-5 * abs(lhs) and e_img <= 1e-4 and sums and (e_g1 <= 2e-4 or (kw.get("norm_dims_offset") and bad_g <= 5e-3 * gv.numel()))
    ok = ok and good and bool(torch.isfinite(gv).all())
    print(f"{kw}: adjoint {abs(lhs - rhs) / abs(lhs):.1e}; lattice vs general image {e_img:.1e}; gather / splat vs scatter {e_g1:.1e} ({bad_g} voxels beyond 1e-4)  {'ok' if good else 'FAILED'}")
print("big detector:", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
