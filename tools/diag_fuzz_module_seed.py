"""Seeds of tests/test_hip_parity.py::test_fuzz_drr_module_end_to_end_against_the_oracle whose pose gradient misses the float32 oracle:
is the gradient ill-conditioned (a sample on a voxel boundary / a tie-broken crossing: the float32 oracle then misses its own float64
run as well) or is the HIP path wrong?   python tools/diag_fuzz_module_seed.py 100860 100515 ...   (on the GPU box)"""
import sys
from pathlib import Path

import numpy as np
import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from conftest import to_oracle_spec  # noqa: E402
from oracle.diffdrr_restated import drr_from_pose  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402

for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(5000 + seed)
    size = int(rng.integers(12, 33))
    vol, lab = make_phantom(size, n_ellipsoids=5, n_labels=3, seed=seed)
    spacing = tuple(float(x) for x in rng.uniform(1.0, 3.0, size=3))
    orientation = str(rng.choice(["AP", "PA"]))
    rev = bool(rng.random() < 0.5)
    H, W = int(rng.integers(2, 28)), int(rng.integers(2, 28))
    sdd = float(rng.uniform(400.0, 1200.0))
    delx, dely = float(rng.uniform(1.0, 4.0)), float(rng.uniform(1.0, 4.0))
    x0, y0 = float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))
    renderer = "trilinear" if rng.random() < 0.65 else "siddon"
    shift = float(rng.choice([0.0, 0.5]))
    sub = read(vol, lab, spacing=spacing, orientation=orientation)
    drr = DRR(sub, sdd, H, delx, width=W, dely=dely, x0=x0, y0=y0, renderer=renderer, reverse_x_axis=rev, voxel_shift=shift).cuda()
    B = int(rng.integers(1, 4))
    rot = torch.tensor(rng.uniform(-1.0, 1.0, size=(B, 3)) * np.array([3.0, 0.6, 0.4]), dtype=torch.float32)
    xyz = torch.tensor(np.stack([rng.uniform(-15, 15, B), rng.uniform(0.45, 0.8, B) * sdd, rng.uniform(-15, 15, B)], 1), dtype=torch.float32)
    n_points = int(rng.integers(40, 200))
    kw = {"n_points": n_points} if renderer == "trilinear" else {}
    spec = drr.renderer._spec(**kw)
    w = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(seed))
    r, t = rot.clone().cuda().requires_grad_(), xyz.clone().cuda().requires_grad_()
    out = drr(r, t, parameterization="euler_angles", convention="ZXY", **kw)
    (out * w.cuda()).sum().backward()
    g = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        ro, to = rot.clone().to(dt).requires_grad_(), xyz.clone().to(dt).requires_grad_()
        o = drr_from_pose(vol.to(dt), sub.affine.to(dt), convert(ro, to, parameterization="euler_angles", convention="ZXY").matrix, H, W, sdd, delx,
                          dely, x0, y0, to_oracle_spec(spec), orientation=orientation, reverse_x_axis=rev)
        (o * w.to(dt)).sum().backward()
        g[name] = (o.detach().double(), ro.grad.double(), to.grad.double())
    o64, r64, t64 = g["f64"]
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()   # noqa: E731
    print(f"seed {seed}: {renderer} det {H}x{W} B {B} size {size} n_points {kw.get('n_points')}")
    print(f"   image      vs f64 oracle: HIP {rel(out.detach(), o64):.2e}   f32 oracle {rel(g['f32'][0], o64):.2e}")
    print(f"   d/d rot    vs f64 oracle: HIP {rel(r.grad, r64):.2e}   f32 oracle {rel(g['f32'][1], r64):.2e}")
    print(f"   d/d xyz    vs f64 oracle: HIP {rel(t.grad, t64):.2e}   f32 oracle {rel(g['f32'][2], t64):.2e}")
    # ---- per ray: the exploded call (detector -> affine inverse -> renderer) with gradients of source / target, HIP against the float64 oracle
    from oracle.diffdrr_restated import _apply, rays_from_pose, render as oracle_render  # noqa: E402
    with torch.no_grad():
        pose = convert(rot.double(), xyz.double(), parameterization="euler_angles", convention="ZXY")
        s_w, t_w = rays_from_pose(pose.matrix, H, W, sdd, delx, dely, x0, y0, orientation, rev)
        L = (t_w - s_w).norm(dim=-1).unsqueeze(1)
        affinv = torch.linalg.inv(sub.affine.double())[None]
        s_v, t_v = _apply(affinv, s_w), _apply(affinv, t_w)
    hs, ht, hl = (x.float().clone().cuda().requires_grad_(True) for x in (s_v, t_v, L))
    ho = drr.renderer(drr.density, hs, ht, hl, **kw)
    (ho * w.reshape(B, 1, -1).cuda()).sum().backward()
    os_, ot, ol = (x.clone().requires_grad_(True) for x in (s_v, t_v, L))
    oo = oracle_render(vol.double(), os_, ot, ol, to_oracle_spec(spec), chunk=4096)
    (oo * w.reshape(B, 1, -1).double()).sum().backward()
    e = (ht.grad.double().cpu() - ot.grad).abs().amax(dim=-1) / ot.grad.abs().max()
    top = e.reshape(-1).sort(descending=True)
    print(f"   exploded call: image {rel(ho.detach(), oo.detach()):.2e}; d/d source {rel(hs.grad, os_.grad):.2e}; d/d target per ray: worst {[f'{v:.1e}' for v in top.values[:4].tolist()]} "
          f"at rays {top.indices[:4].tolist()} of {e.numel()}; rays beyond 2e-3: {int((e > 2e-3).sum())}")
    # ---- is it a kink?  the same comparison with the pose nudged: a sample on a voxel boundary leaves it, a wrong formula stays wrong
    for dlt in (1e-5, 1e-4, 1e-3):
        rr = (rot + dlt).clone()
        r_, t_ = rr.clone().cuda().requires_grad_(), xyz.clone().cuda().requires_grad_()
        (drr(r_, t_, parameterization="euler_angles", convention="ZXY", **kw) * w.cuda()).sum().backward()
        ro, to = rr.clone().double().requires_grad_(), xyz.clone().double().requires_grad_()
        (drr_from_pose(vol.double(), sub.affine.double(), convert(ro, to, parameterization="euler_angles", convention="ZXY").matrix, H, W, sdd, delx,
                       dely, x0, y0, to_oracle_spec(spec), orientation=orientation, reverse_x_axis=rev) * w.double()).sum().backward()
        print(f"   pose + {dlt:g}: d/d rot HIP vs f64 oracle {rel(r_.grad, ro.grad):.2e}, d/d xyz {rel(t_.grad, to.grad):.2e}")
