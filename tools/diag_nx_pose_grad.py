"""Where does the pose gradient of the Siddon render under dims = shape + 1 differ from the oracle's?  One benchmark pose at 511^3:
per-ray d/d target and d/d source of a weighted image sum, HIP against the float64 oracle (and the float32 oracle beside it).
    python tools/diag_nx_pose_grad.py [pose=8] [size=511]        (on the GPU box)"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import to_oracle_spec  # noqa: E402
from oracle.diffdrr_restated import _apply, rays_from_pose, render as oracle_render  # noqa: E402
from test_configs import deepfluoro_poses  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 511
knobs = dict(norm_dims_offset=1) if (len(sys.argv) <= 3 or sys.argv[3] != "exact") else {}
H = 256
vol, _ = make_phantom(size, n_ellipsoids=64, seed=0)
sub = read(vol, orientation="AP")
drr = DRR(sub, 1020.0, H, 1.08821875, renderer="siddon", reverse_x_axis=False, **knobs).cuda()
rot0, xyz0 = deepfluoro_poses(116, seed=0).convert("euler_angles", "ZXY")
rot0, xyz0 = rot0[b:b + 1], xyz0[b:b + 1]
w = torch.from_numpy(np.random.default_rng(1000 + b).uniform(0.0, 1.0, size=(1, 1, H * H))).to(torch.float32)
with torch.no_grad():
    pose = convert(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
    s_w, t_w = rays_from_pose(pose.matrix, H, H, 1020.0, 1.08821875, 1.08821875, 0.0, 0.0, "AP", False)
    L = (t_w - s_w).norm(dim=-1).unsqueeze(1)
    affinv = torch.linalg.inv(sub.affine)[None]
    s_v, t_v = _apply(affinv, s_w), _apply(affinv, t_w)
spec = drr.renderer._spec()
hs, ht, hl = (x.clone().cuda().requires_grad_(True) for x in (s_v, t_v, L))
hout = drr.renderer(drr.density, hs, ht, hl)
(hout * w.cuda()).sum().backward()
res = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    os_, ot, ol = (x.clone().to(dt).requires_grad_(True) for x in (s_v, t_v, L))
    oout = oracle_render(vol.to(dt), os_, ot, ol, to_oracle_spec(spec), chunk=2048)
    (oout * w.to(dt)).sum().backward()
    res[name] = (oout.detach().double(), os_.grad.double(), ot.grad.double(), ol.grad.double())
o, gs, gt, gl = res["f64"]
print("pose", b, "size", size, knobs, "image max", o.max().item())
for name, got in (("HIP", (hout.detach().double().cpu(), hs.grad.double().cpu(), ht.grad.double().cpu(), hl.grad.double().cpu())), ("oracle f32", res["f32"])):
    img, a_s, a_t, a_l = got
    print(f"{name}: image max err {((img - o).abs().max() / o.max()).item():.2e};  d/d source {((a_s - gs).abs().max() / gs.abs().max()).item():.2e} "
          f"(values {a_s.reshape(-1, 3).sum(0).tolist()} vs {gs.reshape(-1, 3).sum(0).tolist()})")
    e = (a_t - gt).reshape(-1, 3)
    per = e.abs().amax(dim=1) / gt.abs().max()
    print(f"   d/d target: rays beyond 2e-3: {(per > 2e-3).double().mean().item():.3e}, worst {per.max().item():.2e};  SUM over rays of the difference "
          f"{e.sum(0).tolist()} against sum |grad| {gt.reshape(-1, 3).abs().sum(0).tolist()};  sum of grad {gt.reshape(-1, 3).sum(0).tolist()}")
    # where on the detector: 4 x 4 blocks of the summed difference (x component)
    blk = e[:, 0].reshape(H, H).reshape(4, H // 4, 4, H // 4).sum(dim=(1, 3))
    print("   difference of d/d target_x summed per 64 x 64 block:\n", np.array2string(blk.numpy(), precision=2))
    print(f"   d/d raylen err {((a_l - gl).abs().max() / gl.abs().max()).item():.2e}")
