"""Random-shape soak of the round-3 glue kernels against their torch formulations (run on the GPU box):
render_samples' tail (xvr_drr_foreground), boolean Dice (xvr_sim_dice_bool), Standardize -> Normalize (xvr_sim_transform_*)
and convert (xvr_pose_convert_*).  python tools/fuzz_glue.py [n=200]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import xvr_amd.pose as P  # noqa: E402
from oracle import loss_restated as oloss, metrics_restated as mref  # noqa: E402  (the checker)
from xvr_amd.loss import DiceMetric  # noqa: E402
from xvr_amd.metrics import XrayTransforms  # noqa: E402
from xvr_amd.training import _Foreground  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
bad = []
for it in range(n):
    B, C, H, W = int(rng.integers(1, 9)), int(rng.integers(1, 13)), int(rng.integers(1, 70)), int(rng.integers(1, 70))
    g = torch.Generator().manual_seed(it)
    img = torch.rand(B, C, H, W, generator=g)
    img[torch.rand(B, C, H, W, generator=g) < float(rng.uniform(0.2, 0.95))] = 0.0
    x = img.cuda()
    thr = 0.10 if C == 1 else 0.05
    tot, msk, keep = _Foreground.apply(x, thr)
    m0 = x > 0
    t0 = x.sum(dim=1, keepdim=True)
    k0 = (m0.to(x).flatten(1).mean(1) > thr) if C == 1 else ((m0[:, 1:].sum(dim=1, keepdim=True) > 0).to(x).flatten(1).mean(1) > thr)
    if not (torch.equal(msk, m0) and torch.equal(keep, k0) and torch.allclose(tot, t0, rtol=1e-6, atol=1e-7)):
        bad.append(("foreground", it, (B, C, H, W)))
    if C >= 2:
        a, b = (torch.rand(B, C, H, W, generator=g) < 0.4).cuda(), (torch.rand(B, C, H, W, generator=g) < 0.5).cuda()
        f = DiceMetric()(a, b)
        r = oloss.dice_metric(a.float(), b.float())
        if not torch.equal(f.nan_to_num(-1.0), r.nan_to_num(-1.0)):
            bad.append(("dice", it, (B, C, H, W)))
    for per_image in (False, True):
        xx = (torch.rand(B, 1, H, W, generator=g) * 5).cuda()
        w = torch.randn(B, 1, H, W, generator=g).cuda()
        tf = XrayTransforms(H, W, per_image=per_image)
        out = []
        for fused in (True, False):
            v = xx.clone().requires_grad_()
            y = tf(v) if fused else mref.xray_transforms(v, H, W, per_image=per_image)
            (y * w).sum().backward()
            out.append((y.detach(), v.grad))
        scale = out[1][1].abs().max().clamp_min(1e-6)
        if not (torch.equal(out[0][0], out[1][0]) and float((out[0][1] - out[1][1]).abs().max()) <= 5e-5 * float(scale)):
            bad.append(("transform", it, (B, H, W, per_image), float((out[0][1] - out[1][1]).abs().max() / scale)))
    par = ["euler_angles", "axis_angle", "quaternion", "quaternion_adjugate", "rotation_6d", "se3_log_map"][it % 6]
    k = P.N_ANGULAR_COMPONENTS[par]
    conv = ["ZXY", "XYZ", "ZYX", "YXZ", "XZX", "ZYZ"][int(rng.integers(0, 6))] if par == "euler_angles" else None
    Bp = int(rng.integers(1, 70))
    rot = torch.randn(Bp, k, generator=g) * float(rng.uniform(0.1, 3.0))
    if par == "quaternion_adjugate":
        q = torch.randn(Bp, 4, generator=g)
        rot = P.quaternion_to_quaternion_adjugate(q) + 0.01 * torch.randn(Bp, 10, generator=g)
    xyz = torch.randn(Bp, 3, generator=g) * 200
    wm = torch.randn(Bp, 4, 4, generator=g).cuda()
    res = []
    for fused in (True, False):
        P.FUSED_CONVERT = fused
        r_, t_ = rot.clone().cuda().requires_grad_(), xyz.clone().cuda().requires_grad_()
        m = P.convert(r_, t_, parameterization=par, convention=conv).matrix
        (m * wm).sum().backward()
        res.append((m.detach(), r_.grad, t_.grad))
    P.FUSED_CONVERT = True
    sc = res[1][1].abs().max().clamp_min(1.0)
    if not (torch.allclose(res[0][0], res[1][0], rtol=3e-5, atol=5e-4) and float((res[0][1] - res[1][1]).abs().max()) <= 5e-4 * float(sc)
            and torch.allclose(res[0][2], res[1][2], rtol=2e-4, atol=2e-4)):
        bad.append(("convert", it, par, conv, Bp, float((res[0][1] - res[1][1]).abs().max() / sc)))
print(f"{n} random cases per kernel family; failures: {len(bad)}")
for b in bad[:20]:
    print("  ", b)
sys.exit(min(len(bad), 100))
