"""Latency of one registration iteration (configs[3]: 512^3 CT, one pose, 256^2 then 512^2 detector):
render -> XrayTransforms -> mNCC + gradNCC -> backward -> Adam.  Run on the GPU box."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.metrics import XrayTransforms  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402
from xvr_amd.registrar import Registrar  # noqa: E402
from xvr_amd.registration import Registration  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
sub = read(vol, orientation="AP")
for H, delx in ((256, 0.1360 * 8), (512, 0.1360 * 4)):
    drr = DRR(sub, 1020.0, H, delx, renderer="trilinear", reverse_x_axis=False, voxel_shift=0.0).to(dev)
    rot, xyz = torch.tensor([[3.1, 0.05, -0.02]], device=dev), torch.tensor([[5.0, 750.0, -8.0]], device=dev)
    with torch.no_grad():
        gt = drr(convert(rot + 0.03, xyz + 5.0, parameterization="euler_angles", convention="ZXY"))
    reg = Registration(drr, rot, xyz, "euler_angles", "ZXY")
    R = Registrar(drr)
    tf = XrayTransforms(H)
    img = tf(gt)
    opt = torch.optim.Adam([{"params": [reg.rotation], "lr": 1e-2}, {"params": [reg.translation], "lr": 1.0}], maximize=True)

    def it():
        opt.zero_grad()
        loss = R.imagesim(img, tf(reg()))
        loss.sum().backward()
        opt.step()
        return loss

    for _ in range(5):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        it()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    # render only
    t0 = time.perf_counter()
    for _ in range(n):
        out = reg()
    torch.cuda.synchronize()
    ms_r = (time.perf_counter() - t0) / n * 1e3
    print(f"detector {H}^2: eager {ms:.2f} ms / iteration (render fwd only {ms_r:.2f} ms)")
    init = convert(rot.cpu(), xyz.cpu(), parameterization="euler_angles", convention="ZXY")
    for use_graph in (False, True):
        Rg = Registrar(drr, scales="1", n_itrs="120", use_graph=use_graph, max_n_plateaus=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = Rg.run(gt, init)
        torch.cuda.synchronize()
        n_it = len(out["nccs"]) - 1
        steady = sum(out["times"][-50:]) / 50 * 1e3
        print(f"  Registrar.run use_graph={use_graph}: {n_it} iterations, {steady:.2f} ms / iteration (last 50), ncc {out['nccs'][0]:.4f} -> {out['nccs'][-1]:.4f}")
    for kw in (dict(sigma=1.0), dict(equalize=True), dict(sigma=1.0, equalize=True)):   # configurations beyond the single fused call
        Rk = Registrar(drr, scales="1", n_itrs="120", max_n_plateaus=100, **kw)
        torch.cuda.synchronize()
        out = Rk.run(gt, init)
        torch.cuda.synchronize()
        steady = sum(out["times"][-50:]) / 50 * 1e3
        print(f"  Registrar.run {kw}: {len(out['nccs']) - 1} iterations, {steady:.2f} ms / iteration (last 50), ncc {out['nccs'][0]:.4f} -> {out['nccs'][-1]:.4f}")
    g = torch.Generator().manual_seed(0)
    B = 8
    inits = convert(rot.cpu() + (torch.rand(B, 3, generator=g) - 0.5) * 0.06, xyz.cpu() + (torch.rand(B, 3, generator=g) - 0.5) * 10.0,
                    parameterization="euler_angles", convention="ZXY")
    Rb = Registrar(drr, scales="1", n_itrs="120", max_n_plateaus=100)
    torch.cuda.synchronize()
    outs = Rb.run_batch(gt, inits)
    torch.cuda.synchronize()
    steady = sum(outs[0]["times"][-50:]) / 50 * 1e3
    print(f"  Registrar.run_batch, {B} starts in one batch: {steady:.2f} ms / iteration of all {B} (= {steady / B:.3f} ms per pose-iteration), "
          f"final ncc {min(o['nccs'][-1] for o in outs):.4f} .. {max(o['nccs'][-1] for o in outs):.4f}")
