"""Registration iteration time by pose parameterisation, with and without the fused convert (xvr_pose_convert_*).  Run on the GPU box."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xvr_amd.data import make_phantom, read
from xvr_amd.drr import DRR
from xvr_amd.pose import convert
import xvr_amd.pose as P
from xvr_amd.registrar import Registrar
dev = torch.device("cuda")
vol, _ = make_phantom(256, n_ellipsoids=40, seed=0, device=dev)
drr = DRR(read(vol, spacing=(2.0, 2.0, 2.0), orientation="AP"), 1020.0, 256, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
rot = torch.tensor([[3.1, 0.05, -0.02]]); xyz = torch.tensor([[5.0, 780.0, -8.0]])
with torch.no_grad():
    gt = drr(convert((rot + 0.03).to(dev), (xyz + 5.0).to(dev), parameterization="euler_angles", convention="ZXY"))
init = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
for par in ("euler_angles", "se3_log_map", "quaternion_adjugate", "axis_angle"):
    for fused in (True, False):
        P.FUSED_CONVERT = fused
        R = Registrar(drr, scales="1", n_itrs="120", max_n_plateaus=100, parameterization=par, convention="ZXY" if par == "euler_angles" else None)
        out = R.run(gt, init)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = R.run(gt, init)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{par:20s} fused_convert={fused}: {1e3 * dt / 120:.3f} ms / iteration, ncc {out['nccs'][0]:.4f} -> {out['nccs'][-1]:.4f}")
P.FUSED_CONVERT = True
