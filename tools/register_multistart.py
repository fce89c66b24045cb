"""Multi-start registration across ranks (configs[3] in miniature unless --size 512):
    python -m torch.distributed.run --nproc-per-node N tools/register_multistart.py [--backend gloo --single-device]"""
import argparse
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.metrics import DoubleGeodesicSE3  # noqa: E402
from xvr_amd.pose import RigidTransform, convert  # noqa: E402
from xvr_amd.registrar import Registrar, register_multistart  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--det", type=int, default=128)
ap.add_argument("--starts", type=int, default=4)
ap.add_argument("--backend", default="nccl")
ap.add_argument("--single-device", action="store_true")
ap.add_argument("--batched", action="store_true", help="refine this rank's starts as one batch")
ap.add_argument("--scales", default="4,2")
ap.add_argument("--itrs", default="60,40")
args = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
vol, _ = make_phantom(args.size, n_ellipsoids=10, seed=8, device=dev)
spacing = 128.0 / args.size
drr = DRR(read(vol, spacing=(spacing,) * 3, orientation="AP"), 1020.0, args.det, 1.4 * 128 / args.det, renderer="trilinear",
          reverse_x_axis=False, voxel_shift=0.0).to(dev)
true_rot, true_xyz = torch.tensor([[3.10, 0.05, -0.03]]), torch.tensor([[4.0, 700.0, -6.0]])
true_pose = convert(true_rot, true_xyz, parameterization="euler_angles", convention="ZXY")
with torch.no_grad():
    gt = drr(convert(true_rot, true_xyz, parameterization="euler_angles", convention="ZXY").to(dev))
g = torch.Generator().manual_seed(0)   # the same starts on every rank; each rank takes its slice
drot = (torch.rand(args.starts, 3, generator=g) - 0.5) * 2 * 0.17      # +-10 degrees
dxyz = (torch.rand(args.starts, 3, generator=g) - 0.5) * 2 * 20.0      # +-20 mm
inits = convert(true_rot + drot, true_xyz + dxyz, parameterization="euler_angles", convention="ZXY")
reg = Registrar(drr, scales=args.scales, n_itrs=args.itrs, patience=6, max_n_plateaus=2)
import time  # noqa: E402
torch.cuda.synchronize()
t0 = time.perf_counter()
score, pose, best_rank, local = register_multistart(reg, gt, inits, batched=args.batched)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
geo = DoubleGeodesicSE3(1020.0)
best = RigidTransform(pose.cpu()[None])
ang, _, dbl = geo(true_pose, best)
err = dbl.item()
rot_deg = ang.item() / (0.5 * 1020.0) * 57.29578                      # the angular part is sdd / 2 x angle
dt = best.convert("euler_angles", "ZXY")[1][0] - true_xyz[0]         # camera frame: y runs along the view, x and z across it
across, along = float((dt[0] ** 2 + dt[2] ** 2).sqrt()), float(dt[1].abs())
errs0 = [geo(true_pose, inits[i])[2].item() for i in range(args.starts)]
print(f"rank {rank}/{world}: refined {len(local)} starts; best ncc {score.item():.4f} from rank {best_rank}; "
      f"pose error {err:.2f} mm = {rot_deg:.3f} deg, {across:.2f} mm across / {along:.2f} mm along the view "
      f"(starts were {min(errs0):.1f}-{max(errs0):.1f} mm off); "
      f"{sum(len(r['trajectory']) for r in local)} iterations in {wall:.2f} s{' (batched)' if args.batched else ''}", flush=True)
# per-rank timeline: what this rank spent on each of its starts (no communication until the final 68-byte all-gather)
for n, r in enumerate(local):
    print(f"  rank {rank} start {n}: {len(r['trajectory'])} iterations, {r['runtime']:.2f} s, ncc {r['nccs'][0]:.3f} -> {r['nccs'][-1]:.3f}", flush=True)
if world > 1:
    dist.destroy_process_group()
