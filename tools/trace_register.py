"""One pyramid stage of Registrar.run at 512^3 / 256^2 under rocprofv3 --kernel-trace: which kernels make
up a graph-replayed iteration.  Run on the GPU box under rocprofv3."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.pose import convert  # noqa: E402
from xvr_amd.registrar import Registrar  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=16, seed=0, device=dev)
det = int(os.environ.get("DET", "256"))
drr = DRR(read(vol, orientation="AP"), 1020.0, det, 0.1360 * 8 * 256 / det, renderer="trilinear", reverse_x_axis=False, voxel_shift=0.0).to(dev)
rot, xyz = torch.tensor([[3.1, 0.05, -0.02]]), torch.tensor([[5.0, 750.0, -8.0]])
with torch.no_grad():
    gt = drr(convert(rot + 0.03, xyz + 5.0, parameterization="euler_angles", convention="ZXY").cuda())
out = Registrar(drr, scales="1", n_itrs="60", max_n_plateaus=100).run(gt, convert(rot, xyz, parameterization="euler_angles", convention="ZXY"))
torch.cuda.synchronize()
print("iterations", len(out["nccs"]) - 1, "ms/iter (last 20)", sum(out["times"][-20:]) / 20 * 1e3)
