"""Loop-trip statistics of the trilinear voxel gather at the benchmark's size (HISTORY.md section 9): how many
(lane, pose) visits, steps, detector rows (and how many of them empty), candidates and wavefront-level inner trips one
backward takes.  Builds a diagnostic copy of the library (-DXVR_GATHER_STATS) next to the product one and loads it
through XVR_DRR_LIBRARY.  Run on the GPU box:  python tools/gather_stats.py [siddon]"""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from xvr_amd.build import build_diagnostic_library  # noqa: E402

stats_lib = build_diagnostic_library("XVR_GATHER_STATS", ROOT / "tools" / "_build" / "libxvr_drr_stats.so")
os.environ["XVR_DRR_LIBRARY"] = str(stats_lib)

import torch  # noqa: E402

from xvr_amd import _lib  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.training import get_random_pose  # noqa: E402

SIDDON = len(sys.argv) > 1 and sys.argv[1] == "siddon"
dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
drr = DRR(read(vol, orientation="AP"), 1020.0, 256, 1.08821875, renderer="siddon" if SIDDON else "trilinear", reverse_x_axis=False).to(dev)
pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, 116,
                       generator=torch.Generator().manual_seed(0)).to(dev)
density = drr.density.clone().requires_grad_()
raw = ctypes.CDLL(str(_lib.library_path()))
out = (ctypes.c_ulonglong * 8)()
drr(pose, density=density).sum().backward()          # warm-up
torch.cuda.synchronize()
raw.xvr_drr_debug_gather_stats(out, 1)
drr(pose, density=density).sum().backward()
torch.cuda.synchronize()
raw.xvr_drr_debug_gather_stats(out, 0)
v = [float(x) for x in out]
if SIDDON:   # k_siddon_gather_vol2: one lane = one 2 x 2 x 2 block, one wavefront = one 8^3 brick
    print(f"(lane, pose) visits with a window {v[0]:.4g} | candidates {v[4]:.4g} ({v[4] / max(v[0], 1):.2f} per visit) | wavefront visits {v[7]:.4g} | "
          f"wavefront rows {v[2]:.4g} ({v[2] / max(v[7], 1):.2f} per visit) | wavefront trips {v[6]:.4g} ({v[6] / max(v[7], 1):.2f} per visit, "
          f"{v[4] / max(v[6], 1):.1f} of 128 candidate slots filled per trip)")
    print(f"wavefront visits served from LDS {v[1]:.4g} ({100 * v[1] / max(v[7], 1):.1f} %, {v[5] / max(v[1], 1):.0f} pixels staged each) | "
          f"on sign-sorted planes {v[3]:.4g} ({100 * v[3] / max(v[7], 1):.1f} %)")
    sys.exit(0)
print(f"(lane, pose) visits {v[0]:.4g} | steps {v[1]:.4g} | rows {v[2]:.4g} (empty {v[3]:.4g}) | candidates {v[4]:.4g} | "
      f"wavefront inner trips {v[6]:.4g} | wavefront pose iterations {v[7]:.4g}")
print(f"steps per visit {v[1] / v[0]:.2f} | rows per step {v[2] / v[1]:.2f} | empty rows {100 * v[3] / v[2]:.1f} % | "
      f"candidates per non-empty row {v[4] / (v[2] - v[3]):.2f} | candidate slots filled per inner trip {v[4] / (2 * v[6]):.1f} of 64 | "
      f"inner trips per wavefront pose iteration {v[6] / v[7]:.1f}")
