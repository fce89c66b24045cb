"""How many ds_adds of k_trilinear_splat_b16 could a merge of equal-cell contributions remove?  
For the benchmark's poses (512^3 -> 256^2, n_points 500): the fraction of samples whose 2 x 2 x 2 base cell equals that of the
previous pixel of the same detector row at the same step (what a lane-to-lane DPP merge or a register accumulator along a run
could fold), of the pixel one row down, and the mean number of distinct base cells in a 2 x 2 block of pixels.  CPU, numpy, float64
(ties are irrelevant for a fraction).   python tools/sim_splat_merge.py [n_poses]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle.diffdrr_restated import rays_from_pose  # noqa: E402  (the oracle's detector geometry: a tool, not the product)
from xvr_amd.pose import convert  # noqa: E402
from xvr_amd.training import get_random_pose  # noqa: E402

n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 12
H, N, D = 256, 500, 512
g = torch.Generator().manual_seed(0)
pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, 116, generator=g)
affinv = torch.eye(4, dtype=torch.float64)
affinv[:3, 3] = (D - 1) / 2.0
same_col, same_row, distinct4, total = 0, 0, 0.0, 0
for b in np.linspace(0, 115, n_poses).astype(int):
    s, t = rays_from_pose(pose.matrix[b:b + 1].double(), H, H, 1020.0, 1.08821875, 1.08821875, 0.0, 0.0, "AP", False)
    s, t = (s[0, 0] + affinv[:3, 3]).numpy(), (t[0] + affinv[:3, 3]).numpy().reshape(H, H, 3)
    for k in range(0, N, 7):
        al = k / (N - 1)
        p = s + al * (t - s)                       # [H, H, 3] index coordinates (voxel_shift 0.5: index = x)
        base = np.floor(p).astype(np.int64)
        inside = np.all((base >= -1) & (base < D), axis=-1)
        key = (base[..., 0] * 2048 + base[..., 1]) * 2048 + base[..., 2]
        v = inside[:, 1:] & inside[:, :-1]
        same_col += int(np.sum((key[:, 1:] == key[:, :-1]) & v))
        v2 = inside[1:, :] & inside[:-1, :]
        same_row += int(np.sum((key[1:, :] == key[:-1, :]) & v2))
        blk = np.stack([key[0::2, 0::2], key[0::2, 1::2], key[1::2, 0::2], key[1::2, 1::2]], axis=-1)
        ins = inside[0::2, 0::2] & inside[0::2, 1::2] & inside[1::2, 0::2] & inside[1::2, 1::2]
        srt = np.sort(blk[ins], axis=-1)
        distinct4 += float(np.sum(1 + np.sum(srt[:, 1:] != srt[:, :-1], axis=-1)))
        total += int(inside.sum())
        nblk = int(ins.sum())
        same_col_blocks = nblk
    print(f"pose {b}: so far same-cell as the previous pixel of the row {same_col / total:.3f}, as the pixel one row up {same_row / total:.3f}", flush=True)
print(f"samples in the volume: {total}")
print(f"fraction with the base cell of the previous pixel in the row: {same_col / total:.3f}  -> adds left after a merge along runs: {8 * (1 - same_col / total):.2f} of 8")
print(f"fraction with the base cell of the pixel one row up:          {same_row / total:.3f}")
