"""How many distinct 128-byte cache lines one wavefront load of the Siddon forward touches, for volume layouts (CPU
simulation at the benchmark geometry; the texture-address unit's cost per load grows by ~12 clocks per extra line,
profiles/r01_microbench_gather_lines.txt).  Layouts: natural [x][y][z]; tiled [y][x/4][z/8][4][8] (a line = a 4 x 8 patch
across the y axis); bricks [x/4][y/2][z/4][4][2][4].
    python tools/sim_siddon_lines.py [--poses 12 --tiles 24]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import deepfluoro_poses  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--poses", type=int, default=12)
ap.add_argument("--tiles", type=int, default=24)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--det", type=int, default=256)
args = ap.parse_args()
D, H = args.size, args.det
rng = np.random.default_rng(0)

from oracle.diffdrr_restated import rays_from_pose, _apply  # noqa: E402  (CPU geometry only)
pose = deepfluoro_poses(args.poses, seed=0)
delx = 1.08821875 * 256 / H
src, tgt = rays_from_pose(pose.matrix, H, H, 1020.0, delx, delx, 0.0, 0.0)
# world -> index space of a D^3 volume with 1 mm voxels centred at the origin, orientation "AP" as in bench.py (xvr_amd.data.read)
from xvr_amd.data import make_phantom, read  # noqa: E402
sub = read(torch.zeros(D, D, D), orientation="AP")
affinv = torch.linalg.inv(torch.as_tensor(sub.affine, dtype=torch.float32))[None] if hasattr(sub, "affine") else None
src_i, tgt_i = _apply(affinv, src).numpy().astype(np.float64), _apply(affinv, tgt).numpy().astype(np.float64)


def voxels_of_ray(s, t):
    d = t - s
    a = []
    for ax in range(3):
        planes = np.arange(0, D + 1) - 0.5
        with np.errstate(divide="ignore", invalid="ignore"):
            al = (planes - s[ax]) / d[ax]
        a.append(al[(al > 0) & (al < 1)])
    lo = max(min((-0.5 - s[k]) / d[k], (D - 0.5 - s[k]) / d[k]) for k in range(3))
    hi = min(max((-0.5 - s[k]) / d[k], (D - 0.5 - s[k]) / d[k]) for k in range(3))
    if not hi > lo:
        return np.zeros((0, 3), dtype=np.int64)
    al = np.sort(np.concatenate(a + [np.array([lo, hi])]))
    al = al[(al >= lo) & (al <= hi)]
    mid = 0.5 * (al[1:] + al[:-1])
    p = s[None] + mid[:, None] * d[None]
    v = np.rint(p).astype(np.int64)
    return v[((v >= 0) & (v < D)).all(axis=1)]


def lines(v, layout):
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    if layout == "natural":
        off = (x * D + y) * D + z
    elif layout == "tiled_y":
        off = ((y * (D // 4) + x // 4) * (D // 8) + z // 8) * 32 + (x % 4) * 8 + z % 8
    else:
        off = (((x // 4) * (D // 2) + y // 2) * (D // 4) + z // 4) * 32 + (x % 4) * 8 + (y % 2) * 4 + z % 4
    return np.unique(off // 32).size


res = {k: [] for k in ("natural", "tiled_y", "bricks")}
for b in range(args.poses):
    for _ in range(args.tiles // args.poses + 1):
        ty, tx = rng.integers(4, H // 8 - 4, size=2)
        rays = [(ty * 8 + i) * H + tx * 8 + j for i in range(8) for j in range(8)]
        seqs = [voxels_of_ray(src_i[b, 0], tgt_i[b, r]) for r in rays]
        n = max(len(q) for q in seqs)
        for k in range(0, n, 7):
            v = np.array([q[k] for q in seqs if len(q) > k])
            if len(v) >= 32:
                for lay in res:
                    res[lay].append(lines(v, lay))
for lay, r in res.items():
    r = np.array(r)
    print(f"{lay:8s}: {r.mean():.2f} lines per wavefront load (median {np.median(r):.0f}, 90 % {np.percentile(r, 90):.0f}) -> ~{15 + 12 * (r.mean() - 1):.0f} clocks per load")
