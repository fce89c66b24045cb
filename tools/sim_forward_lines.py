#!/usr/bin/env python
"""CPU count of the 128-byte lines the trilinear forward's workgroups touch per step on the benchmark geometry (512^3 -> 256^2,
DeepFluoro pose ranges, n_points 500), for candidate volume layouts and workgroup tiles.  Only geometry, no volume.

The forward is bound by fabric bandwidth (profiles/r04_fetch_calibration.txt: one L2 miss = one 128-byte line whatever part of it
is used; the launch's 2.98e8 misses are 38 GB in 5.7 ms).  A workgroup in lockstep touches, per step, the lines its 256 samples'
two 16-byte loads fall into; consecutive steps are 2 voxels apart along the ray, so almost nothing is shared between steps: the
sum over (tile, step) of distinct lines is the L1-miss traffic, and the layout that makes it small is the one to build.

Layouts: y-pair copy [x][yp][z][2] with z contiguous (today), the same with x contiguous (axes 0 and 2 swapped), and the better
of the two per pose.  Lines = 16 consecutive entries of 8 bytes along the contiguous axis.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tools.sim_gather_divergence import pose_lattices  # noqa: E402


def count_lines(s, t, N, tile, contig, D=512, steps_stride=4):
    """distinct lines per (tile, step), summed, and the number of (tile, step, sample) triples, for one pose.
    s [3], t [H, W, 3] in voxel index coordinates (index = position: a = 1, b = 0 -- the default map is x + shift - 1/2, a
    constant offset that does not change the statistics)."""
    H, W = t.shape[:2]
    th, tw = tile
    d = t - s
    alphas = np.linspace(0.0, 1.0, N)[::steps_stride]
    lines_total, samples_total = 0, 0
    for ty in range(0, H, th):
        for tx in range(0, W, tw):
            dd = d[ty:ty + th, tx:tx + tw].reshape(-1, 3)                     # [256, 3]
            p = s[None, None, :] + alphas[:, None, None] * dd[None]            # [K, 256, 3]
            f = np.floor(p).astype(np.int64)
            inside = ((f >= -1) & (f <= D - 1)).all(-1)                         # taps can touch the volume
            if contig == 2:
                row0 = (np.clip(f[..., 0], 0, D - 1) * (D + 1) + np.clip(f[..., 1] + 1, 0, D)) * (D // 16 + 1) + np.clip(f[..., 2], 0, D - 2) // 16
                row0b = (np.clip(f[..., 0], 0, D - 1) * (D + 1) + np.clip(f[..., 1] + 1, 0, D)) * (D // 16 + 1) + (np.clip(f[..., 2], 0, D - 2) + 1) // 16
                step = (D + 1) * (D // 16 + 1)                                 # x0 -> x1 row
                ids = np.stack([row0, row0b, row0 + step, row0b + step], -1)
            else:   # x contiguous: entries (z, yp, x..x+1) -- the 16-byte load is the pair along x, the second load the next z
                row0 = (np.clip(f[..., 2], 0, D - 1) * (D + 1) + np.clip(f[..., 1] + 1, 0, D)) * (D // 16 + 1) + np.clip(f[..., 0], 0, D - 2) // 16
                row0b = (np.clip(f[..., 2], 0, D - 1) * (D + 1) + np.clip(f[..., 1] + 1, 0, D)) * (D // 16 + 1) + (np.clip(f[..., 0], 0, D - 2) + 1) // 16
                step = (D + 1) * (D // 16 + 1)
                ids = np.stack([row0, row0b, row0 + step, row0b + step], -1)
            ids = np.where(inside[..., None], ids, -1)
            for k in range(ids.shape[0]):
                u = np.unique(ids[k])
                lines_total += len(u) - (1 if u[0] == -1 else 0)
            samples_total += int(inside.sum())
    return lines_total * steps_stride, samples_total * steps_stride


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=12)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    args = ap.parse_args()
    S, T = pose_lattices(args.size, args.det, 116, seed=0)
    pick = np.linspace(0, 115, args.poses).astype(int)
    rows = []
    for b in pick:
        s, t = S[b], T[b]
        dcen = t[args.det // 2, args.det // 2] - s
        dcen = dcen / np.linalg.norm(dcen)
        r = {"pose": int(b), "dir": dcen}
        for tile in ((16, 16), (32, 8), (8, 32)):
            for contig in (2, 0):
                L, n = count_lines(s, t, 500, tile, contig, args.size)
                r[(tile, contig)] = L
                r["samples"] = n
        rows.append(r)
        print(f"pose {b:3d} dir ({dcen[0]:+.2f} {dcen[1]:+.2f} {dcen[2]:+.2f})  samples {r['samples']:.3e}  lines per sample: "
              + "  ".join(f"{tile[0]}x{tile[1]}/{'z' if c == 2 else 'x'} {r[(tile, c)] / r['samples']:.3f}" for tile in ((16, 16), (32, 8), (8, 32)) for c in (2, 0)),
              flush=True)
    tot = sum(r["samples"] for r in rows)
    print("\nmean lines per sample (x 1.81e9 samples per launch = lines per launch; x 128 B):")
    for tile in ((16, 16), (32, 8), (8, 32)):
        z = sum(r[(tile, 2)] for r in rows) / tot
        x = sum(r[(tile, 0)] for r in rows) / tot
        best = sum(min(r[(tile, 2)], r[(tile, 0)]) for r in rows) / tot
        print(f"  tile {tile[0]:2d} rows x {tile[1]:2d} cols: z-contiguous {z:.3f} ({z * 1.81e9 * 128 / 1e9:.0f} GB)   x-contiguous {x:.3f}   "
              f"better of the two per pose {best:.3f} ({best * 1.81e9 * 128 / 1e9:.0f} GB)")


if __name__ == "__main__":
    main()
