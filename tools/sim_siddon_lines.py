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


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: would a SLAB-SYNCHRONOUS walk pay?  All 64 rays of a wavefront advance one unit slab of the
# wavefront's dominant axis at a time; inside the slab a ray owns 1-4 voxels (its segments between two planes of that axis).
#   direct   every lane loads its own voxels: load slot q of the slab = the q-th voxel of every ray that has one (masked);
#   staged   ("LDS-staged slab tiles"): the wavefront loads the bounding box of all its rays' voxels in the slab with
#            row-contiguous cooperative loads (64 voxels per instruction), the lanes then read their voxels from LDS.
# Layouts: natural [x][y][z], and a copy whose 128-byte lines are 4 x 8 patches ACROSS the dominant axis ("tiled": needs one
# copy per axis).  Cost of a load instruction from the gather microbenchmark: 15 + 12 (lines - 1) clocks.
# ---------------------------------------------------------------------------------------------------------------------
def line_ids(v, layout, m):
    o = [a for a in range(3) if a != m]
    if layout == "natural":
        off = (v[:, 0] * D + v[:, 1]) * D + v[:, 2]
    else:   # [m][o0 / 4][o1 / 8][4][8]
        off = ((v[:, m] * (D // 4) + v[:, o[0]] // 4) * (D // 8) + v[:, o[1]] // 8) * 32 + (v[:, o[0]] % 4) * 8 + v[:, o[1]] % 8
    return np.unique(off // 32)


cost = lambda n_lines: 15 + 12 * (n_lines - 1)
tot = {k: 0.0 for k in ("now natural", "now bricks", "direct natural", "direct tiled", "staged natural", "staged tiled")}
instr = {k: 0 for k in tot}
slots_used, slots_needed, n_slabs = 0, 0, 0
rng = np.random.default_rng(1)
for b in range(args.poses):
    for _ in range(max(1, args.tiles // args.poses)):
        ty, tx = rng.integers(4, H // 8 - 4, size=2)
        rays = [(ty * 8 + i) * H + tx * 8 + j for i in range(8) for j in range(8)]
        seqs = [voxels_of_ray(src_i[b, 0], tgt_i[b, r]) for r in rays]
        if min(len(q) for q in seqs) == 0:
            continue
        dmean = np.abs(np.mean([tgt_i[b, r] - src_i[b, 0] for r in rays], axis=0))
        m = int(np.argmax(dmean))
        # today's walk: every ray's k-th segment in one load
        n = max(len(q) for q in seqs)
        for k in range(n):
            v = np.array([q[k] for q in seqs if len(q) > k])
            for lay, key in (("natural", "now natural"), ("bricks", "now bricks")):
                tot[key] += cost(lines(v, lay)); instr[key] += 1
        # slab-synchronous
        lo = min(q[:, m].min() for q in seqs); hi = max(q[:, m].max() for q in seqs)
        for j in range(lo, hi + 1):
            per_ray = [q[q[:, m] == j] for q in seqs]
            depth = max(len(p) for p in per_ray)
            if depth == 0:
                continue
            n_slabs += 1
            slots_needed += sum(len(p) for p in per_ray); slots_used += 64 * depth
            allv = np.concatenate([p for p in per_ray if len(p)])
            for qslot in range(depth):
                v = np.array([p[qslot] for p in per_ray if len(p) > qslot])
                for lay in ("natural", "tiled"):
                    tot[f"direct {lay}"] += cost(line_ids(v, lay, m).size); instr[f"direct {lay}"] += 1
            o = [a for a in range(3) if a != m]
            b0, b1 = allv[:, o].min(axis=0), allv[:, o].max(axis=0)
            box = np.array([[j if a == m else 0 for a in range(3)]], dtype=np.int64).repeat((b1[0] - b0[0] + 1) * (b1[1] - b0[1] + 1), axis=0)
            gi, gj = np.meshgrid(np.arange(b0[0], b1[0] + 1), np.arange(b0[1], b1[1] + 1), indexing="ij")
            box[:, o[0]], box[:, o[1]] = gi.ravel(), gj.ravel()
            for lay in ("natural", "tiled"):
                for c0 in range(0, len(box), 64):
                    tot[f"staged {lay}"] += cost(line_ids(box[c0:c0 + 64], lay, m).size); instr[f"staged {lay}"] += 1
print(f"\nslab-synchronous walk along the wavefront's dominant axis: {slots_needed / max(n_slabs, 1) / 64:.2f} voxels per ray and slab, "
      f"{slots_used / max(n_slabs, 1) / 64:.2f} load slots per slab (lane use {slots_needed / max(slots_used, 1):.2f})")
base = tot["now bricks"]
for k in tot:
    print(f"{k:15s}: {instr[k]:7d} load instructions, {tot[k] / max(instr[k], 1):6.1f} clocks each, texture-address clocks relative to today's bricked walk: {tot[k] / base:.2f}"
          + ("  (+ an LDS write and 1-4 LDS reads per lane and slab, and the branch-free 4-slot walk's VALU)" if k.startswith("staged") else ""))
