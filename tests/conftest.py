import dataclasses
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def to_oracle_spec(spec):
    """xvr_amd.spec.RenderSpec -> oracle RenderSpec (two independent definitions, same fields)."""
    from oracle.diffdrr_restated import RenderSpec as OSpec

    return OSpec(**dataclasses.asdict(spec))


def make_case(shape=(20, 24, 28), height=12, width=10, sdd=400.0, delx=6.0, n_labels=3, seed=0,
              rot=((170.0, 10.0, 5.0), (20.0, -20.0, -8.0)), xyz=((5.0, 300.0, -4.0), (-3.0, 200.0, 6.0)),
              spacing=(1.0, 1.0, 1.0)):
    """A small seeded render case in the renderer's own input space (voxel-index source/target)."""
    from oracle.diffdrr_restated import _apply, rays_from_pose
    from xvr_amd.pose import convert

    g = torch.Generator().manual_seed(seed)
    vol = torch.rand(*shape, generator=g)
    mask = (torch.rand(*shape, generator=g) * n_labels).floor()
    affine = torch.diag(torch.tensor([*spacing, 1.0]))
    affine[:3, 3] = -(affine[:3, :3] @ ((torch.tensor(shape, dtype=torch.float32) - 1) / 2))
    pose = convert(torch.tensor(rot), torch.tensor(xyz), parameterization="euler_angles",
                   convention="ZXY", degrees=True)
    src, tgt = rays_from_pose(pose.matrix, height, width, sdd, delx, delx, 0.0, 0.0)
    img = (tgt - src).norm(dim=-1).unsqueeze(1)
    affinv = torch.linalg.inv(affine)[None]
    return dict(volume=vol, mask=mask, source=_apply(affinv, src), target=_apply(affinv, tgt), img=img,
                affine=affine, pose=pose, height=height, width=width, sdd=sdd, delx=delx)


def accuracy_by_magnitude(ref, candidates, decades=7):
    """Relative error of voxel gradients binned by the magnitude of the reference: for decade d, the voxels with
    10^-(d+1) < |ref| / max|ref| <= 10^-d.  ``candidates``: name -> tensor.  Returns a list of rows
    {"decade": d, "voxels": n, name: {"median": ..., "p99": ..., "max": ...}} (relative to each voxel's OWN reference
    value -- what a per-voxel optimiser such as Adam sees -- not to the largest gradient)."""
    import torch

    ref = ref.detach().double().cpu().flatten()
    top = ref.abs().max().item()
    rows = []
    for d in range(decades):
        sel = (ref.abs() <= top * 10.0 ** -d) & (ref.abs() > top * 10.0 ** -(d + 1))
        n = int(sel.sum())
        row = {"decade": d, "voxels": n}
        if n:
            r = ref[sel]
            for name, g in candidates.items():
                e = ((g.detach().double().cpu().flatten()[sel] - r).abs() / r.abs())
                k99 = max(int(0.99 * n) - 1, 0)
                row[name] = {"median": e.median().item(), "p99": e.kthvalue(k99 + 1).values.item(), "max": e.max().item()}
        rows.append(row)
    return rows


def format_accuracy_table(rows, names):
    lines = ["| |g| / max|g| | voxels | " + " | ".join(f"{n}: median / p99 / max" for n in names) + " |",
             "|---|---|" + "---|" * len(names)]
    for r in rows:
        cells = [f"{r[n]['median']:.1e} / {r[n]['p99']:.1e} / {r[n]['max']:.1e}" if n in r else "-" for n in names]
        lines.append(f"| 1e-{r['decade'] + 1} .. 1e-{r['decade']} | {r['voxels']} | " + " | ".join(cells) + " |")
    return "\n".join(lines)


def has_structural_tie(size: int, voxel_shift=0.5, norm_dims_offset=0, align_corners=False, tol=2e-3) -> bool:
    """Non-exact Siddon index maps (idx = rint(a x + b), oracle/diffdrr_restated.py::index_map) can put the MIDPOINT of a fully
    crossed plane cell exactly on a rounding boundary: under dims = shape + 1 the middle cell of an even-sized axis maps to
    k + 1/2 in exact arithmetic, and the last bit of alpha then decides which of two voxels gets the whole segment -- in torch
    and in every HIP kernel family differently.  True if an axis of this size has such a cell (comparisons ACROSS
    implementations use sizes without one; the forward / backward pair of one family is held to the adjoint identity on sizes WITH)."""
    S = size
    dims = S + norm_dims_offset
    a, b = ((S - 1) / dims, voxel_shift * (S - 1) / dims) if align_corners else (S / dims, voxel_shift * S / dims - 0.5)
    for c in range(S):
        t = a * (c + 0.5 - voxel_shift) + b
        if abs((t - 0.5) - round(t - 0.5)) < tol:
            return True
    return False


def tie_free_shape(rng, lo, hi, **map_kw):
    """Three sizes in [lo, hi) none of which has a structural tie under the map (see has_structural_tie)."""
    out = []
    while len(out) < 3:
        s = int(rng.integers(lo, hi))
        if not has_structural_tie(s, **map_kw):
            out.append(s)
    return tuple(out)
