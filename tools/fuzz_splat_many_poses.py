"""k_siddon_splat with more poses than one pass of the kernel takes (128): 130..400 random poses per case on small volumes, the recalled
index maps; the march + splat pair against the adjoint identity, the splat against the atomic scatter (totals; voxel by voxel on
tie-free shapes), two runs bit for bit.    python tools/fuzz_splat_many_poses.py [first_seed=0] [count=40]      (on the GPU box)"""
import sys
from pathlib import Path

import numpy as np
import torch

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
from conftest import has_structural_tie, make_case  # noqa: E402
from test_siddon_splat import _differing, _voxel_grad  # noqa: E402
from xvr_amd.spec import RenderSpec  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(47000 + seed)
    kw = [dict(norm_dims_offset=1), dict(norm_dims_offset=1, voxel_shift=0.0), dict(align_corners=True)][seed % 3]
    shape = tuple(int(x) for x in rng.integers(10, 50, size=3))
    B, H, W = int(rng.integers(130, 400)), int(rng.integers(8, 36)), int(rng.integers(8, 36))
    ext = float(max(shape))
    rot = tuple((float(rng.uniform(0, 360)), float(rng.uniform(-60, 60)), float(rng.uniform(-30, 30))) for _ in range(B))
    xyz = tuple((float(rng.uniform(-0.2, 0.2) * ext), float(rng.uniform(0.2, 6.0) * ext), float(rng.uniform(-0.2, 0.2) * ext)) for _ in range(B))
    what = f"seed {seed}: {kw} shape {shape} det {H}x{W} B {B}"
    try:
        case = make_case(seed=seed, shape=shape, height=H, width=W, rot=rot, xyz=xyz, delx=float(rng.uniform(0.3, 1.5)) * ext / max(H, W))
        case["volume"] = torch.rand(shape, generator=torch.Generator().manual_seed(seed))
        spec = RenderSpec(renderer="siddon", **kw)
        w = torch.randn(B, 1, H * W, generator=torch.Generator().manual_seed(seed))
        out, g = _voxel_grad(case, spec, w, W)[:2]
        assert torch.equal(g, _voxel_grad(case, spec, w, W)[1]), "two runs differ"
        lhs, rhs = (out.double() * w.cuda().double()).sum().item(), (g.double() * case["volume"].cuda().double()).sum().item()
        scale = (out.double().abs() * w.cuda().double().abs()).sum().item()
        assert abs(lhs - rhs) <= 1e-6 * scale, f"adjoint {lhs} {rhs} (scale {scale})"
        sc = _voxel_grad(case, spec, w, W, gather=False)[1]
        # (totals: no tie moves weight in or out of the volume.  The upstream gradient has both signs: the voxel sums cancel, and the
        #  fixed-point rounding of a few 1e-6 of the largest voxel each does not -- seed 101: per pose the totals agree to 5e-7)
        assert abs(g.double().sum().item() - sc.double().sum().item()) <= 1e-4 * sc.double().abs().sum().item(), "totals"
        if not any(has_structural_tie(S, **{"voxel_shift": 0.5, **kw}) for S in shape):
            assert _differing(g, sc) <= 8 + int(2.5e-3 * B * H * W), f"{_differing(g, sc)} voxels differ"
    except BaseException as e:  # noqa: BLE001
        bad.append(seed)
        print(f"FAILED {what}: {type(e).__name__}: {str(e)[:300]}")
print(f"splat with many poses: {count} cases from {first}, {len(bad)} failed {bad}")
