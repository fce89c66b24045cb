"""Randomised parity of the LARGE-launch code paths against the torch oracle.

The seeded fuzz of tests/test_hip_parity.py draws small cases, which take the alpha-split kernels, the natural volume layout and
the merge walk.  A launch of >= 2048 wavefronts takes other code: the Siddon slab march on the bricked copy (k_siddon_slab), the
trilinear march on the tiled y-pair copy (volume_layout 3, built at the third sight of a volume), the rebuilt Siddon voxel gather
with four bricks per workgroup (k_siddon_gather_vol2<true>) and the 16^3-brick splat with persistent workgroups.  Here: random
volume shapes (odd, non-multiples of the brick and tile sizes, thin slabs), random oblique poses -- every third seed with one
source INSIDE the volume --, random detector aspect and pixel size, image, pose gradients and voxel gradient against the oracle
at the suite's tolerances, and two runs bit-identical.

The volumes are SMOOTH fields (a coarse random grid, interpolated) and the upstream weights positive: with 10^5 rays and 10^7
plane crossings / samples per case, some hundred of them sit within float32 resolution of a voxel face or of another crossing,
where the pose gradient of a piecewise function jumps by the local voxel-to-voxel difference -- HIP and the oracle then pick
different one-sided derivatives for that ray.  On white-noise volumes (the small cases' choice) that is a 1e-2 error of many
rays in the maximum norm, for either implementation.  On a smooth field a handful of rays per case remain -- rays that graze a
plane family (d_x / |d| ~ 1e-2: the crossing alpha is ill-conditioned) or enter through an edge of the volume; measured on ten
seeds: 1-4 of 1.8e5 rays beyond 2e-3, and on every one of them the float32 torch oracle is as far from its own float64 run as
the HIP kernels are (tools/diag_fuzz_large.py prints them: e.g. d out / d target_x = 0.00 (slab march), 4.51 (merge walk and
float32 oracle), -0.17 (float64 oracle) for a ray with d = (-5, 402, -22)).  On the two seeds with most such rays (16 and 26 of 1.3e5 beyond 2e-3 against the float32
oracle) the HIP kernels agree with the float64 oracle to 1e-8 on two thirds of them and the float32 oracle does not.  So the
reference here is the oracle run in FLOAT64, the per-ray gradients are held to the tolerance on ALL BUT 16 RAYS per case (1e-4 of
them; or as many as the float32 oracle itself has beyond the tolerance of its float64 run, where that is more), the source gradient -- the sum over a pose's rays, which inherits those rays -- to 2e-2, and the image and the voxel
gradient to the suite's tolerances in the maximum norm."""
import numpy as np
import pytest
import torch

from conftest import make_case
from test_hip_parity import FWD_TOL, GRAD_TOL, _close, _hip_render

pytestmark = pytest.mark.gpu
MAX_OUTLIER_RAYS = 16


def _outlier_rays(a, b, tol):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.dim() == 3 and a.shape[1] == 1:
        a, b = a[:, 0, :, None], b[:, 0, :, None]
    err = (a - b).abs().amax(dim=-1) / max(b.abs().max().item(), 1e-12)
    return int((err > tol).sum()), err.numel(), err.max().item()


def _close_rays(a, b, tol, what, b32):
    """Per-ray gradients [B, n, 3] or [B, 1, n]: relative to the largest reference entry, within tol on all but a few rays -- 16, or as
    many as the float32 oracle itself has beyond tol of its float64 run where a case has more of them (a fresh seed with 21 such rays
    for the HIP kernels had 49 for the float32 oracle)."""
    bad, n, worst = _outlier_rays(a, b, tol)
    allowed = max(MAX_OUTLIER_RAYS, _outlier_rays(b32, b, tol)[0])
    assert bad <= allowed, f"{what}: {bad} of {n} rays beyond {tol:.1e} (max {worst:.3e}; allowed {allowed})"


def _oracle(case, spec, w, dtype):
    """The torch oracle in float64 / float32: (out, grad_volume, grad_source, grad_target, grad_img)."""
    from conftest import to_oracle_spec
    from oracle.diffdrr_restated import render

    vol, src, tgt, img = (case[k].to(dtype).requires_grad_(True) for k in ("volume", "source", "target", "img"))
    out = render(vol, src, tgt, img, to_oracle_spec(spec), None)
    (out * w.to(dtype)).sum().backward()
    return out.detach(), vol.grad, src.grad, tgt.grad, img.grad


def _oracle64(case, spec, w):
    return _oracle(case, spec, w, torch.float64)


def _check_grads(hip, ref, ref32):
    """(grad_volume, grad_source, grad_target, grad_img) of the HIP path against the float64 oracle's (ref32: the float32 oracle's)."""
    _close(hip[0], ref[0], GRAD_TOL, "grad_volume")
    _close(hip[1], ref[1], 2e-2, "grad_source")
    _close_rays(hip[2], ref[2], GRAD_TOL, "grad_target", ref32[2])
    _close_rays(hip[3], ref[3], GRAD_TOL, "grad_img", ref32[3])


def _smooth(shape, seed):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(1, 1, 4, 4, 4, generator=g)
    v = torch.nn.functional.interpolate(coarse, size=shape, mode="trilinear", align_corners=True)[0, 0]
    return (v + 0.02 * torch.rand(*shape, generator=g)).contiguous()


def _case(seed, n_poses):
    rng = np.random.default_rng(9000 + seed)
    shape = tuple(int(x) for x in rng.integers(6, 58, size=3))
    if seed % 5 == 0:
        shape = (shape[0], int(rng.integers(2, 5)), shape[2])          # a thin slab: fewer voxels than a brick along y
    h, w = [(128, 128), (64, 256), (192, 64), (96, 160)][seed % 4]
    n_poses = max(n_poses, -(-2048 * 64 // (h * w)))
    rot = tuple((float(rng.uniform(90, 270)), float(rng.uniform(-60, 60)), float(rng.uniform(-30, 30))) for _ in range(n_poses))
    depth = [float(rng.uniform(100, 420)) for _ in range(n_poses)]
    if seed % 3 == 0:
        depth[-1] = float(rng.uniform(0.0, 4.0))                       # this pose's source sits inside the volume
    xyz = tuple((float(rng.uniform(-12, 12)), d, float(rng.uniform(-12, 12))) for d in depth)
    if seed % 4 == 1 or __import__("os").environ.get("XVR_FUZZ_WILD") == "1":
        # two poses whose source sits BESIDE the volume, close to it, the detector looking past it: rays that miss, rays that graze a
        # face, rays with the volume behind them (the class of the slab-march bug round 5's soak found at a small launch, seed 70034)
        ext = float(max(shape))
        xyz = xyz[:-2] + tuple((float(rng.choice([-1.0, 1.0]) * rng.uniform(0.4, 1.1) * ext), float(rng.uniform(0.0, 0.6) * ext),
                                float(rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.9) * ext)) for _ in range(2))
    delx = float(rng.uniform(0.25, 1.6)) * 128.0 / max(h, w)
    case = make_case(shape=shape, height=h, width=w, seed=seed, rot=rot, xyz=xyz, delx=delx)
    case["volume"] = _smooth(shape, seed)
    return case, h, w, n_poses


@pytest.mark.parametrize("seed", range(10))
def test_large_siddon_launch_against_the_oracle(seed):
    from xvr_amd import renderers
    from xvr_amd.spec import RenderSpec

    case, h, w, n_poses = _case(seed, 8 if seed % 2 else 11)
    assert n_poses * h * w // 64 >= 2048
    spec = RenderSpec(renderer="siddon", voxel_shift=0.5 if seed % 2 else 0.0)
    wgt = torch.rand(n_poses, 1, h * w, generator=torch.Generator().manual_seed(seed))
    renderers.PROFILER = []
    hip = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    names = {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    assert "siddon_forward+jac" in names and "siddon_backward[vol]" in names, names
    again = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    assert torch.equal(hip[0], again[0]) and torch.equal(hip[1], again[1])          # image and voxel gradient: deterministic
    ref = _oracle64(case, spec, wgt)
    _close(hip[0], ref[0], FWD_TOL, "out")
    _check_grads(hip[1:], ref[1:], _oracle(case, spec, wgt, torch.float32)[1:])


@pytest.mark.parametrize("seed", range(8))
def test_large_siddon_launch_under_the_recalled_index_maps(seed):
    """Round 5: dims = shape + 1 -- what SURVEY.md Appendix A recalls for upstream's Siddon -- and align_corners
    at launch sizes that take the slab march on the BRICKED copy (k_siddon_slab<.., BRICK, NX>) and the ray-driven brick splat
    (k_siddon_splat).  Two checks.  (i) The pair against itself, on the shape as drawn (even sizes carry the map's structural tie,
    conftest.has_structural_tie): <A v, w> = <v, A^T w> to rounding, and two runs bit-identical.  (ii) Against the float64 oracle
    on a tie-free shape: a lookup a x_mid + b computed from float32 alphas lands on the other side of a threshold for ~1e-5 of the
    segments, which moves one segment of a ray to the neighbouring voxel -- so image and per-ray gradients are held on all but
    2.5e-3 of the rays (measured ~1e-3), the voxel gradient on all but as many voxels."""
    from conftest import has_structural_tie
    from xvr_amd import renderers
    from xvr_amd.spec import RenderSpec

    kw = [dict(norm_dims_offset=1), dict(norm_dims_offset=1, voxel_shift=0.0), dict(align_corners=True),
          dict(norm_dims_offset=1, align_corners=True, voxel_shift=0.0)][seed % 4]
    case, h, w, n_poses = _case(200 + seed, 9)
    assert n_poses * h * w // 64 > 2048
    spec = RenderSpec(renderer="siddon", **kw)
    wgt = torch.rand(n_poses, 1, h * w, generator=torch.Generator().manual_seed(seed))
    renderers.PROFILER = []
    hip = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    names = [e[0] for e in renderers.PROFILER]
    renderers.PROFILER = None
    assert "pack_bricks" in names and "siddon_forward+jac" in names and "siddon_backward[vol]" in names, names
    again = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    assert torch.equal(hip[0], again[0]) and torch.equal(hip[1], again[1])
    lhs = (hip[0].double() * wgt.cuda().double()).sum().item()
    rhs = (hip[1].double() * case["volume"].cuda().double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs), (kw, case["volume"].shape, lhs, rhs)
    # (ii) a tie-free shape: every size with a tie grows by one
    shape = tuple(S + 1 if has_structural_tie(S, **kw) else S for S in case["volume"].shape)
    if shape != tuple(case["volume"].shape):
        case = dict(case, volume=_smooth(shape, 300 + seed))
        hip = _hip_render(case, spec, grid_w=w, grads=True, w=wgt)
    ref, ref32 = _oracle64(case, spec, wgt), _oracle(case, spec, wgt, torch.float32)
    rays = n_poses * h * w
    allowed = 16 + int(2.5e-3 * rays)
    for name, a, b, b32, tol in (("out", hip[0], ref[0], ref32[0], FWD_TOL), ("grad_target", hip[3], ref[3], ref32[3], GRAD_TOL),
                                 ("grad_img", hip[4], ref[4], ref32[4], GRAD_TOL)):
        bad, n, worst = _outlier_rays(a, b, tol)
        assert bad <= max(allowed, _outlier_rays(b32, b, tol)[0]), f"{name}: {bad} of {n} rays beyond {tol:.1e} (max {worst:.2e}; allowed {allowed})"
    gv, rv = hip[1].double().cpu(), ref[1].double()
    bad = int(((gv - rv).abs() > GRAD_TOL * rv.abs().max()).sum())
    assert bad <= allowed, f"grad_volume: {bad} of {gv.numel()} voxels beyond {GRAD_TOL:.1e} (allowed {allowed})"
    assert abs(gv.sum().item() - rv.sum().item()) <= 1e-4 * rv.abs().sum().item()
    _close(hip[2], ref[2], 5e-2, "grad_source")


@pytest.mark.parametrize("seed", range(10))
def test_large_trilinear_launch_on_the_tiled_copy_against_the_oracle(seed, monkeypatch):
    from xvr_amd import renderers
    from xvr_amd.renderers import render
    from xvr_amd.spec import RenderSpec

    # (the third-render rule, so that the first two renders are the natural layout's to compare bits with; round 6's first-sight
    #  rule for launches of many samples per voxel: tests/test_hip_parity.py::test_large_launch_builds_the_tiled_copy_at_first_sight...)
    monkeypatch.setattr(renderers, "YPAIR_FIRST_SIGHT_SAMPLES_PER_VOXEL", float("inf"))

    case, h, w, n_poses = _case(100 + seed, 8 if seed % 2 else 9)
    kw = [dict(), dict(voxel_shift=0.0, step_mode="n_minus_1"), dict(norm_dims_offset=-1), dict(near=0.15, far=0.95)][seed % 4]
    spec = RenderSpec(renderer="trilinear", n_points=int(np.random.default_rng(seed).integers(40, 110)), **kw)
    wgt = torch.rand(n_poses, 1, h * w, generator=torch.Generator().manual_seed(seed))
    vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
    with torch.no_grad():                                                            # first and second sight: natural layout
        first = render(vol, src, tgt, img, spec, ray_grid_w=w)
        render(vol, src, tgt, img, spec, ray_grid_w=w)
    for t in (vol, src, tgt, img):
        t.requires_grad_(True)
    renderers.PROFILER = []
    out = render(vol, src, tgt, img, spec, ray_grid_w=w)                              # third sight: the tiled y-pair copy
    (out * wgt.cuda()).sum().backward()
    names = [e[0] for e in renderers.PROFILER]
    renderers.PROFILER = None
    assert "pack_ypairs" in names and "trilinear_backward[vol]" in names, names
    assert torch.equal(first, out.detach())                                          # the copy changes no bit
    ref = _oracle64(case, spec, wgt)
    _close(out, ref[0], FWD_TOL, "out")
    _check_grads((vol.grad, src.grad, tgt.grad, img.grad), ref[1:], _oracle(case, spec, wgt, torch.float32)[1:])
