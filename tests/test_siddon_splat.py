"""k_siddon_splat (round 5): the Siddon voxel gradient as a ray-driven, brick-local fixed-point splat -- the default for the
non-exact index maps SURVEY.md Appendix A recalls (norm_dims_offset = +1, align_corners), an A/B for the exact one (option
siddon_splat = 2).  Against the atomic scatter (the merge walk's backward) and the oracle on tie-free sizes, against its OWN forward
(the slab march with the same plane alphas and index arithmetic) through the adjoint identity on sizes WITH the map's structural tie,
on several bricks, ragged sizes, a source inside the volume, more poses than one pass of the kernel takes, zero / non-finite upstream gradients, and for the
guard band that replaces the silent wrap of an optimistic bound."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import has_structural_tie, make_case
from test_hip_parity import GRAD_TOL, _close, _hip_render, _oracle_render

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
NX = [dict(norm_dims_offset=1), dict(norm_dims_offset=1, voxel_shift=0.0), dict(align_corners=True),
      dict(norm_dims_offset=1, align_corners=True, voxel_shift=0.0)]
_ids = lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "exact"   # noqa: E731


def _voxel_grad(case, spec, w, grid_w, splat=1, gather=True):
    from xvr_amd import _lib, renderers

    renderers.VOXEL_GATHER = gather
    try:
        with _lib.option("siddon_splat", splat), _lib.option("fwd_split", 1):
            return _hip_render(case, spec, grid_w=grid_w, grads=True, w=w)
    finally:
        renderers.VOXEL_GATHER = True


def _differing(a, b, tol=1e-4):
    return int(((a - b).abs() > tol * b.abs().max()).sum())


@pytest.mark.parametrize("kw", NX + [dict()], ids=_ids)
@pytest.mark.parametrize("shape,hw,why", [
    ((41, 37, 45), (96, 80), "several bricks"),
    ((9, 7, 11), (24, 20), "a volume smaller than one brick"),
    ((33, 17, 49), (31, 57), "ragged: one voxel past a brick, one row past a wavefront"),
    ((17, 33, 21), (150, 140), "a fine detector: many rays per voxel, queues that wrap"),
])
def test_siddon_splat_equals_the_scatter_and_the_oracle(kw, shape, hw, why):
    from xvr_amd.spec import RenderSpec

    assert not any(has_structural_tie(S, **kw) for S in shape)
    spec = RenderSpec(renderer="siddon", **kw)
    case = make_case(seed=23, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw),
                     xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
    w = torch.randn(2, 1, hw[0] * hw[1], generator=torch.Generator().manual_seed(3))
    splat = _voxel_grad(case, spec, w, hw[1], splat=2 if not kw else 1)[1]
    scatter = _voxel_grad(case, spec, w, hw[1], gather=False)[1]
    assert splat.abs().max() > 0 and torch.isfinite(splat).all(), why
    # The splat's plane alphas are the slab march's, the scatter's the merge walk's, torch's its own: they differ in the last bit,
    # an index a x_mid + b computed from them by ~1e-5, so about that fraction of the lookups lands on the other side of a
    # threshold and moves a segment between two neighbouring voxels (measured: ~1e-3 of the RAYS have one).  None under the exact
    # map, where a segment's midpoint is half a cell from any threshold.
    allowed = 8 + int(2.5e-3 * 2 * hw[0] * hw[1]) if kw else 0
    assert _differing(splat, scatter) <= allowed, (why, _differing(splat, scatter), allowed)
    assert abs(splat.double().sum().item() - scatter.double().sum().item()) <= 1e-4 * scatter.double().abs().sum().item()
    # ... and every voxel that differs must BE such a move: +c here, -c in a neighbour, the 3 x 3 x 3 box sum conserved
    # (conftest.unpaired_moves) -- a wrong weight, a dropped or a doubled segment would not cancel
    from conftest import unpaired_moves
    import conftest
    conftest.UNPAIRED_DETAILS.clear()
    n_bad, n_unpaired = unpaired_moves(splat, scatter, 1e-4)
    assert n_unpaired <= max(1, n_bad // 20), (why, n_bad, n_unpaired, conftest.UNPAIRED_DETAILS)
    if hw[0] * hw[1] <= 10000:
        ref = _oracle_render(case, spec, grads=True, w=w)[1]
        assert _differing(splat.cpu(), ref, 2e-3) <= allowed, (why, _differing(splat.cpu(), ref, 2e-3), allowed)
        n_bad, n_unpaired = unpaired_moves(splat.cpu(), ref, 2e-3)
        assert n_unpaired <= max(1, n_bad // 20), (why, "vs oracle", n_bad, n_unpaired)


@pytest.mark.parametrize("kw", NX + [dict()], ids=_ids)
@pytest.mark.parametrize("shape", [(40, 36, 44), (24, 24, 24)], ids=["even", "cube-24"])
def test_forward_and_voxel_gradient_are_one_pair_even_where_the_map_has_a_tie(kw, shape):
    """<A v, w> = <v, A^T w>: the render is linear in the volume, so the voxel gradient of sum(w * out) dotted with the volume IS
    sum(w * out).  Under dims = shape + 1 an even-sized axis has a cell whose midpoint maps to exactly k + 1/2; forward (slab
    march) and backward (splat) evaluate plane alphas and indices with the same expressions, so they credit the same voxel --
    a pair that broke the tie differently would miss by the values of whole slabs (1e-2), not by rounding."""
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", **kw)
    if kw.get("norm_dims_offset"):
        assert any(has_structural_tie(S, **kw) for S in shape)
    case = make_case(seed=5, shape=shape, height=40, width=44, delx=1.1)
    case["volume"] = torch.rand(shape, generator=torch.Generator().manual_seed(8))     # (no smoothness to hide behind)
    w = torch.rand(2, 1, 40 * 44, generator=torch.Generator().manual_seed(9))
    for splat in ((1, 0) if kw else (1, 2)):
        out, gvol = _voxel_grad(case, spec, w, 44, splat=splat)[:2]
        lhs = (out.double() * w.cuda().double()).sum().item()
        rhs = (gvol.double() * case["volume"].cuda().double()).sum().item()
        if splat == 0 and kw:
            continue   # (the round-2 per-cell gather / the scatter carry the merge walk's alphas: with the march's forward not a pair on a tie)
        assert abs(lhs - rhs) <= 2e-5 * abs(lhs), (kw, splat, lhs, rhs)


@pytest.mark.parametrize("B", [37, 150], ids=["two cull words", "two passes over the poses (128 per pass)"])
def test_siddon_splat_many_poses_source_inside_and_determinism(B):
    from xvr_amd.spec import RenderSpec

    rng = np.random.default_rng(11)
    rot = tuple((float(rng.uniform(100, 260)), float(rng.uniform(-50, 50)), float(rng.uniform(-20, 20))) for _ in range(B))
    xyz = tuple((float(rng.uniform(-6, 6)), 4.0 if i % 9 == 0 else float(rng.uniform(120, 300)), float(rng.uniform(-6, 6))) for i in range(B))
    case = make_case(seed=2, shape=(35, 29, 41), height=26, width=30, rot=rot, xyz=xyz, delx=1.7)
    spec = RenderSpec(renderer="siddon", norm_dims_offset=1)
    w = torch.randn(B, 1, 26 * 30, generator=torch.Generator().manual_seed(1))
    a = _voxel_grad(case, spec, w, 30)[1]
    assert torch.equal(a, _voxel_grad(case, spec, w, 30)[1]), "integer sums: same bits whatever order the bricks and the rays were taken in"
    scatter = _voxel_grad(case, spec, w, 30, gather=False)[1]
    assert _differing(a, scatter) <= 8 + int(2.5e-3 * B * 26 * 30), _differing(a, scatter)     # (cross-family ties, see above)
    from conftest import unpaired_moves
    n_bad, n_unpaired = unpaired_moves(a, scatter, 1e-4)     # ... each of them a neighbour swap, not a wrong weight
    assert n_unpaired <= max(1, n_bad // 20), (n_bad, n_unpaired)


def test_siddon_splat_zero_and_non_finite_upstream_gradients():
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", norm_dims_offset=1)
    case = make_case(seed=37, shape=(37, 41, 35), height=40, width=44, delx=0.9)
    w = torch.randn(2, 1, 40 * 44, generator=torch.Generator().manual_seed(6))
    w0 = w.clone()
    w0[1] = 0.0
    a = _voxel_grad(case, spec, w0, 44)[1]
    sc0 = _voxel_grad(case, spec, w0, 44, gather=False)[1]
    assert _differing(a, sc0) <= 8 + int(2.5e-3 * 40 * 44)
    from conftest import unpaired_moves
    n_bad, n_unpaired = unpaired_moves(a, sc0, 1e-4)
    assert n_unpaired <= max(1, n_bad // 20), (n_bad, n_unpaired)
    wn = w.clone()
    wn[1, 0, 17] = float("nan")
    bad = _voxel_grad(case, spec, wn, 44)[1]
    assert not torch.isfinite(bad).all(), "a NaN upstream gradient must not disappear"
    # (the poisoned pose's footprint covers every brick of this small volume; where only the healthy pose reaches, values stay:
    #  a narrow second pose)
    narrow = make_case(seed=37, shape=(37, 41, 35), height=40, width=44, delx=0.15, xyz=((0.0, 300.0, 0.0), (9.0, 300.0, 8.0)))
    bad2 = _voxel_grad(narrow, spec, wn, 44)[1]
    assert not torch.isfinite(bad2).all() and torch.isfinite(bad2).any(), "bricks the poisoned pose does not reach keep their values"


_OVERFLOW_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from conftest import make_case
from xvr_amd import renderers
from xvr_amd.renderers import render
from xvr_amd.spec import RenderSpec
case = make_case(seed=37, shape=(37, 41, 35), height=40, width=44, delx=0.9)
for renderer, kw in (("siddon", dict(norm_dims_offset=1)), ("trilinear", dict(n_points=80))):
    vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
    vol.requires_grad_(True)
    out = render(vol, src, tgt, img, RenderSpec(renderer=renderer, **kw), ray_grid_w=44)
    out.sum().backward()          # (every upstream value 1: the sums are as large as the bound allows)
    torch.cuda.synchronize()
    print(renderer, "nan" if torch.isnan(vol.grad).any() else "finite", renderers.last_backward_overflowed())
"""


def test_an_optimistic_bound_is_seen_not_wrapped(tmp_path):
    """The fixed-point sums live in three quarters of the int32 range; a sum in the guard band (or wrapped through it) means the
    bound on a voxel's sum was optimistic.  A diagnostic build whose bound is 1/64 of the derived one (XVR_SPLAT_BOUND_SCALE) must
    poison the voxels it cannot represent (NaN, never a wrapped finite number) and raise the workspace's sticky word -- both
    splats; the product build on the same input stays finite with the word clear."""
    from xvr_amd.build import build_diagnostic_library, diagnostic_path

    lib = build_diagnostic_library("XVR_SPLAT_BOUND_SCALE=0.015625f", diagnostic_path("optimistic_bound"), only=["drr_gather.hip"])
    script = tmp_path / "overflow.py"
    script.write_text(_OVERFLOW_SCRIPT.format(root=str(ROOT)))
    for env_lib, want in ((str(lib), "nan True"), (None, "finite False")):
        env = dict(os.environ)
        if env_lib:
            env["XVR_DRR_LIBRARY"] = env_lib
        out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith(("siddon", "trilinear"))]
        assert lines == [f"siddon {want}", f"trilinear {want}"], (env_lib, out.stdout)


_FALLBACK_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from conftest import make_case
from xvr_amd import renderers
from xvr_amd.renderers import render
from xvr_amd.spec import RenderSpec
case = make_case(seed=37, shape=(37, 41, 35), height=40, width=44, delx=0.9)
vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
vol.requires_grad_(True)
w = torch.randn(2, 1, 40 * 44, generator=torch.Generator().manual_seed(6)).cuda()
out = render(vol, src, tgt, img, RenderSpec(renderer="siddon", norm_dims_offset=1), ray_grid_w=44)
(out * w.reshape(out.shape)).sum().backward()
torch.cuda.synchronize()
torch.save(vol.grad.cpu(), {out!r})
print("workspace", renderers._LAST_VOL_WORKSPACE.numel() * renderers._LAST_VOL_WORKSPACE.element_size())
"""


def test_a_detector_beyond_the_queue_entries_takes_the_per_cell_gather(tmp_path):
    """k_siddon_splat's queue entries hold (pose slot, pixel) in 32 bits: detectors beyond 2^25 pixels keep round 2's per-cell gather
    -- dispatch AND workspace size (the gather needs 32 B per voxel more).  Nobody allocates such a detector in a test: a
    diagnostic build with 10 pixel bits makes this 40 x 44 one too large."""
    from xvr_amd.build import build_diagnostic_library, diagnostic_path

    lib = build_diagnostic_library("XVR_SIDDON_SPLAT_PX_BITS=10", diagnostic_path("small_splat_detector"), only=["drr_gather.hip", "drr_siddon.hip"])
    grads, sizes = [], []
    for env_lib in (None, str(lib)):
        env = dict(os.environ)
        if env_lib:
            env["XVR_DRR_LIBRARY"] = env_lib
        script, out_pt = tmp_path / f"fallback{len(grads)}.py", tmp_path / f"g{len(grads)}.pt"
        script.write_text(_FALLBACK_SCRIPT.format(root=str(ROOT), out=str(out_pt)))
        out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        sizes.append(int([l for l in out.stdout.splitlines() if l.startswith("workspace")][0].split()[1]))
        grads.append(torch.load(out_pt))
    assert sizes[1] >= sizes[0] + 32 * 37 * 41 * 35, sizes     # (the per-cell scratch)
    assert torch.isfinite(grads[1]).all() and grads[1].abs().max() > 0
    assert _differing(grads[0], grads[1]) <= 8 + int(2.5e-3 * 2 * 40 * 44), _differing(grads[0], grads[1])    # (cross-family ties, see above)
