"""The dominant-axis slab march of the Siddon forward (k_siddon_slab, round 4) against the merge walk it replaces (option
siddon_slab = 0), the torch oracle and analytic answers -- incl. the tie-breaking cases a fixed three-segment body must get
right: rays exactly along voxel planes, exactly through voxel corners (45 degrees), a source inside the volume, rays that miss."""
import math

import numpy as np
import pytest
import torch

from conftest import make_case, tie_free_shape, to_oracle_spec

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-4, 2e-3


def _close(a, b, tol, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol:.1e} (scale {scale:.3e})"


def _render(case, spec, slab, w=None, grid_w=0, count=False):
    """The UNSPLIT forward (+ jacobian-borne pose gradients) through the slab march (slab = 1) or the merge walk (0)."""
    from xvr_amd import _lib
    from xvr_amd.renderers import render

    vol, src, tgt, img = (case[k].cuda() for k in ("volume", "source", "target", "img"))
    work = torch.zeros(1, dtype=torch.int64, device="cuda") if count else None
    if w is not None:
        for t in (src, tgt, img):
            t.requires_grad_(True)
    with _lib.option("fwd_split", 1), _lib.option("siddon_slab", slab):
        out = render(vol, src, tgt, img, spec, None, ray_grid_w=grid_w, **({"work": work} if count else {}))
        if w is not None:
            (out * w.cuda()).sum().backward()
    if count:
        return out, int(work.item())
    return (out, src.grad, tgt.grad, img.grad) if w is not None else out


def _close_but(a, b, tol, allowed, what=""):
    """_close for all but ``allowed`` elements (non-exact index maps: a segment whose midpoint sits within an ulp of the lookup
    threshold goes to one voxel or its neighbour depending on the last bit of alpha -- a whole segment's value on that ray)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(b.abs().max().item(), 1e-12)
    bad = int(((a - b).abs() > tol * scale).sum())
    assert bad <= allowed, f"{what}: {bad} elements beyond {tol:.1e} (allowed {allowed})"


@pytest.mark.parametrize("kw", [dict(norm_dims_offset=1), dict(norm_dims_offset=1, voxel_shift=0.0), dict(align_corners=True),
                                dict(norm_dims_offset=1, align_corners=True, voxel_shift=0.0)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
@pytest.mark.parametrize("seed", range(4))
def test_slab_march_serves_the_non_exact_index_maps(seed, kw):
    """Round 5: norm_dims_offset = +1 (what SURVEY.md Appendix A recalls for upstream's Siddon) and align_corners on the slab
    march (k_siddon_slab<.., NX>): the same planes, every segment's voxel from its midpoint.  Against the merge walk's EXACT =
    false branch and the oracle, forward and pose gradients; a map that looks up voxels OUTSIDE the volume (norm_dims_offset = -1)
    keeps the merge walk, and the work counters agree."""
    from oracle.diffdrr_restated import render as oracle_render
    from xvr_amd.spec import RenderSpec

    rng = np.random.default_rng(300 + seed)
    # (sizes without a structural tie of the map, conftest.has_structural_tie: the merge walk, the march and torch evaluate the
    #  plane alphas with different roundings, and a midpoint that maps to EXACTLY k + 1/2 goes either way)
    shape = tie_free_shape(rng, 17, 40, **kw)
    rot = tuple((float(rng.uniform(135, 225)), float(rng.uniform(-45, 45)), float(rng.uniform(-15, 15))) for _ in range(2))
    xyz = tuple((float(rng.uniform(-8, 8)), float(rng.uniform(150, 320)), float(rng.uniform(-8, 8))) for _ in range(2))
    case = make_case(shape=shape, height=18, width=26, seed=seed, rot=rot, xyz=xyz, delx=float(rng.uniform(2.0, 7.0)))
    spec = RenderSpec(renderer="siddon", **kw)
    w = torch.rand(2, 1, 18 * 26, generator=torch.Generator().manual_seed(seed))
    new = _render(case, spec, 1, w, grid_w=26)
    old = _render(case, spec, 0, w, grid_w=26)
    vol, src, tgt, img = (case[k].clone() for k in ("volume", "source", "target", "img"))
    for t in (src, tgt, img):
        t.requires_grad_(True)
    ref = oracle_render(vol, src, tgt, img, to_oracle_spec(spec), None)
    (ref * w).sum().backward()
    _close_but(new[0], ref, FWD_TOL, 2, "NX slab forward vs oracle")
    _close_but(new[0], old[0], 2e-5, 2, "NX slab forward vs merge walk")
    for n, o, r, name in zip(new[1:], old[1:], (src.grad, tgt.grad, img.grad), ("grad_source", "grad_target", "grad_img")):
        _close_but(n, r, GRAD_TOL, 4, f"{name} vs oracle")
        _close_but(n, o, GRAD_TOL, 4, f"{name} vs merge walk")
    _close_but(_render(case, spec, 1), new[0], 1e-6, 0, "linear vs tiled ray order")
    _, c_new = _render(case, spec, 1, count=True)
    _, c_old = _render(case, spec, 0, count=True)
    assert abs(c_new - c_old) <= 0.01 * c_old + 8, (c_new, c_old)
    # a map that leaves the volume is not the march's: same bits with the option on and off
    out_spec = RenderSpec(renderer="siddon", norm_dims_offset=-1)
    assert torch.equal(_render(case, out_spec, 1), _render(case, out_spec, 0))


def test_slab_march_non_exact_known_answers():
    """Uniform box under norm_dims_offset = +1: every lookup lands inside the volume, so the integral is still density x chord --
    axis-aligned, along planes, diagonal; and a hot voxel is seen through the SHIFTED map (the segment whose midpoint maps to it)."""
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", norm_dims_offset=1)
    D = (10, 12, 14)
    vol = torch.full(D, 0.75)
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]
        for cu, cv in ((3.3, 4.1), (2.5, 4.1), (2.5, 4.5), (-0.499, -0.499)):
            for sign in (1.0, -1.0):
                a, b = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
                a[u] = b[u] = cu
                a[v] = b[v] = cv
                a[axis], b[axis] = (-50.0, 150.0) if sign > 0 else (150.0, -50.0)
                got = _render(dict(volume=vol, **_rays(a, [b])), spec, 1).item()
                assert abs(got - 0.75 * D[axis]) < 2e-3, (axis, cu, cv, sign, got)
    lo, hi = -0.5, 12 - 0.5
    volc = torch.full((12, 12, 12), 0.5)
    rays = _rays([lo - 20.0, lo - 20.0 + 0.3, 5.3], [[hi + 20.0, hi + 20.0 + 0.3, 5.3]])
    got = _render(dict(volume=volc, **rays), spec, 1).item()
    L = rays["img"].item()
    assert abs(got - 0.5 * L * (12.0 - 0.3) / (12.0 + 40.0)) < 2e-3 * got
    # one hot voxel at (7, 5, 6): a ray along x at y = 5.2, z = 6.1 crosses unit cells [c, c + 1] of the plane coordinate p = x + 1/2;
    # a whole cell is credited to voxel floor(a (c + 1/2)) of its MIDPOINT, a = 10 / 11: cell 8 -> 7.7 -> voxel 7, cell 7 -> 6.8 ->
    # voxel 6.  So the hot voxel is seen through exactly one whole cell: 2.0 x 1 mm
    hot = torch.zeros(10, 12, 14)
    hot[7, 5, 6] = 2.0
    rays = _rays([-40.0, 5.2, 6.1], [[60.0, 5.2, 6.1]])
    got = _render(dict(volume=hot, **rays), spec, 1).item()
    old = _render(dict(volume=hot, **rays), spec, 0).item()
    assert abs(got - 2.0) < 1e-4 and abs(old - 2.0) < 1e-4, (got, old)


@pytest.mark.parametrize("shift", [0.5, 0.0])
@pytest.mark.parametrize("seed", range(6))
def test_slab_march_matches_merge_walk_and_oracle(seed, shift):
    from oracle.diffdrr_restated import render as oracle_render
    from xvr_amd.spec import RenderSpec

    rng = np.random.default_rng(100 + seed)
    shape = tuple(int(x) for x in rng.integers(17, 40, size=3))
    rot = tuple((float(rng.uniform(135, 225)), float(rng.uniform(-45, 45)), float(rng.uniform(-15, 15))) for _ in range(2))
    xyz = tuple((float(rng.uniform(-8, 8)), float(rng.uniform(150, 320)), float(rng.uniform(-8, 8))) for _ in range(2))
    case = make_case(shape=shape, height=18, width=26, seed=seed, rot=rot, xyz=xyz, delx=float(rng.uniform(2.0, 7.0)))
    spec = RenderSpec(renderer="siddon", voxel_shift=shift)
    w = torch.rand(2, 1, 18 * 26, generator=torch.Generator().manual_seed(seed))
    new = _render(case, spec, 1, w, grid_w=26)
    old = _render(case, spec, 0, w, grid_w=26)
    vol, src, tgt, img = (case[k].clone() for k in ("volume", "source", "target", "img"))
    for t in (src, tgt, img):
        t.requires_grad_(True)
    ref = oracle_render(vol, src, tgt, img, to_oracle_spec(spec), None)
    (ref * w).sum().backward()
    _close(new[0], ref, FWD_TOL, "slab forward vs oracle")
    _close(new[0], old[0], 2e-5, "slab forward vs merge walk")
    for n, o, r, name in zip(new[1:], old[1:], (src.grad, tgt.grad, img.grad), ("grad_source", "grad_target", "grad_img")):
        _close(n, r, GRAD_TOL, f"{name} vs oracle")
        _close(n, o, GRAD_TOL, f"{name} vs merge walk")
    # linear ray order (no detector lattice) takes the same kernel
    _close(_render(case, spec, 1), new[0], 1e-6, "linear vs tiled ray order")
    # the kernels' own segment counts agree (ties aside)
    _, c_new = _render(case, spec, 1, count=True)
    _, c_old = _render(case, spec, 0, count=True)
    assert abs(c_new - c_old) <= 0.01 * c_old + 8, (c_new, c_old)


def _rays(src, tgts):
    s = torch.tensor([[src]], dtype=torch.float32)
    t = torch.tensor([tgts], dtype=torch.float32)
    return dict(source=s, target=t, img=(t - s).norm(dim=-1).unsqueeze(1))


@pytest.mark.parametrize("shift", [0.5, 0.0])
def test_slab_march_known_answers_axis_aligned_and_on_planes(shift):
    """Uniform box: the integral is density x chord, whatever plane or corner the ray runs along.  Rays exactly ALONG voxel planes
    (both minor coordinates on plane positions) and through the volume's corner columns are the ties of a Siddon walk."""
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", voxel_shift=shift)
    D = (10, 12, 14)
    vol = torch.full(D, 0.75)
    p0 = -shift                     # plane p of an axis sits at p + plane0, plane0 = -shift
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]
        tg = []
        for cu, cv in ((3.3, 4.1), (3.0 + p0, 4.1), (3.0 + p0, 5.0 + p0), (0.0 + p0 + 1e-3, 0.0 + p0 + 1e-3), (D[u] + p0 - 1e-3, 2.5)):
            s = [0.0, 0.0, 0.0]
            s[u], s[v] = cu, cv
            tg.append(s)
        for sign in (1.0, -1.0):
            outs = []
            for s in tg:
                a, b = list(s), list(s)
                a[axis], b[axis] = (-50.0, 150.0) if sign > 0 else (150.0, -50.0)
                case = dict(volume=vol, **_rays(a, [b]))
                outs.append(_render(case, spec, 1).item())
            assert all(abs(o - 0.75 * D[axis]) < 2e-3 for o in outs), (shift, axis, sign, outs)


@pytest.mark.parametrize("shift", [0.5, 0.0])
def test_slab_march_axis_aligned_rays_with_eps_zero(shift):
    """RenderSpec.eps = 0 is allowed, and an axis-aligned ray then has direction components of exactly (+-)0: 1 / d = inf made
    the far-plane bound -inf and the ray rendered as 0.  Uniform box => density x chord, as the merge walk gives."""
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", voxel_shift=shift, eps=0.0)
    D = (10, 12, 14)
    vol = torch.full(D, 0.75)
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]
        for cu, cv in ((3.3, 4.1), (0.2, 7.6), (D[u] - 1.4, 0.1)):
            for sign in (1.0, -1.0):
                a, b = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
                a[u] = b[u] = cu
                a[v] = b[v] = cv
                a[axis], b[axis] = (-50.0, 150.0) if sign > 0 else (150.0, -50.0)
                case = dict(volume=vol, **_rays(a, [b]))
                new, old = _render(case, spec, 1).item(), _render(case, spec, 0).item()
                assert abs(new - 0.75 * D[axis]) < 2e-3 and abs(old - new) < 1e-4, (shift, axis, sign, cu, cv, new, old)
    # one zero component only (a ray in a coordinate plane, oblique in the other two): against the oracle
    from oracle.diffdrr_restated import render as oracle_render

    volr = torch.rand(D, generator=torch.Generator().manual_seed(3))
    rays = _rays([-30.0, 4.25, -20.0], [[45.0, 4.25, 31.0], [40.0, 4.25, 2.0]])
    out = _render(dict(volume=volr, **rays), spec, 1)
    ref = oracle_render(volr, rays["source"], rays["target"], rays["img"], to_oracle_spec(spec), None)
    _close(out, ref, FWD_TOL, "eps = 0, one zero direction component")


@pytest.mark.parametrize("shift", [0.5, 0.0])
def test_slab_march_known_answers_diagonals(shift):
    """45-degree rays: |d_u| = |d_m| exactly, every slab has a minor crossing, and a ray aimed through voxel corners crosses two
    planes at the same alpha in every slab.  Uniform box => density x chord (L x fraction of alpha inside)."""
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", voxel_shift=shift)
    D = 12
    vol = torch.full((D, D, D), 0.5)
    lo, hi = -shift, D - shift               # the box in index coordinates
    cases = []
    for off in (0.0, 0.3, 0.5):              # through the corners / off the corners
        cases.append(([lo - 20.0, lo - 20.0 + off, 5.3], [hi + 20.0, hi + 20.0 + off, 5.3]))          # x-y diagonal
        cases.append(([5.3, hi + 20.0 + off, lo - 20.0], [5.3, lo - 20.0 + off, hi + 20.0]))          # y-z anti-diagonal
        cases.append(([lo - 20.0, lo - 20.0 + off, lo - 20.0 + off], [hi + 20.0, hi + 20.0 + off, hi + 20.0 + off]))   # space diagonal
    for s, t in cases:
        case = dict(volume=vol, **_rays(s, [t]))
        got = _render(case, spec, 1).item()
        d = np.array(t, dtype=np.float64) - np.array(s, dtype=np.float64)
        a0 = np.where(d > 0, (lo - np.array(s)) / d, (hi - np.array(s)) / d)
        a1 = np.where(d > 0, (hi - np.array(s)) / d, (lo - np.array(s)) / d)
        mask = np.abs(d) > 1e-9
        span = max(0.0, min(a1[mask].min(), 1.0) - max(a0[mask].max(), 0.0))
        want = 0.5 * span * np.linalg.norm(d)
        assert abs(got - want) < 2e-3 * max(want, 1.0), (shift, s, t, got, want)
        old = _render(case, spec, 0).item()
        assert abs(got - old) < 2e-3 * max(want, 1.0), (shift, s, t, got, old)


def test_slab_march_source_inside_miss_and_hot_voxel():
    from oracle.diffdrr_restated import render as oracle_render
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", voxel_shift=0.0)
    vol = torch.full((10, 10, 10), 0.5)
    # source inside the volume: integration clamped to alpha >= 0; a target inside as well; a ray that misses
    case = dict(volume=vol, **_rays([2.25, 5.0, 5.0], [[30.0, 5.0, 5.0], [7.5, 6.0, 4.0], [2.25, 5.0, 40.0]]))
    out = _render(case, spec, 1)[0, 0]
    assert abs(out[0].item() - 0.5 * (10 - 2.25)) < 1e-4
    assert abs(out[1].item() - 0.5 * math.dist([2.25, 5.0, 5.0], [7.5, 6.0, 4.0])) < 1e-4
    case = dict(volume=vol, **_rays([-20.0, 30.0, 5.0], [[40.0, 31.0, 5.0], [-20.0, 5.0, 5.0]]))
    assert float(_render(case, spec, 1).abs().max()) == 0.0
    # one hot voxel: exactly the path length inside it, against the oracle for oblique rays
    spec = RenderSpec(renderer="siddon", voxel_shift=0.5)
    vol = torch.zeros(9, 9, 9)
    vol[4, 5, 3] = 2.0
    case = dict(volume=vol, **_rays([-50.0, 5.2, 2.9], [[150.0, 5.2, 2.9], [150.0, 5.2, 3.6], [150.0, 9.0, 2.0], [150.0, 1.0, 4.0]]))
    got = _render(case, spec, 1)
    assert abs(got[0, 0, 0].item() - 2.0) < 1e-4
    ref = oracle_render(case["volume"], case["source"], case["target"], case["img"], to_oracle_spec(spec), None)
    _close(got, ref, FWD_TOL, "hot voxel")


@pytest.mark.parametrize("kw", [dict(), dict(voxel_shift=0.0), dict(voxel_shift=0.0, align_corners=True), dict(norm_dims_offset=1)],
                         ids=["exact", "exact, corner convention", "align_corners", "dims = shape + 1"])
def test_slab_march_rays_with_the_volume_behind_them_render_as_zero(kw):
    """A source beside a corner of a flat volume and a detector that looks past it: for some rays the volume lies BEHIND the ray --
    every far plane at a negative alpha, a_hi < a_lo = 0.  Round 5's soak (seed 70034, this geometry) found the march clamping its
    plane alphas into that reversed interval and crediting a positive "segment" between the two minor planes to whatever voxel the
    clamped index named: six pixels at 7-31 where the oracle has 0.  The exact map's small launches take the merge walk and never
    showed it; this test forces the march for every map.  Rays of both kinds share wavefronts."""
    from oracle.diffdrr_restated import render as oracle_render
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="siddon", **kw)
    g = torch.Generator().manual_seed(5)
    vol = torch.rand((22, 11, 6), generator=g) + 0.1
    src = torch.tensor([9.4953, -2.8751, -17.4703])
    i, j = torch.meshgrid(torch.arange(30.0), torch.arange(24.0), indexing="ij")
    d = (torch.tensor([19.384, -1.283, -35.511]) + (i - 8)[..., None] * torch.tensor([-0.64815, 1.16294, -1.11767])
         + j[..., None] * torch.tensor([-0.2992, 0.21322, 4.21897])).reshape(-1, 3)
    case = dict(volume=vol, **_rays(src.tolist(), (src + d).tolist()))
    ref = oracle_render(case["volume"], case["source"], case["target"], case["img"], to_oracle_spec(spec), None)
    assert (ref > 0).float().mean().item() > 0.05 and (ref == 0).float().mean().item() > 0.2      # both kinds of rays
    w = torch.rand(1, 1, d.shape[0], generator=g)
    got, walk = _render(case, spec, 1, w=w, grid_w=24), _render(case, spec, 0, w=w, grid_w=24)
    missed = (ref == 0) & (walk[0].cpu() == 0)
    assert bool((got[0].cpu()[missed] == 0).all()), f"{int((got[0].cpu()[missed] != 0).sum())} rays that miss the volume render as non-zero"
    if not kw.get("norm_dims_offset"):     # (dims + 1 on even sizes: the map's structural tie, compared within the pair elsewhere)
        _close_but(got[0], ref, FWD_TOL, 4 if kw.get("align_corners") else 0, "image")
        _close_but(got[2], walk[2], 5 * GRAD_TOL, 8, "d/d target, march against merge walk")


def test_slab_march_is_the_default_for_large_one_channel_launches_and_deterministic():
    """A launch of >= 2048 wavefronts with the exact index map takes the slab march by itself (no option), on the bricked copy
    (built at first sight); two runs give the same bits; image and pose gradient of DRR.forward agree with the merge walk's."""
    from xvr_amd import renderers
    from xvr_amd.data import make_phantom, read
    from xvr_amd.drr import DRR
    from xvr_amd.pose import convert

    vol, _ = make_phantom(64, n_ellipsoids=10, seed=3, device="cuda")
    drr = DRR(read(vol, spacing=(2.0,) * 3, orientation="AP"), 1020.0, 128, 2.2, renderer="siddon", reverse_x_axis=False).cuda()
    B = 10                                                              # 10 x 256 wavefronts
    g = torch.Generator().manual_seed(1)
    rot = (torch.tensor([[3.14, 0.0, 0.0]]) + (torch.rand(B, 3, generator=g) - 0.5) * 0.8).cuda().requires_grad_()
    xyz = (torch.tensor([[0.0, 700.0, 0.0]]) + (torch.rand(B, 3, generator=g) - 0.5) * 30).cuda().requires_grad_()
    renderers.PROFILER = []
    a = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    names = {e[0] for e in renderers.PROFILER}
    renderers.PROFILER = None
    assert "pack_bricks" in names and "siddon_forward+jac" in names, names
    b = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert torch.equal(a, b)
    w = torch.rand(a.shape, generator=g).cuda()
    (a * w).sum().backward()
    g_new = (rot.grad.clone(), xyz.grad.clone())
    # the same DRR.forward through the merge walk (on its bricked copy): the image and the pose gradient agree
    from xvr_amd import _lib
    rot.grad = xyz.grad = None
    with _lib.option("siddon_slab", 0):
        c = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        (c * w).sum().backward()
    _close(a, c, 2e-5, "DRR.forward: slab march vs merge walk")
    _close(g_new[0], rot.grad, GRAD_TOL, "d / d rot")
    _close(g_new[1], xyz.grad, GRAD_TOL, "d / d xyz")
