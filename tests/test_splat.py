"""The default voxel gradient of the trilinear renderer: the brick-local fixed-point splat (k_trilinear_splat_b16,
HISTORY.md section 4.1) against the fp32 table gather (option gather_splat = 0), the atomic scatter and the oracle -- on
the paths the small parity cases do not reach (list overflow and more steps than a list holds -> "safe mode", a pose with
an all-zero or a non-finite upstream gradient, volumes smaller than a brick) and bit-for-bit repeatability."""
import pytest
import torch

from conftest import accuracy_by_magnitude, format_accuracy_table, make_case, to_oracle_spec
from test_hip_parity import GRAD_TOL, _close, _hip_render, _oracle_render

pytestmark = pytest.mark.gpu


def _grad(case, spec, w, grid_w, monkeypatch, splat=True, gather=True):
    """splat=True: option gather_splat = 3, the splat whatever the sampling density (the default, 1, hands launches with more than
    ~48 samples of a pose per voxel to the fp32 table gather: test_default_picks_the_fp32_gather_where_the_fixed_point_floor_shows)"""
    from xvr_amd import _lib, renderers

    renderers.VOXEL_GATHER = gather
    try:
        with _lib.option("gather_splat", 3 if splat else 0):
            return _hip_render(case, spec, grid_w=grid_w, grads=True, w=w)[1]
    finally:
        renderers.VOXEL_GATHER = True


@pytest.mark.parametrize("shape,hw,n_points,why", [
    ((40, 36, 44), (96, 80), 90, "several bricks, ordinary visits"),
    ((20, 18, 22), (300, 280), 160, "a fine detector on a small volume: more runs per visit than the list holds"),
    ((18, 20, 16), (40, 36), 6000, "more steps across a brick than the list has slots"),
    ((9, 7, 11), (24, 20), 50, "a volume smaller than one brick"),
    ((33, 17, 49), (31, 57), 70, "ragged: one voxel / one row past a brick / a wavefront"),
])
def test_splat_equals_table_gather_scatter_and_oracle(shape, hw, n_points, why, monkeypatch):
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", n_points=n_points)
    case = make_case(seed=23, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw),
                     xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
    w = torch.randn(2, 1, hw[0] * hw[1], generator=torch.Generator().manual_seed(3))
    splat = _grad(case, spec, w, hw[1], monkeypatch)
    table = _grad(case, spec, w, hw[1], monkeypatch, splat=False)
    scatter = _grad(case, spec, w, hw[1], monkeypatch, gather=False)
    assert splat.abs().max() > 0, why
    # the table gather adds the same weights in fp32; the splat rounds every product to 2^-30 of the bound on a voxel's sum
    # (the fine detector puts ~1500 samples of a pose on every voxel: the bound, hence the LSB, is 30 x the benchmark's)
    _close(splat, table, 4e-5 if hw[0] >= 300 else 1e-5, f"splat vs table gather ({why})")
    _close(splat, scatter, 4e-5, f"splat vs scatter ({why})")
    if n_points <= 200 and hw[0] * hw[1] <= 10000:
        _close(splat, _oracle_render(case, spec, grads=True, w=w)[1], GRAD_TOL, f"splat vs oracle ({why})")


def test_splat_is_bit_reproducible(monkeypatch):
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", n_points=120)
    case = make_case(seed=29, shape=(48, 40, 36), height=64, width=72, delx=0.7)
    w = torch.randn(2, 1, 64 * 72, generator=torch.Generator().manual_seed(4))
    a = _grad(case, spec, w, 72, monkeypatch)
    for _ in range(3):
        assert torch.equal(a, _grad(case, spec, w, 72, monkeypatch))


def test_splat_is_as_close_to_the_f64_sum_as_the_fp32_gather(monkeypatch):
    """Fixed point costs nothing visible: against the oracle run in float64, the splat's error is of the size of the fp32
    table gather's (both dominated by the fp32 sample positions)."""
    from oracle.diffdrr_restated import render
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", n_points=150)
    case = make_case(seed=31, shape=(40, 44, 36), height=48, width=52, delx=0.9)
    w = torch.randn(2, 1, 48 * 52, generator=torch.Generator().manual_seed(5))
    vol = case["volume"].double().requires_grad_(True)
    out = render(vol, case["source"].double(), case["target"].double(), case["img"].double(), to_oracle_spec(spec), None)
    (out * w.double()).sum().backward()
    ref = vol.grad
    err = [((_grad(case, spec, w, 52, monkeypatch, splat=s).cpu().double() - ref).abs().max() / ref.abs().max()).item() for s in (True, False)]
    assert err[0] < 2e-5 and err[0] < 2.0 * err[1] + 1e-6, err


def fixed_point_accuracy_case(which, monkeypatch=None):
    """(f64 oracle, splat, fp32 table gather) voxel gradients of one of the accuracy cases; also used by
    tools/splat_accuracy.py to print the committed table."""
    from oracle.diffdrr_restated import render
    from xvr_amd import _lib, renderers
    from xvr_amd.spec import RenderSpec

    shape, hw, n_points = {"ordinary": ((40, 44, 36), (48, 52), 150), "fine-detector": ((20, 18, 22), (300, 280), 160)}[which]
    spec = RenderSpec(renderer="trilinear", n_points=n_points)
    case = make_case(seed=23, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw),
                     xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
    w = torch.randn(2, 1, hw[0] * hw[1], generator=torch.Generator().manual_seed(3))
    vol = case["volume"].double().requires_grad_(True)
    out = render(vol, case["source"].double(), case["target"].double(), case["img"].double(), to_oracle_spec(spec), None, chunk=8192)
    (out * w.double()).sum().backward()
    got = {}
    for name, flag in (("splat", 3), ("gather", 0), ("default", 1)):
        with _lib.option("gather_splat", flag):
            got[name] = _hip_render(case, spec, grid_w=hw[1], grads=True, w=w)[1]
    return vol.grad, got


def assert_fixed_point_floor(rows, what, factor=4.0):
    """The splat's fixed point is an ABSOLUTE error floor (every product is rounded to bound * 2^-31 per (pose, brick)):
    voxel by voxel, relative to the voxel's own gradient, it must stay within `factor` x the fp32 gather's error (whose error
    is the fp32 sample positions and sums) down to 1e-4 of the largest gradient, and not fall apart below that."""
    for r in rows:
        if r["voxels"] < 50 or "splat" not in r:
            continue
        s, g = r["splat"], r["gather"]
        if r["decade"] <= 3:
            assert s["p99"] <= factor * g["p99"] + 1e-6 and s["median"] <= factor * g["median"] + 1e-7, (what, r)
        elif r["decade"] <= 5:
            # + the floor: ~1e-7 of max|g| absolute, i.e. 10^(decade - 6) relative for the smallest voxels of the decade
            # (|g| down to 10^-(decade + 1) max|g|; measured against the decade's upper edge the row passed or failed with the
            #  rounding pattern of its 50-140 voxels)
            assert s["p99"] <= factor * g["p99"] + 10.0 ** (r["decade"] - 6), (what, r)


@pytest.mark.parametrize("which", ["ordinary", "fine-detector"])
def test_splat_relative_accuracy_by_gradient_magnitude(which, monkeypatch):
    """the only consumer of dL/dvoxel is a per-voxel optimiser, which normalises every voxel by its own
    magnitude -- so the splat's accuracy is stated per decade of |g| / max|g| against the float64 oracle, next to the fp32
    table gather's (include/xvr_drr.h documents the floor; tools/splat_accuracy.py prints the table)."""
    ref, got = fixed_point_accuracy_case(which, monkeypatch)
    rows = accuracy_by_magnitude(ref, got)
    print(format_accuracy_table(rows, ["splat", "gather"]))
    assert sum(r["voxels"] for r in rows[:4]) > 1000
    # "ordinary" (a few samples of a pose per voxel, as in the benchmark, with a SIGNED noise gradient upstream -- the sums
    # cancel, the bound on them cannot): within 4 x of the fp32 gather, the criterion of VERDICT r2.  "fine-detector" is the
    # documented worst case -- pixels 15 x finer than voxels put ~1500 samples of a pose on every voxel, the bound (hence the
    # LSB) is 30 x the benchmark's, and the floor shows: 4-7 x the gather's error in the top decades, 3e-4 (median) relative
    # at 1e-4 of the largest gradient.  Callers who need fp32 sums there set the option gather_splat = 0.
    # Round 5: the DEFAULT no longer runs the splat there -- k_gather_prep counts the samples per voxel on the device and the fp32
    # table gather takes such launches -- so the default is held to the ordinary 4 x everywhere; the forced splat (option
    # gather_splat = 3) keeps its documented floor.
    assert_fixed_point_floor(rows, which, factor=4.0 if which == "ordinary" else 150.0)
    drows = accuracy_by_magnitude(ref, {"splat": got["default"], "gather": got["gather"]})
    assert_fixed_point_floor(drows, which + " (default)", factor=4.0)
    assert torch.equal(got["default"], got["gather"] if which == "fine-detector" else got["splat"])
    if which == "fine-detector":
        top = rows[0]
        assert top["splat"]["median"] < 1e-5 and top["splat"]["p99"] < 1e-4, top
    # and directly against the gather (same fp32 weights: the difference IS the fixed-point rounding): absolute, in units
    # of the largest gradient
    d = (got["splat"] - got["gather"]).abs().max().item() / got["gather"].abs().max().item()
    assert d < (4e-5 if which == "fine-detector" else 2e-6), d


def test_splat_skips_a_pose_whose_upstream_gradient_is_zero_and_flags_a_non_finite_one(monkeypatch):
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", n_points=80)
    case = make_case(seed=37, shape=(36, 40, 34), height=40, width=44, delx=0.9)
    w = torch.randn(2, 1, 40 * 44, generator=torch.Generator().manual_seed(6))
    w0 = w.clone()
    w0[1] = 0.0
    both = _grad(case, spec, w0, 44, monkeypatch)
    table = _grad(case, spec, w0, 44, monkeypatch, splat=False)
    _close(both, table, 2e-5, "second pose has no upstream gradient")
    wn = w.clone()
    wn[1, 0, 17] = float("nan")
    bad = _grad(case, spec, wn, 44, monkeypatch)
    assert not torch.isfinite(bad).all(), "a NaN upstream gradient must not disappear"
    # voxels only the first pose touches keep their finite value
    only_first = (_grad(case, spec, torch.cat([w[:1], torch.zeros_like(w[1:])]), 44, monkeypatch) != 0) & torch.isfinite(bad)
    assert only_first.any()


@pytest.mark.parametrize("kw,masked", [
    (dict(n_points=70, clip_to_volume=True), False),
    (dict(n_points=45, clip_to_volume=True, near=0.1, far=0.95), False),
    (dict(n_points=60), True),
    (dict(n_points=50, norm_dims_offset=-1), True),
], ids=["clip", "clip-near-far", "mask-per-channel", "mask-per-channel-dims-1"])
@pytest.mark.parametrize("shape,hw", [((40, 36, 44), (48, 40)), ((9, 7, 11), (24, 20)), ((33, 17, 49), (31, 57))])
def test_ray_major_splat_equals_the_voxel_driven_gather_and_the_scatter(kw, masked, shape, hw, monkeypatch):
    """k_trilinear_splat_px (clip_to_volume, masks with a per-channel gradient) against k_trilinear_gather_px
    (option gather_splat = 0) and the atomic scatter."""
    from xvr_amd import _lib, renderers
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", **kw)
    case = make_case(seed=41, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw), n_labels=4,
                     xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
    C = 4 if masked else 1
    w = torch.randn(2, C, hw[0] * hw[1], generator=torch.Generator().manual_seed(8))
    mask = case["mask"] if masked else None
    out = []
    for splat, gather in ((True, True), (False, True), (False, False)):
        renderers.VOXEL_GATHER = gather
        try:
            with _lib.option("gather_splat", 1 if splat else 0):
                out.append(_hip_render(case, spec, mask=mask, grid_w=hw[1], grads=True, w=w)[1])
        finally:
            renderers.VOXEL_GATHER = True
    assert out[0].abs().max() > 0
    # (under clip the fixed-point bound is a count of rays x samples, 4-10 x looser than the lattice bound of the plain splat)
    _close(out[0], out[1], 4e-5 if kw.get("clip_to_volume") else 2e-5, "ray-major splat vs voxel-driven pixel-major gather")
    _close(out[0], out[2], 4e-5, "ray-major splat vs scatter")


@pytest.mark.parametrize("what,shape,hw,kw,case_kw,batch", [
    ("the default render through the ray-major kernel (option gather_splat = 2)", (40, 36, 44), (48, 40), dict(n_points=70), {}, 2),
    ("a source inside the volume: a brick's footprint is the whole detector", (40, 36, 44), (24, 20), dict(n_points=60, clip_to_volume=True),
     dict(sdd=120.0, xyz=((1.0, 12.0, -2.0), (-2.0, 8.0, 3.0))), 2),
    ("fewer rays than a wavefront, one and two samples per ray", (20, 24, 28), (3, 5), dict(n_points=2, clip_to_volume=True), {}, 2),
    ("one sample per ray", (20, 24, 28), (6, 5), dict(n_points=1, clip_to_volume=True), {}, 2),
    ("more poses than one word of the cull mask", (24, 20, 28), (16, 12), dict(n_points=40, clip_to_volume=True), {}, 37),
    ("more samples per ray than the list's 16-bit step field: the voxel-driven gather takes over", (12, 10, 14), (6, 4),
     dict(n_points=70000, clip_to_volume=True), {}, 2),
], ids=["default-render", "source-inside", "tiny-detector", "one-sample", "37-poses", "70000-samples"])
def test_ray_major_splat_edge_cases(what, shape, hw, kw, case_kw, batch, monkeypatch):
    """The balanced visit of k_trilinear_splat_px (wavefront-private run lists, sample-exact shares) at the corners of
    its bookkeeping, against the atomic scatter and, where it is another kernel, the default splat."""
    from xvr_amd import _lib, renderers
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", **kw)
    g = torch.Generator().manual_seed(5)
    ckw = dict(seed=43, shape=shape, height=hw[0], width=hw[1], delx=0.9 * max(shape) / max(hw))
    if batch > 2:   # poses around the default pair
        rot = torch.tensor([170.0, 10.0, 5.0]) + 25.0 * (torch.rand(batch, 3, generator=g) - 0.5)
        xyz = torch.tensor([2.0, 300.0, -1.0]) + torch.tensor([6.0, 80.0, 6.0]) * (torch.rand(batch, 3, generator=g) - 0.5)
        ckw.update(rot=tuple(map(tuple, rot.tolist())), xyz=tuple(map(tuple, xyz.tolist())))
    else:
        ckw.update(xyz=((2.0, 300.0, -1.0), (-1.5, 200.0, 3.0)))
    ckw.update(case_kw)
    case = make_case(**ckw)
    w = torch.randn(batch, 1, hw[0] * hw[1], generator=g)
    mode = 2 if not kw.get("clip_to_volume") else 1
    with _lib.option("gather_splat", mode):
        ray_major = _hip_render(case, spec, grid_w=hw[1], grads=True, w=w)[1]
    renderers.VOXEL_GATHER = False
    try:
        scatter = _hip_render(case, spec, grid_w=hw[1], grads=True, w=w)[1]
    finally:
        renderers.VOXEL_GATHER = True
    assert ray_major.abs().max() > 0, what
    # (70 000 samples per ray: both sides are fp32 sums of ~10^4 terms per voxel in different orders)
    _close(ray_major, scatter, 4e-5 if kw["n_points"] < 10000 else 5e-4, f"ray-major splat vs scatter: {what}")
    if mode == 2:
        with _lib.option("gather_splat", 1):
            _close(ray_major, _hip_render(case, spec, grid_w=hw[1], grads=True, w=w)[1], 2e-5, f"ray-major vs plane-major splat: {what}")
    if hw[0] * hw[1] * kw["n_points"] * batch <= 4_000_000:
        _close(ray_major, _oracle_render(case, spec, grads=True, w=w)[1], GRAD_TOL, f"ray-major splat vs oracle: {what}")


@pytest.mark.parametrize("what,case_kw,spec_kw", [
    ("the source inside a brick: the box reaches the source plane, the bound falls back to counting runs",
     dict(shape=(40, 36, 44), height=40, width=36, sdd=140.0, xyz=((1.0, 6.0, -2.0), (-3.0, 9.0, 4.0)), delx=1.2), dict(n_points=90)),
     ("a source just outside the volume, a magnifying detector: hundreds of samples of a pose on the nearest voxels",
     dict(shape=(33, 29, 31), height=64, width=60, sdd=90.0, xyz=((0.0, 24.0, 0.0), (2.0, 27.0, -1.0)), delx=0.6), dict(n_points=200)),
    ("a near-degenerate gap: near ~ far, every sample plane of a pose within a voxel of the next",
     dict(shape=(36, 40, 34), height=48, width=44, delx=0.9), dict(n_points=300, near=0.70, far=0.74)),
    ("two sample planes per ray, both inside the volume", dict(shape=(24, 20, 28), height=32, width=28, delx=1.1), dict(n_points=2, near=0.70, far=0.78)),
    ("all samples of a ray in ONE plane of voxels: a detector pixel pitch of 0.05 voxels",
     dict(shape=(20, 18, 22), height=128, width=120, delx=0.9 * 22 / 2400), dict(n_points=40)),
], ids=["source-inside", "source-at-the-face", "degenerate-gap", "two-steps", "pitch-0.05"])
def test_adversarial_geometries_stay_inside_the_fixed_point_range(what, case_kw, spec_kw):
    """geometries chosen to stress the BOUND on a voxel's sum that the splat's fixed-point scale is made from
    (every upstream value +1: the sums cannot cancel).  The forced splat (option gather_splat = 3) must neither poison a voxel nor
    raise the sticky word -- the bound holds -- and must equal the fp32 table gather; the default route (which hands dense
    samplings to that gather) likewise."""
    from xvr_amd import _lib, renderers
    from xvr_amd.spec import RenderSpec

    spec = RenderSpec(renderer="trilinear", **spec_kw)
    case = make_case(seed=41, **case_kw)
    hw = (case_kw["height"], case_kw["width"])
    w = torch.ones(2, 1, hw[0] * hw[1])
    res = {}
    for mode in (3, 1, 0):
        with _lib.option("gather_splat", mode):
            res[mode] = _hip_render(case, spec, grid_w=hw[1], grads=True, w=w)[1]
        assert torch.isfinite(res[mode]).all(), (what, mode)
        assert not renderers.last_backward_overflowed(), (what, mode)
    assert res[3].abs().max() > 0, what
    _close(res[3], res[0], 2e-5, f"forced splat vs fp32 gather: {what}")
    _close(res[1], res[0], 2e-5, f"default route vs fp32 gather: {what}")
