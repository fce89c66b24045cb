"""Property-based checks (hypothesis) of the host-side pose algebra and of the sharding helpers -- the code on
either side of the render path that every loop goes through.  CPU only."""
import math

import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from xvr_amd.distributed import shard_bounds, shard_counts
from xvr_amd.pose import N_ANGULAR_COMPONENTS, RigidTransform, convert

angles = st.floats(-3.0, 3.0, allow_nan=False)
shifts = st.floats(-500.0, 500.0, allow_nan=False)
pose_params = st.tuples(angles, st.floats(-1.4, 1.4), angles, shifts, shifts, shifts)   # middle Euler angle off the gimbal lock


def _pose(p):
    return convert(torch.tensor([p[:3]], dtype=torch.float64), torch.tensor([p[3:]], dtype=torch.float64),
                   parameterization="euler_angles", convention="ZXY")


@settings(max_examples=60, deadline=None)
@given(pose_params)
def test_every_parameterisation_round_trips_to_the_same_matrix(p):
    pose = _pose(p)
    R = pose.matrix[0, :3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3, dtype=R.dtype), atol=1e-10) and abs(float(torch.linalg.det(R)) - 1) < 1e-10
    for name in list(N_ANGULAR_COMPONENTS) + ["matrix"]:
        conv = "ZXY" if name == "euler_angles" else None
        rot, xyz = pose.convert(name, conv)
        back = convert(rot, xyz, parameterization=name, convention=conv)
        assert torch.allclose(back.matrix, pose.matrix, atol=1e-6), name


@settings(max_examples=60, deadline=None)
@given(pose_params, pose_params)
def test_compose_inverse_and_point_action(pa, pb):
    A, B = _pose(pa), _pose(pb)
    x = torch.tensor([[[1.0, -2.0, 3.0], [40.0, 5.0, -6.0]]], dtype=torch.float64)
    # compose = first self, then other
    assert torch.allclose(A.compose(B)(x), B(A(x)), atol=1e-8)
    assert torch.allclose((B @ A)(x), B(A(x)), atol=1e-8)
    # rigid inverse
    assert torch.allclose(A.inverse()(A(x)), x, atol=1e-8)
    assert torch.allclose(A.compose(A.inverse()).matrix, torch.eye(4, dtype=torch.float64)[None], atol=1e-9)
    # the source (camera origin) sits at R t with t the camera-frame translation
    assert torch.allclose(A.matrix[0, :3, 3], A.matrix[0, :3, :3] @ torch.tensor(pa[3:], dtype=torch.float64), atol=1e-9)


@settings(max_examples=40, deadline=None)
@given(pose_params)
def test_double_geodesic_is_a_distance_like_function(p):
    from xvr_amd.metrics import DoubleGeodesicSE3

    geo = DoubleGeodesicSE3(1020.0)
    A = _pose(p)
    B = _pose((p[0] + 0.1, p[1], p[2] - 0.05, p[3] + 3.0, p[4], p[5] - 4.0))
    ang, trans, d = geo(A, B)
    ang2, trans2, d2 = geo(B, A)
    assert float(d) >= 0 and abs(float(d) - float(d2)) < 1e-6 and abs(float(ang) - float(ang2)) < 1e-6
    assert float(geo(A, A)[2]) < 2e-3                                  # sqrt(eps) at coincidence
    assert abs(float(d) ** 2 - (float(ang) ** 2 + float(trans) ** 2)) < 1e-3
    # angular part = sdd / 2 x rotation angle of A^-1 B
    R = A.matrix[0, :3, :3].T @ B.matrix[0, :3, :3]
    angle = math.acos(max(-1.0, min(1.0, (float(torch.trace(R)) - 1) / 2)))
    assert abs(float(ang) - 510.0 * angle) < 1e-4


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 64))
def test_shard_bounds_partition_any_batch(n, world):
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))           # contiguous, no gap, no overlap
    sizes = [hi - lo for lo, hi in bounds]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1                 # balanced to within one pose
    assert list(shard_counts(n, world)) == sizes


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 6), st.integers(2, 9), st.integers(2, 9))
def test_detector_rays_have_the_stated_geometry(B, H, W):
    """Every target lies on the detector plane sdd away from the source along the viewing axis, and the pixel
    pitch is dely between rows and delx between columns, for any pose."""
    from xvr_amd.detector import Detector, make_reorient

    sdd, delx, dely = 1020.0, 1.5, 2.0
    det = Detector(sdd, H, W, delx, dely, 0.0, 0.0, reorient=make_reorient("AP"), reverse_x_axis=True)
    g = torch.Generator().manual_seed(B * 100 + H * 10 + W)
    pose = convert((torch.rand(B, 3, generator=g) - 0.5) * 2, (torch.rand(B, 3, generator=g) - 0.5) * 100 + torch.tensor([0.0, 800.0, 0.0]),
                   parameterization="euler_angles", convention="ZXY")
    source, target = det(pose, None)
    assert source.shape == (B, 1, 3) and target.shape == (B, H * W, 3)
    t = target.reshape(B, H, W, 3)
    centre = t.mean(dim=(1, 2))
    axis = centre - source[:, 0]
    assert torch.allclose(axis.norm(dim=1), torch.full((B,), sdd), atol=1e-2)
    # in-plane: every (target - centre) is orthogonal to the viewing axis
    off = t - centre[:, None, None]
    assert float((off * axis[:, None, None]).sum(-1).abs().max()) < 1e-1
    # rows are dely apart, columns delx (the calibration is diag(dely, delx, sdd): SURVEY.md Appendix A)
    assert torch.allclose((t[:, 1:, :] - t[:, :-1, :]).norm(dim=-1), torch.full((B, H - 1, W), dely), atol=1e-3)
    assert torch.allclose((t[:, :, 1:] - t[:, :, :-1]).norm(dim=-1), torch.full((B, H, W - 1), delx), atol=1e-3)
