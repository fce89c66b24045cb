import math

import pytest
import torch

from xvr_amd.pose import (N_ANGULAR_COMPONENTS, RigidTransform, convert, euler_angles_to_matrix,
                          make_matrix, matrix_to_euler_angles)


def _random_pose(n=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(n, 3, generator=g) - 0.5) * 2.0
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * 400
    return convert(rot, xyz, parameterization="euler_angles", convention="ZXY")


def test_euler_zxy_is_rz_rx_ry():
    a = torch.tensor([[0.3, -0.2, 0.5]])
    R = euler_angles_to_matrix(a, "ZXY")[0]
    cz, sz, cx, sx, cy, sy = math.cos(.3), math.sin(.3), math.cos(-.2), math.sin(-.2), math.cos(.5), math.sin(.5)
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    Rx = torch.tensor([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    assert torch.allclose(R, Rz @ Rx @ Ry, atol=1e-6)
    assert torch.allclose(matrix_to_euler_angles(R[None], "ZXY"), a, atol=1e-6)


def test_convert_orbits_the_isocentre():
    """Fixed xyz = (0, sid, 0) with varying gantry angles must orbit: |source| constant and the
    optical axis through the origin (pins x_world = R (x_cam + t); xray.py:77-90, inference.py:51-55)."""
    sid = 700.0
    for yaw in (0.0, 45.0, 180.0, 270.0):
        T = convert(torch.tensor([[yaw, 10.0, 0.0]]), torch.tensor([[0.0, sid, 0.0]]),
                    parameterization="euler_angles", convention="ZXY", degrees=True)
        src = T(torch.zeros(1, 1, 3))[0, 0]
        far = T(torch.tensor([[[0.0, -2 * sid, 0.0]]]))[0, 0]  # a point down the camera's -y axis
        assert abs(src.norm().item() - sid) < 1e-3
        assert torch.allclose(src + far, torch.zeros(3), atol=1e-3)  # symmetric about the origin


@pytest.mark.parametrize("param", sorted(N_ANGULAR_COMPONENTS))
def test_convert_round_trip(param):
    T = _random_pose()
    conv = "ZXY" if param == "euler_angles" else None
    rot, xyz = T.convert(param, conv)
    assert rot.shape[-1] == N_ANGULAR_COMPONENTS[param]
    T2 = convert(rot, xyz, parameterization=param, convention=conv)
    assert torch.allclose(T2.matrix, T.matrix, atol=2e-4)


def test_compose_order_and_inverse():
    A, B = _random_pose(3, 1), _random_pose(3, 2)
    x = torch.randn(3, 7, 3)
    assert torch.allclose(A.compose(B)(x), B(A(x)), atol=1e-3)
    assert torch.allclose(A.compose(A.inverse()).matrix, torch.eye(4).expand(3, 4, 4), atol=1e-4)
    assert torch.allclose((A @ B).matrix, A.matrix @ B.matrix)
    assert len(A[torch.tensor([True, False, True])]) == 2 and len(A[0]) == 1


def test_make_matrix_and_grad_flow():
    rot = torch.zeros(2, 3, requires_grad=True)
    xyz = torch.tensor([[0.0, 500.0, 0.0], [1.0, 2.0, 3.0]], requires_grad=True)
    T = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    T(torch.ones(1, 4, 3)).sum().backward()
    assert rot.grad is not None and xyz.grad is not None and torch.isfinite(rot.grad).all()
    M = make_matrix(torch.eye(3)[None], torch.tensor([[1.0, 2.0, 3.0]]))
    assert torch.equal(M[0, :3, 3], torch.tensor([1.0, 2.0, 3.0])) and M[0, 3, 3] == 1


def test_quaternion_adjugate_head_is_differentiable():
    rot = torch.randn(4, 10, requires_grad=True)
    T = convert(rot, torch.zeros(4, 3), parameterization="quaternion_adjugate")
    R = T.matrix[:, :3, :3]
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3).expand(4, 3, 3), atol=1e-5)
    R.sum().backward()
    assert torch.isfinite(rot.grad).all()


def test_projection_helpers_invert_the_detector_geometry():
    """perspective_projection / inverse_projection (metrics/evaluator.py:19-25) against Detector.forward:
    the pixel a detector target projects to is its own (col, row), and back."""
    from xvr_amd.data import make_phantom, read
    from xvr_amd.drr import DRR

    vol, _ = make_phantom(16, seed=1)
    for rev, (H, W) in ((False, (6, 8)), (True, (7, 5))):
        drr = DRR(read(vol, orientation="PA"), 900.0, H, 2.0, width=W, dely=3.0, x0=5.0, y0=-4.0, reverse_x_axis=rev)
        pose = _random_pose(2, seed=4)
        source, target = drr.detector(pose, None)
        pix = drr.perspective_projection(pose, target)               # [B, n, 2] as (col, row)
        cols = torch.arange(W, dtype=torch.float32).repeat(H)
        rows = torch.arange(H, dtype=torch.float32).repeat_interleave(W)
        assert torch.allclose(pix[..., 0], cols.expand(2, -1), atol=1e-3)
        assert torch.allclose(pix[..., 1], rows.expand(2, -1), atol=1e-3)
        # points half way along the rays project to the same pixels; inverse_projection returns the targets
        mid = 0.5 * (source + target)
        assert torch.allclose(drr.perspective_projection(pose, mid), pix, atol=1e-3)
        assert torch.allclose(drr.inverse_projection(pose, pix), target, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("param,convention", [("euler_angles", "ZXY"), ("euler_angles", "XYZ"), ("euler_angles", "ZYZ"), ("axis_angle", None),
                                               ("quaternion", None), ("quaternion_adjugate", None), ("rotation_6d", None),
                                               ("se3_log_map", None), ("rotation_10d", None)])
def test_fused_convert_matches_the_torch_formulation_value_and_gradient(param, convention):
    """xvr_pose_convert_forward / _backward (one launch each, forward-mode Jacobians) against the torch formulas of this module,
    which the regressor's head goes through every training step (network.py:49-56): matrices and the gradients w.r.t. the
    rotation parameters and the translation, small-angle branch and non-unit quaternions included."""
    from xvr_amd import pose as P

    g = torch.Generator().manual_seed(7)
    B, k = 37, P.N_ANGULAR_COMPONENTS[param]
    rot = torch.randn(B, k, generator=g)
    if param in ("axis_angle", "se3_log_map"):
        rot[0] = 0.0                        # exactly the identity: the Taylor branch, finite gradients
        rot[1] = 1e-5 * rot[1]              # |omega|^2 < 1e-8
        rot[2] = 3.0 * rot[2] / rot[2].norm()
    if param == "quaternion_adjugate":      # q q^T of a random quaternion (any scale), mildly perturbed as a regressor's output would be
        q = torch.randn(B, 4, generator=g) * (0.5 + torch.rand(B, 1, generator=g))
        rot = P.quaternion_to_quaternion_adjugate(q) + 0.02 * torch.randn(B, 10, generator=g)
    if param == "rotation_10d":             # a symmetric 4 x 4 whose smallest eigenvector is a random quaternion, perturbed: I - q q^T + noise
        q = torch.nn.functional.normalize(torch.randn(B, 4, generator=g), dim=-1)
        A = (0.5 + torch.rand(B, 1, 1, generator=g)) * (torch.eye(4) - q[:, :, None] * q[:, None, :])
        idx, jdx = torch.triu_indices(4, 4)
        rot = A[:, idx, jdx] + 0.03 * torch.randn(B, 10, generator=g)
    xyz = torch.randn(B, 3, generator=g) * torch.tensor([50.0, 300.0, 50.0])
    w = torch.randn(B, 4, 4, generator=g)
    kw = dict(parameterization=param, convention=convention)
    res = []
    for fused in (True, False):
        P.FUSED_CONVERT = fused
        try:
            r, t = rot.clone().cuda().requires_grad_(), xyz.clone().cuda().requires_grad_()
            m = P.convert(r, t, **kw).matrix
            (m * w.cuda()).sum().backward()
            res.append((m.detach().cpu(), r.grad.cpu(), t.grad.cpu()))
        finally:
            P.FUSED_CONVERT = True
    (m1, gr1, gt1), (m0, gr0, gt0) = res
    assert torch.isfinite(m1).all() and torch.isfinite(gr1).all() and torch.isfinite(gt1).all()
    assert torch.allclose(m1, m0, rtol=2e-5, atol=2e-4), (m1 - m0).abs().max()        # (translations of hundreds of mm in float32)
    assert torch.allclose(gt1, gt0, rtol=1e-4, atol=1e-4), (gt1 - gt0).abs().max()
    scale = gr0.abs().max().clamp_min(1.0)
    assert (gr1 - gr0).abs().max() <= 2e-4 * scale, ((gr1 - gr0).abs().max(), scale)
    # the same map in float64 on the CPU: the fused kernel is as close to it as the float32 torch chain is
    P.FUSED_CONVERT = False
    try:
        m64 = P.convert(rot.double(), xyz.double(), **kw).matrix
    finally:
        P.FUSED_CONVERT = True
    assert (m1.double() - m64).abs().max() <= 2.0 * (m0.double() - m64).abs().max() + 1e-4
    if param == "euler_angles":   # degrees, as the sampler uses them (sampler.py:29-31)
        a = P.convert(torch.rad2deg(rot).cuda(), xyz.cuda(), parameterization=param, convention=convention, degrees=True).matrix
        assert torch.allclose(a.cpu(), m0, rtol=2e-5, atol=2e-4)
