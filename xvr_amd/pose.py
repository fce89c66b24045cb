"""SE(3) pose algebra for the DRR render path (host side, plain torch, autograd-friendly).

Mirrors the surface xvr uses from ``diffdrr.pose`` (diffdrr==0.6.0, pinned in the reference's
``uv.lock:955-977``; the module itself is NOT in /root/reference, so names/semantics follow the
reference's call sites):

* ``convert(rot, xyz, parameterization=, convention=, degrees=)`` -> ``RigidTransform``
  (reference call sites: src/xvr/model/sampler.py:29-31, src/xvr/model/network.py:49-54,
  src/xvr/model/trainer.py:335-337)
* ``RigidTransform(matrix)``, ``.matrix``, ``.compose()``, ``.inverse()``, ``.convert()``, ``@``,
  ``[idx]``, ``len()``, ``.cuda()``  (src/xvr/model/trainer.py:193,204,210,268,275;
  src/xvr/model/loss.py:45-49; src/xvr/registrar/base.py:168,201)
* ``make_matrix(R, t)``  (src/xvr/utils/ants.py:43,82)

Convention: ``A.compose(B)`` applies ``A`` first and then ``B``, i.e. its
matrix is ``B.matrix @ A.matrix``; xvr relies on this in ``pose.compose(offset)``
(src/xvr/model/trainer.py:189-193).

This is B x 4 x 4 scalar math -- it is on the autograd chain of the renderer but is not a kernel.

Provenance: the rotation-conversion helpers (``euler_angles_to_matrix`` / ``matrix_to_euler_angles`` with ``_angle_from_tan``,
``quaternion_to_matrix`` / ``matrix_to_quaternion`` with ``_sqrt_positive_part``, ``axis_angle_to_matrix``,
``rotation_6d_to_matrix`` / ``matrix_to_rotation_6d``, the se(3) exp / log maps) keep the names, signatures and formulas of
PyTorch3D's ``pytorch3d.transforms`` (BSD 3-Clause, Copyright (c) Meta Platforms, Inc. and affiliates), which diffdrr.pose itself
re-exports and xvr's call sites therefore assume; they were written here from those public definitions (neither package is in
this tree) and the HIP ``xvr_pose_convert_*`` kernels evaluate the same formulas branch for branch.
"""

from __future__ import annotations

import torch

__all__ = [
    "RigidTransform",
    "convert",
    "make_matrix",
    "N_ANGULAR_COMPONENTS",
    "euler_angles_to_matrix",
    "matrix_to_euler_angles",
    "axis_angle_to_matrix",
    "matrix_to_axis_angle",
    "quaternion_to_matrix",
    "matrix_to_quaternion",
    "rotation_6d_to_matrix",
    "matrix_to_rotation_6d",
]

# Width of the rotational head of the pose regressor (src/xvr/model/network.py:4,28).
N_ANGULAR_COMPONENTS = {
    "axis_angle": 3,
    "euler_angles": 3,
    "se3_log_map": 3,
    "quaternion": 4,
    "rotation_6d": 6,
    "rotation_10d": 10,
    "quaternion_adjugate": 10,
}


# --------------------------------------------------------------------------------------
# Rotation parameterisations
# --------------------------------------------------------------------------------------
def _axis_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    c, s = torch.cos(angle), torch.sin(angle)
    one, zero = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == "X":
        flat = (one, zero, zero, zero, c, -s, zero, s, c)
    elif axis == "Y":
        flat = (c, zero, s, zero, one, zero, -s, zero, c)
    elif axis == "Z":
        flat = (c, -s, zero, s, c, zero, zero, zero, one)
    else:
        raise ValueError(f"invalid axis letter {axis!r}")
    return torch.stack(flat, dim=-1).reshape(angle.shape + (3, 3))


def _check_convention(convention: str) -> None:
    if not isinstance(convention, str) or len(convention) != 3:
        raise ValueError("convention must be a string of three letters from XYZ")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"invalid convention {convention!r}")
    for letter in convention:
        if letter not in "XYZ":
            raise ValueError(f"invalid letter {letter!r} in convention {convention!r}")


def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """Intrinsic Euler angles (radians) -> rotation matrices, R = R_c0(a0) @ R_c1(a1) @ R_c2(a2)."""
    if euler_angles.shape[-1] != 3:
        raise ValueError("euler angles must have a trailing dimension of 3")
    _check_convention(convention)
    mats = [_axis_rotation(c, a) for c, a in zip(convention, torch.unbind(euler_angles, -1))]
    return mats[0] @ mats[1] @ mats[2]


def _index_from_letter(letter: str) -> int:
    return "XYZ".index(letter)


def _angle_from_tan(axis: str, other_axis: str, data: torch.Tensor, horizontal: bool, tait_bryan: bool):
    i1, i2 = {"X": (2, 1), "Y": (0, 2), "Z": (1, 0)}[axis]
    if horizontal:
        i2, i1 = i1, i2
    even = (axis + other_axis) in ("XY", "YZ", "ZX")
    if horizontal == even:
        return torch.atan2(data[..., i1], data[..., i2])
    if tait_bryan:
        return torch.atan2(-data[..., i2], data[..., i1])
    return torch.atan2(data[..., i2], -data[..., i1])


def matrix_to_euler_angles(matrix: torch.Tensor, convention: str) -> torch.Tensor:
    """Rotation matrices -> intrinsic Euler angles (radians) in the given convention."""
    _check_convention(convention)
    i0 = _index_from_letter(convention[0])
    i2 = _index_from_letter(convention[2])
    tait_bryan = i0 != i2
    if tait_bryan:
        sign = -1.0 if (i0 - i2) in (-1, 2) else 1.0
        central = torch.asin(torch.clamp(matrix[..., i0, i2] * sign, -1.0, 1.0))
    else:
        central = torch.acos(torch.clamp(matrix[..., i0, i0], -1.0, 1.0))
    out = (
        _angle_from_tan(convention[0], convention[1], matrix[..., i2], False, tait_bryan),
        central,
        _angle_from_tan(convention[2], convention[1], matrix[..., i0, :], True, tait_bryan),
    )
    return torch.stack(out, dim=-1)


def _hat(v: torch.Tensor) -> torch.Tensor:
    x, y, z = v.unbind(-1)
    zero = torch.zeros_like(x)
    return torch.stack((zero, -z, y, z, zero, -x, -y, x, zero), dim=-1).reshape(v.shape[:-1] + (3, 3))


def _so3_coeffs(theta2: torch.Tensor):
    """sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3 with Taylor fallbacks near 0 (differentiable)."""
    small = theta2 < 1e-8
    safe = torch.where(small, torch.ones_like(theta2), theta2)
    theta = safe.sqrt()
    a = torch.where(small, 1 - theta2 / 6, torch.sin(theta) / theta)
    b = torch.where(small, 0.5 - theta2 / 24, (1 - torch.cos(theta)) / safe)
    c = torch.where(small, 1.0 / 6 - theta2 / 120, (theta - torch.sin(theta)) / (safe * theta))
    return a, b, c


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    """Rodrigues' formula (so(3) exponential map)."""
    theta2 = (axis_angle * axis_angle).sum(-1)
    a, b, _ = _so3_coeffs(theta2)
    K = _hat(axis_angle)
    eye = torch.eye(3, dtype=axis_angle.dtype, device=axis_angle.device).expand_as(K)
    return eye + a[..., None, None] * K + b[..., None, None] * (K @ K)


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """Real-first quaternions (not necessarily unit) -> rotation matrices."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """Rotation matrices -> unit quaternions (real part first, real part >= 0)."""
    batch = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
            dim=-1,
        )
    )
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(floor))
    best = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    out = candidates[best, :].reshape(batch + (4,))
    return torch.where(out[..., :1] < 0, -out, out)


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    q = matrix_to_quaternion(matrix)
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    sin_half_over_angle = torch.empty_like(angles)
    sin_half_over_angle[~small] = torch.sin(half[~small]) / angles[~small]
    sin_half_over_angle[small] = 0.5 - (angles[small] * angles[small]) / 48
    return q[..., 1:] / sin_half_over_angle


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt of two 3-vectors (Zhou et al. 2019); the vectors become the first two rows."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(matrix: torch.Tensor) -> torch.Tensor:
    return matrix[..., :2, :].clone().reshape(matrix.shape[:-2] + (6,))


def _vec10_to_sym4(vec: torch.Tensor) -> torch.Tensor:
    A = torch.zeros(vec.shape[:-1] + (4, 4), dtype=vec.dtype, device=vec.device)
    idx, jdx = torch.triu_indices(4, 4)
    A[..., idx, jdx] = vec
    A[..., jdx, idx] = vec
    return A


def rotation_10d_to_quaternion(rotation: torch.Tensor) -> torch.Tensor:
    """10-vector -> symmetric 4x4 -> eigenvector of the smallest eigenvalue (Peretroukhin et al. 2020)."""
    A = _vec10_to_sym4(rotation)
    return torch.linalg.eigh(A).eigenvectors[..., 0]


def quaternion_adjugate_to_quaternion(rotation: torch.Tensor) -> torch.Tensor:
    """10-vector = upper triangle of adj(A) ~ q q^T; the column of largest norm, normalised (Lin et al. 2023)."""
    A = _vec10_to_sym4(rotation)
    col_norms = A.norm(dim=-2)
    best = col_norms.argmax(dim=-1)
    gather = best[..., None, None].expand(A.shape[:-2] + (4, 1))
    col = torch.gather(A, -1, gather).squeeze(-1)
    return col / col_norms.amax(dim=-1, keepdim=True)


def quaternion_to_quaternion_adjugate(q: torch.Tensor) -> torch.Tensor:
    outer = q[..., :, None] * q[..., None, :]
    idx, jdx = torch.triu_indices(4, 4)
    return outer[..., idx, jdx]


def se3_exp_map(log_rot: torch.Tensor, log_trans: torch.Tensor):
    """se(3) twist (omega, v) -> (R, t) with t = V(omega) v."""
    theta2 = (log_rot * log_rot).sum(-1)
    a, b, c = _so3_coeffs(theta2)
    K = _hat(log_rot)
    K2 = K @ K
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device).expand_as(K)
    R = eye + a[..., None, None] * K + b[..., None, None] * K2
    V = eye + b[..., None, None] * K + c[..., None, None] * K2
    t = (V @ log_trans[..., None]).squeeze(-1)
    return R, t


def se3_log_map(R: torch.Tensor, t: torch.Tensor):
    omega = matrix_to_axis_angle(R)
    theta2 = (omega * omega).sum(-1)
    a, b, c = _so3_coeffs(theta2)
    K = _hat(omega)
    eye = torch.eye(3, dtype=R.dtype, device=R.device).expand_as(K)
    V = eye + b[..., None, None] * K + c[..., None, None] * (K @ K)
    v = torch.linalg.solve(V, t[..., None]).squeeze(-1)
    return omega, v


# --------------------------------------------------------------------------------------
# RigidTransform
# --------------------------------------------------------------------------------------
def make_matrix(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Stack [R | t; 0 0 0 1] (batched)."""
    if R.dim() == 2:
        R = R[None]
    if t.dim() == 1:
        t = t[None]
    top = torch.cat([R, t[..., None]], dim=-1)
    bottom = torch.zeros(R.shape[:-2] + (1, 4), dtype=R.dtype, device=R.device)
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], dim=-2)


class RigidTransform(torch.nn.Module):
    """A batch of 4x4 homogeneous transforms acting on column vectors: ``y = R x + t``.

    Despite the name (kept for drop-in compatibility) the matrix may be a general affine, e.g. the
    voxel<->world affine of a CT (src/xvr/model/trainer.py:266-275 wraps ``affine.inverse()`` in it).
    ``inverse()`` therefore uses the rigid closed form only when asked (``rigid=True``, the default,
    matching the reference class' use for poses); affines are inverted with ``torch.linalg.inv``.
    """

    def __init__(self, matrix: torch.Tensor):
        super().__init__()
        matrix = torch.as_tensor(matrix)
        if matrix.dim() == 2:
            matrix = matrix[None]
        if matrix.shape[-2:] != (4, 4):
            raise ValueError(f"expected [..., 4, 4] matrix, got {tuple(matrix.shape)}")
        # A plain attribute when it carries autograd history, a buffer otherwise, so that
        # .to()/.cuda()/deepcopy behave like the reference module.
        if matrix.requires_grad or matrix.grad_fn is not None:
            self.matrix = matrix
        else:
            self.register_buffer("matrix", matrix)

    # -- container protocol (src/xvr/model/trainer.py:204, src/xvr/model/loss.py:44-49) --
    def __len__(self) -> int:
        return len(self.matrix)

    def __getitem__(self, idx) -> "RigidTransform":
        m = self.matrix[idx]
        return RigidTransform(m if m.dim() == 3 else m[None])

    def __matmul__(self, other: "RigidTransform") -> "RigidTransform":
        return RigidTransform(self.matrix @ other.matrix)

    @property
    def rotation(self) -> torch.Tensor:
        return self.matrix[..., :3, :3]

    @property
    def translation(self) -> torch.Tensor:
        return self.matrix[..., :3, 3]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Apply to points ``x[B|1, n, 3]`` (batch broadcasts against the transform's batch)."""
        R, t = self.matrix[..., :3, :3], self.matrix[..., :3, 3]
        return torch.einsum("bij,bnj->bni", R, x) + t[:, None, :]

    def inverse(self, rigid: bool = True) -> "RigidTransform":
        if rigid:
            Rt = self.matrix[..., :3, :3].transpose(-1, -2)
            t = -(Rt @ self.matrix[..., :3, 3:]).squeeze(-1)
            return RigidTransform(make_matrix(Rt, t))
        return RigidTransform(torch.linalg.inv(self.matrix))

    def compose(self, other: "RigidTransform") -> "RigidTransform":
        """``self`` first, then ``other`` (matrix = other @ self)."""
        return RigidTransform(other.matrix @ self.matrix)

    def convert(self, parameterization: str, convention: str | None = None):
        """Inverse of :func:`convert`: returns ``(rotation_params, translation)``."""
        R, t = self.rotation, self.translation
        if parameterization == "se3_log_map":
            return se3_log_map(R, t)
        t = (R.transpose(-1, -2) @ t[..., None]).squeeze(-1)  # inverse of convert's x = R (x_cam + t)
        if parameterization == "axis_angle":
            return matrix_to_axis_angle(R), t
        if parameterization == "euler_angles":
            return matrix_to_euler_angles(R, convention), t
        if parameterization == "matrix":
            return R, t
        if parameterization == "quaternion":
            return matrix_to_quaternion(R), t
        if parameterization == "quaternion_adjugate":
            return quaternion_to_quaternion_adjugate(matrix_to_quaternion(R)), t
        if parameterization == "rotation_6d":
            return matrix_to_rotation_6d(R), t
        if parameterization == "rotation_10d":
            # q q^T's complement: any symmetric A whose smallest eigenvector is q; use I - q q^T.
            q = matrix_to_quaternion(R)
            A = torch.eye(4, dtype=q.dtype, device=q.device) - q[..., :, None] * q[..., None, :]
            idx, jdx = torch.triu_indices(4, 4)
            return A[..., idx, jdx], t
        raise ValueError(f"unknown parameterization {parameterization!r}")


# parameters -> 4x4 as ONE HIP launch (and one for the backward) for float32 CUDA batches: xvr_pose_convert_forward /
# _backward (include/xvr_pose.h) evaluate the formulas below in forward-mode dual numbers.  False: the torch formulation
# (the cross-check in tests/test_pose.py); "matrix" always takes it.  Round 5: "rotation_10d" too is a kernel -- a cyclic Jacobi
# eigen-decomposition of its 4 x 4 per pose, the eigenvector's derivative by perturbation theory -- so that every parameterisation
# of N_ANGULAR_COMPONENTS (/root/reference/src/xvr/model/network.py:4,28) runs the device-resident registration loop.
FUSED_CONVERT = True
_FUSED_KINDS = {"euler_angles": 0, "axis_angle": 1, "quaternion": 2, "quaternion_adjugate": 3, "rotation_6d": 4, "se3_log_map": 5,
                "rotation_10d": 6}


class _ConvertFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rotation, translation, kind, axes):
        import ctypes

        from . import _lib
        from .renderers import _ptr, _stream

        lib = _lib.load()
        B = rotation.shape[0]
        rot, xyz = rotation.contiguous(), translation.contiguous()
        matrix = torch.empty(B, 4, 4, device=rot.device, dtype=torch.float32)
        jac = torch.empty(lib.xvr_pose_convert_jacobian_floats(B), device=rot.device, dtype=torch.float32)
        _lib.check(lib.xvr_pose_convert_forward(_ptr(rot), _ptr(xyz), B, kind, (ctypes.c_int * 3)(*axes), _ptr(matrix), _ptr(jac),
                                                _stream()), "xvr_pose_convert_forward")
        ctx.save_for_backward(jac)
        ctx.cfg = (B, kind, rotation.shape[1])
        return matrix

    @staticmethod
    def backward(ctx, g_matrix):
        from . import _lib
        from .renderers import _ptr, _stream

        (jac,) = ctx.saved_tensors
        B, kind, k = ctx.cfg
        g = g_matrix.contiguous()
        g_rot = torch.empty(B, k, device=g.device, dtype=torch.float32)
        g_xyz = torch.empty(B, 3, device=g.device, dtype=torch.float32)
        _lib.check(_lib.load().xvr_pose_convert_backward(_ptr(jac), _ptr(g), B, kind, _ptr(g_rot), _ptr(g_xyz), _stream()),
                   "xvr_pose_convert_backward")
        return g_rot, g_xyz, None, None


def convert(*args, parameterization: str, convention: str | None = None, degrees: bool = False) -> RigidTransform:
    """Pose parameters -> :class:`RigidTransform` (camera-to-world; the source sits at ``R @ xyz``)."""
    if parameterization == "matrix" and len(args) == 1:
        return RigidTransform(args[0])
    rotation, translation = args
    if (FUSED_CONVERT and parameterization in _FUSED_KINDS and torch.is_tensor(rotation) and rotation.is_cuda and rotation.dim() == 2
            and rotation.dtype == torch.float32 and torch.is_tensor(translation) and translation.shape == (rotation.shape[0], 3)
            and translation.dtype == torch.float32 and translation.device == rotation.device and rotation.shape[0] > 0
            and rotation.shape[1] == N_ANGULAR_COMPONENTS[parameterization]):
        axes = (0, 1, 2)
        if parameterization == "euler_angles":
            _check_convention(convention)
            axes = tuple(_index_from_letter(c) for c in convention)
            if degrees:
                rotation = torch.deg2rad(rotation)
        return RigidTransform(_ConvertFused.apply(rotation, translation, _FUSED_KINDS[parameterization], axes))
    if parameterization == "axis_angle":
        R = axis_angle_to_matrix(rotation)
    elif parameterization == "euler_angles":
        if degrees:
            rotation = torch.deg2rad(rotation)
        R = euler_angles_to_matrix(rotation, convention)
    elif parameterization == "matrix":
        R = rotation
    elif parameterization == "quaternion":
        R = quaternion_to_matrix(rotation)
    elif parameterization == "quaternion_adjugate":
        R = quaternion_to_matrix(quaternion_adjugate_to_quaternion(rotation))
    elif parameterization == "rotation_6d":
        R = rotation_6d_to_matrix(rotation)
    elif parameterization == "rotation_10d":
        R = quaternion_to_matrix(rotation_10d_to_quaternion(rotation))
    elif parameterization == "se3_log_map":
        R, translation = se3_exp_map(rotation, translation)
    else:
        raise ValueError(
            f"parameterization must be one of {sorted(N_ANGULAR_COMPONENTS) + ['matrix']}, got {parameterization!r}"
        )
    if parameterization != "se3_log_map":
        # C-arm convention: translate in the camera frame, then rotate -- x_world = R (x_cam + t).
        # Pinned by the reference's call sites: a fixed xyz=(0, sid, 0) with varying gantry angles
        # must orbit the isocentre (src/xvr/io/xray.py:77-90) and adding pi to the yaw at fixed
        # xyz must give the antipodal view (src/xvr/model/inference.py:51-55).
        translation = (R @ translation[..., None]).squeeze(-1)
    return RigidTransform(make_matrix(R, translation))
