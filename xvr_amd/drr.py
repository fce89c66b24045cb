"""``DRR`` module: the drop-in for ``diffdrr.drr.DRR`` as xvr uses it.

Reference call sites honoured here:
* construction ``DRR(subject, sdd, height, delx, width, dely, x0, y0, reverse_x_axis=, renderer=,
  voxel_shift=...)``                       /root/reference/src/xvr/renderer/load.py:32-44,
                                           /root/reference/src/xvr/model/utils.py:154-171
* buffers ``density / mask / volume / center``, ``affine_inverse``, ``detector``, ``renderer``,
  ``reshape_transform``                    /root/reference/src/xvr/model/trainer.py:251-256,279-304
* ``drr(pose)``                            /root/reference/src/xvr/registrar/base.py:249,315,317
* ``set_intrinsics_`` / ``rescale_detector_``  /root/reference/src/xvr/registrar/base.py:155,212
* ``perspective_projection`` / ``inverse_projection``  /root/reference/src/xvr/metrics/evaluator.py:19-25
"""

from __future__ import annotations

import torch

from . import _lib
from .data import Subject
from .detector import Detector, make_reorient
from .pose import RigidTransform, convert
from .renderers import Siddon, Trilinear, _ptr, _stream, _timed, render_from_camera

__all__ = ["DRR", "rays_from_camera"]


class _RaysFromCamera(torch.autograd.Function):
    """cam [B,24] -> (source [B,1,3], target [B,H*W,3], raylen [B,1,H*W]) in one HIP launch; the
    backward reduces the three gradients to d/d cam [B,24] in one launch (xvr_drr_rays_*)."""

    @staticmethod
    def forward(ctx, cam, H, W):
        lib = _lib.load()
        cam_c = cam.contiguous()
        B, n = cam_c.shape[0], H * W
        source = torch.empty(B, 1, 3, device=cam.device, dtype=torch.float32)
        target = torch.empty(B, n, 3, device=cam.device, dtype=torch.float32)
        raylen = torch.empty(B, 1, n, device=cam.device, dtype=torch.float32)
        rc = _timed("rays_forward", lib.xvr_drr_rays_forward, _ptr(cam_c), B, H, W, _ptr(source), _ptr(target),
                    _ptr(raylen), _stream())
        _lib.check(rc, "xvr_drr_rays_forward")
        ctx.save_for_backward(cam_c)
        ctx.hw = (H, W)
        return source, target, raylen

    @staticmethod
    def backward(ctx, g_source, g_target, g_raylen):
        lib = _lib.load()
        (cam_c,) = ctx.saved_tensors
        H, W = ctx.hw
        B = cam_c.shape[0]
        g_cam = torch.zeros_like(cam_c)
        if g_target is None:
            g_target = torch.zeros(B, H * W, 3, device=cam_c.device, dtype=torch.float32)
        gs = g_source.contiguous() if g_source is not None else None
        gl = g_raylen.contiguous() if g_raylen is not None else None
        rc = _timed("rays_backward", lib.xvr_drr_rays_backward, _ptr(cam_c), B, H, W, _ptr(gs), _ptr(g_target.contiguous()),
                    _ptr(gl), _ptr(g_cam), _stream())
        _lib.check(rc, "xvr_drr_rays_backward")
        return g_cam, None, None


def rays_from_camera(cam: torch.Tensor, height: int, width: int):
    if not cam.is_cuda or cam.dtype != torch.float32:
        raise RuntimeError("rays_from_camera needs a float32 CUDA tensor (HIP kernel, no CPU path)")
    if cam.shape[0] == 0:
        z = cam.sum() * 0
        n = int(height) * int(width)
        return z.expand(0, 1, 3), z.expand(0, n, 3), z.expand(0, 1, n)
    return _RaysFromCamera.apply(cam, int(height), int(width))


class DRR(torch.nn.Module):
    def __init__(self, subject: Subject, sdd: float, height: int, delx: float, width: int | None = None,
                 dely: float | None = None, x0: float = 0.0, y0: float = 0.0, reshape: bool = True,
                 reverse_x_axis: bool = True, renderer: str = "siddon", voxel_shift: float = 0.5,
                 **renderer_kwargs):
        super().__init__()
        width = height if width is None else width
        dely = delx if dely is None else dely
        self.detector = Detector(sdd, height, width, delx, dely, x0, y0,
                                 reorient=make_reorient(subject.orientation), reverse_x_axis=reverse_x_axis)
        self.subject = subject
        affine = torch.as_tensor(subject.affine, dtype=torch.float32)
        self.register_buffer("_affine", affine[None].clone())
        self.register_buffer("_affine_inverse", torch.linalg.inv(affine)[None])
        density = subject.density if subject.density is not None else subject.volume
        self.register_buffer("density", density.to(torch.float32).contiguous())
        if subject.mask is not None:
            self.register_buffer("mask", subject.mask.to(torch.float32).contiguous())
        if renderer == "siddon":
            self.renderer = Siddon(voxel_shift=voxel_shift, **renderer_kwargs)
        elif renderer == "trilinear":
            self.renderer = Trilinear(voxel_shift=voxel_shift, **renderer_kwargs)
        else:
            raise ValueError(f"renderer must be 'siddon' or 'trilinear', got {renderer!r}")
        self.reshape = reshape
        # one fused HIP launch for detector -> ray length -> inverse affine instead of ~40 torch launches
        # (the explicit ``detector`` / ``affine_inverse`` calls of xvr's trainer keep working unchanged)
        self.fused_rays = True
        self.register_buffer("_e3", torch.tensor([0.0, 0.0, 1.0]), persistent=False)
        self._sync_ray_grid()

    def _sync_ray_grid(self):
        self.renderer.ray_grid = (self.detector.height, self.detector.width)
        self._cam_affine = None   # the pose -> camera constants belong to the detector just replaced

    @property
    def affine(self) -> RigidTransform:
        return RigidTransform(self._affine)

    @property
    def affine_inverse(self) -> RigidTransform:
        return RigidTransform(self._affine_inverse)

    def reshape_transform(self, img, batch_size):
        if self.reshape:
            # (channels spelled out: -1 is ambiguous for an empty batch)
            img = img.reshape(batch_size, img.shape[1], self.detector.height, self.detector.width)
        return img

    # ------------------------------------------------------------------ rendering
    def forward(self, *args, parameterization=None, convention=None, calibration=None,
                mask_to_channels=False, density=None, **kwargs):
        """``drr(pose)`` with a RigidTransform, or ``drr(rot, xyz, parameterization=, convention=)``.
        ``density`` overrides the module's buffer (e.g. a leaf tensor whose gradient is wanted)."""
        density = self.density if density is None else density
        if self.fused_rays and calibration is None and density.is_cuda:
            # pose -> camera through the affine map of camera_affine(): ONE HIP launch from Euler angles
            # (xvr_pose_camera_forward), one addmm from a 4x4 pose -- instead of the ~60 tiny torch launches of
            # convert() + camera(), which cost 0.5 ms per call and dominate a one-pose render
            G, c = self._camera_affine_cached()
            if (parameterization == "euler_angles" and len(args) == 2 and args[0].is_cuda and args[1].is_cuda
                    and args[0].dtype == args[1].dtype == torch.float32 and args[0].dim() == 2 and len(args[0]) > 0):
                from .pose_opt import pose_camera
                batch_size = len(args[0])
                cam = pose_camera(args[0], args[1], G, c, convention)
            else:
                pose = args[0] if parameterization is None else convert(
                    *args, parameterization=parameterization, convention=convention)
                batch_size = len(pose)
                cam = torch.addmm(c, pose.matrix[:, :3, :].reshape(batch_size, 12), G.T)
            spec_keys = {"n_points", "align_corners"} if self.renderer.renderer_name == "trilinear" else {"align_corners"}
            batch_window = self.renderer.spec_overrides.get("clip_to_volume") == "batch"   # (needs the rays in memory)
            if (not mask_to_channels and set(kwargs) <= spec_keys and not density.requires_grad and self.detector.width > 1
                    and density.dtype == torch.float32 and density.dim() == 3 and not batch_window):
                # pose gradient only, one channel (the registration loop): the render kernel generates the rays
                # itself and the backward is one fixed-order kernel -- no [B, n, 3] targets in between
                img = render_from_camera(density, cam, self.renderer.make_spec(**kwargs), self.detector.height,
                                         self.detector.width)
            else:
                source, target, img = rays_from_camera(cam, self.detector.height, self.detector.width)
                kwargs["mask"] = self.mask if mask_to_channels else None
                img = self.renderer(density, source, target, img, **kwargs)
        else:
            pose = args[0] if parameterization is None else convert(
                *args, parameterization=parameterization, convention=convention)
            batch_size = len(pose)
            source, target = self.detector(pose, calibration)
            img = self.render(density, source, target, mask_to_channels, **kwargs)
        return self.reshape_transform(img, batch_size=batch_size)

    def _camera_affine_cached(self):
        key = (id(self.detector), self._affine_inverse.device, self._affine_inverse._version)
        hit = getattr(self, "_cam_affine", None)
        if hit is None or hit[0] != key:
            hit = self._cam_affine = (key, *self.camera_affine())
        return hit[1], hit[2]

    def camera(self, pose: RigidTransform) -> torch.Tensor:
        """[B,24] = {Mv, s_v, Mw, s_w}: target_vox(i,j) = Mv (i,j,1)^T, raylen(i,j) = |Mw (i,j,1)^T - s_w|.
        Folds the calibration, the reorientation, the pose and the CT's inverse affine (tiny torch ops,
        differentiable w.r.t. the pose)."""
        d = self.detector
        return self._camera_from_matrix(pose.matrix, d._calibration, d._reorient, self._affine_inverse[0], self._e3)

    def _camera_from_matrix(self, matrix, C, reorient, A, e3):
        d = self.detector
        t0 = float((-d.height) // 2) + (1.0 if d.height % 2 else 0.5)
        s0 = float((-d.width) // 2) + (1.0 if d.width % 2 else 0.5)
        sg = -1.0 if d.reverse_x_axis else 1.0
        z = e3[0]   # a 0 that lives on the device already (no host->device copy: graph-capture safe)
        Kc = torch.stack([
            torch.stack([C[0, 0], z, C[0, 0] * t0 + C[0, 3]]),
            torch.stack([z, sg * C[1, 1], sg * C[1, 1] * s0 + C[1, 3]]),
            torch.stack([z, z, C[2, 2]]),
        ])
        P = matrix @ reorient                    # reorient first, then the camera pose
        t_P = P[:, :3, 3]
        Mw = P[:, :3, :3] @ Kc + t_P[:, :, None] * e3
        Mv = A[:3, :3] @ Mw + A[:3, 3][None, :, None] * e3
        s_v = t_P @ A[:3, :3].T + A[:3, 3]
        B = matrix.shape[0]
        return torch.cat([Mv.reshape(B, 9), s_v, Mw.reshape(B, 9), t_P], dim=1)

    def camera_affine(self):
        """The camera vector is AFFINE in the top three rows of the pose matrix: cam = G vec(M[:3,:4]) + c.
        Returns (G [24,12], c [24]) as float32 on the module's device, evaluated in float64 on the 12 basis
        matrices -- the constants of one pyramid stage that xvr_pose_camera_forward consumes."""
        d = self.detector
        f64 = dict(dtype=torch.float64, device="cpu")
        basis = torch.zeros(13, 4, 4, **f64)
        for k in range(12):
            basis[k + 1, k // 4, k % 4] = 1.0
        cams = self._camera_from_matrix(basis, d._calibration.to(**f64), d._reorient.to(**f64),
                                        self._affine_inverse[0].to(**f64), self._e3.to(**f64))
        c = cams[0]
        G = (cams[1:] - c).T.contiguous()
        dev = self._affine.device
        return G.to(device=dev, dtype=torch.float32), c.to(device=dev, dtype=torch.float32)

    def render(self, density, source, target, mask_to_channels=False, **kwargs):
        img = (target - source).norm(dim=-1).unsqueeze(1)        # world-mm length of every ray
        source = self.affine_inverse(source)                      # world -> voxel index coordinates
        target = self.affine_inverse(target)
        kwargs["mask"] = self.mask if mask_to_channels else None
        return self.renderer(density, source, target, img, **kwargs)

    # ------------------------------------------------------------------ intrinsics
    def set_intrinsics_(self, sdd=None, delx=None, dely=None, x0=None, y0=None, height=None, width=None):
        d = self.detector
        self.detector = Detector(
            d.sdd if sdd is None else sdd,
            d.height if height is None else height,
            d.width if width is None else width,
            d.delx if delx is None else delx,
            d.dely if dely is None else dely,
            -d.x0 if x0 is None else x0,   # the calibration stores x0/y0 as given; the properties negate
            -d.y0 if y0 is None else y0,
            reorient=d._reorient, reverse_x_axis=d.reverse_x_axis,
        ).to(self._affine.device)
        self._sync_ray_grid()

    def rescale_detector_(self, scale: float):
        d = self.detector
        self.set_intrinsics_(height=int(d.height * scale), width=int(d.width * scale),
                             delx=d.delx / scale, dely=d.dely / scale)

    # ------------------------------------------------------------------ point projection
    # Exact inverses of Detector.forward's pixel -> world map (pixel (row, col) has plane
    # coefficients t = row - H/2 + 0.5, s = +/-(col - W/2 + 0.5) for even sizes).
    def _pixel_offsets(self):
        d = self.detector
        h_off = 1.0 if d.height % 2 else 0.5
        w_off = 1.0 if d.width % 2 else 0.5
        return (-d.height // 2) + h_off, (-d.width // 2) + w_off

    def perspective_projection(self, pose: RigidTransform, pts: torch.Tensor) -> torch.Tensor:
        """World points [B|1, P, 3] -> detector pixel coordinates [B, P, 2] as (col, row)."""
        d = self.detector
        cam = d.reorient.compose(pose).inverse()(pts)
        C = d._calibration
        u = cam[..., 0] * (C[2, 2] / cam[..., 2])
        v = cam[..., 1] * (C[2, 2] / cam[..., 2])
        t = (u - C[0, 3]) / C[0, 0]
        s = (v - C[1, 3]) / C[1, 1]
        if d.reverse_x_axis:
            s = -s
        t0, s0 = self._pixel_offsets()
        return torch.stack([s - s0, t - t0], dim=-1)

    def inverse_projection(self, pose: RigidTransform, pts: torch.Tensor) -> torch.Tensor:
        """Detector pixel coordinates [B, P, 2] (col, row) -> world points on the detector plane."""
        d = self.detector
        t0, s0 = self._pixel_offsets()
        s = pts[..., 0] + s0
        t = pts[..., 1] + t0
        if d.reverse_x_axis:
            s = -s
        C = d._calibration
        cam = torch.stack([t * C[0, 0] + C[0, 3], s * C[1, 1] + C[1, 3], torch.full_like(s, C[2, 2].item())], dim=-1)
        return d.reorient.compose(pose)(cam)
