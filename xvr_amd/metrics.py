"""Image-similarity metrics of the two outer loops (the step right after the renderer).

Drop-ins for what xvr takes from ``diffdrr.metrics``:
  MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])   /root/reference/src/xvr/registrar/base.py:119-121,
                                                                   /root/reference/src/xvr/model/loss.py:16
  GradientNormalizedCrossCorrelation2d(11, sigma)                  /root/reference/src/xvr/registrar/base.py:122
  DoubleGeodesicSE3(sdd)                                           /root/reference/src/xvr/model/loss.py:18

The reference's patch NCC unfolds every p x p patch into a channel ([b, (H-p+1)^2, p, p]: 81x the image
at p = 9, 121x at p = 11) and z-scores each; here the same quantity is computed from five
box filters -- mean(x1), mean(x2), mean(x1^2), mean(x2^2), mean(x1 x2) over every patch --

    ncc_patch = (E[x1 x2] - mu1 mu2) / sqrt((var1 + eps)(var2 + eps)),   score = mean over patches,

with the box sums accumulated in float64 (E[x^2] - mu^2 cancels catastrophically in fp32 on flat
patches, where eps = 1e-5 decides the value).  That formulation serves the configurations of the shim that the fused HIP kernels (similarity.py) do not: torch ops on the
GPU, autograd-differentiable, no CPU path; checked against the literal unfold formulation in oracle/metrics_restated.py.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from .pose import RigidTransform


def _box_mean(x: torch.Tensor, p: int) -> torch.Tensor:
    return F.avg_pool2d(x, kernel_size=p, stride=1)


class NormalizedCrossCorrelation2d(torch.nn.Module):
    def __init__(self, patch_size: int | None = None, eps: float = 1e-5):
        super().__init__()
        self.patch_size = patch_size
        self.eps = eps

    def forward(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        assert x1.shape == x2.shape, "Input images must be the same size"
        if not (x1.is_cuda and x2.is_cuda):
            raise RuntimeError("NormalizedCrossCorrelation2d: CUDA images only (the similarity runs on the device, no CPU path)")
        dt = x1.dtype
        a, b = x1.double(), x2.double()
        if self.patch_size is None:
            mu1, mu2 = a.mean(dim=[-1, -2], keepdim=True), b.mean(dim=[-1, -2], keepdim=True)
            d1, d2 = a - mu1, b - mu2
            v1 = (d1 * d1).mean(dim=[-1, -2]) + self.eps
            v2 = (d2 * d2).mean(dim=[-1, -2]) + self.eps
            score = (d1 * d2).mean(dim=[-1, -2]) / (v1 * v2).sqrt()      # [b, c]
            return score.mean(dim=1).to(dt)
        p = self.patch_size
        # centre globally first (NCC is shift invariant; keeps the box sums small)
        a = a - a.mean(dim=[-1, -2], keepdim=True)
        b = b - b.mean(dim=[-1, -2], keepdim=True)
        m1, m2 = _box_mean(a, p), _box_mean(b, p)
        v1 = (_box_mean(a * a, p) - m1 * m1).clamp_min(0) + self.eps
        v2 = (_box_mean(b * b, p) - m2 * m2).clamp_min(0) + self.eps
        cov = _box_mean(a * b, p) - m1 * m2
        return (cov / (v1 * v2).sqrt()).mean(dim=[1, 2, 3]).to(dt)


class MultiscaleNormalizedCrossCorrelation2d(torch.nn.Module):
    def __init__(self, patch_sizes=(None,), patch_weights=(1.0,), eps: float = 1e-5):
        super().__init__()
        assert len(patch_sizes) == len(patch_weights), "Each scale must have a weight"
        self.nccs = torch.nn.ModuleList([NormalizedCrossCorrelation2d(p, eps) for p in patch_sizes])
        self.patch_weights = list(patch_weights)

    def forward(self, x1, x2):
        if self._hip_ok(x1, x2):
            from .similarity import fused_mncc   # (imported late: similarity.py imports this module)

            return fused_mncc(x1, x2, self.nccs[1].patch_size, self.nccs[1].eps)
        # (any other configuration of the shim -- other weights, patches beyond 15, several channels: box filters on the device)
        return sum(w * ncc(x1, x2) for w, ncc in zip(self.patch_weights, self.nccs))

    # ``[None, p]`` with weights ``[0.5, 0.5]`` -- the configuration of both of xvr's loops
    # (/root/reference/src/xvr/registrar/base.py:119-121, /root/reference/src/xvr/model/loss.py:16) -- on float32
    # CUDA images of one channel goes through the fused HIP kernels (xvr_sim_ncc_forward_backward with
    # beta = 1, pre_transformed = 1): one thread per patch from LDS tiles instead of fp64 box filters.
    # FUSED = False keeps the torch formulation (the cross-check in tests).
    FUSED = True

    def _hip_ok(self, x1, x2):
        if not (self.FUSED and len(self.nccs) == 2 and self.patch_weights == [0.5, 0.5]):
            return False
        p0, p1 = self.nccs[0].patch_size, self.nccs[1].patch_size
        if p0 is not None or p1 is None or not (2 <= p1 <= 15) or self.nccs[0].eps != self.nccs[1].eps:
            return False
        return (x1.is_cuda and x2.is_cuda and x1.dtype == x2.dtype == torch.float32 and x1.shape == x2.shape and x1.dim() == 4
                and x1.shape[1] == 1 and x1.shape[0] > 0 and min(x1.shape[2:]) >= p1)


class Sobel(torch.nn.Module):
    """Fixed 3x3 Sobel pair (2 output channels), optional 5x5 Gaussian pre-blur when sigma > 0."""

    def __init__(self, sigma: float = 0.0):
        super().__init__()
        self.sigma = sigma
        gx = torch.tensor([[1.0, 0.0, -1.0], [2.0, 0.0, -2.0], [1.0, 0.0, -1.0]])
        gy = torch.tensor([[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]])
        self.register_buffer("weight", torch.stack([gx, gy]).unsqueeze(1))

    def _blur(self, img):
        x = torch.linspace(-2.0, 2.0, 5, dtype=img.dtype, device=img.device)
        k = torch.exp(-0.5 * (x / self.sigma) ** 2)
        k = k / k.sum()
        c = img.shape[1]
        img = F.pad(img, (2, 2, 2, 2), mode="reflect")
        img = F.conv2d(img, k.view(1, 1, 1, 5).expand(c, 1, 1, 5), groups=c)
        return F.conv2d(img, k.view(1, 1, 5, 1).expand(c, 1, 5, 1), groups=c)

    def forward(self, img):
        if self.sigma and self.sigma > 0:
            img = self._blur(img)
        return F.conv2d(img, self.weight.to(img), padding=1)


class GradientNormalizedCrossCorrelation2d(NormalizedCrossCorrelation2d):
    def __init__(self, patch_size: int | None = None, sigma: float = 1.0, **kwargs):
        super().__init__(patch_size, **kwargs)
        self.sobel = Sobel(sigma)

    # A patch gradient-NCC without pre-blur -- the second half of the registration similarity,
    # /root/reference/src/xvr/registrar/base.py:122 with the default sigma = 0 -- on float32 CUDA images of one
    # channel goes through the fused HIP kernels (beta = 0, pre_transformed = 1).  FUSED = False: torch.
    FUSED = True

    def forward(self, x1, x2):
        p = self.patch_size
        if (self.FUSED and p is not None and 2 <= p <= 15 and x1.is_cuda and x2.is_cuda
                and x1.dtype == x2.dtype == torch.float32 and x1.shape == x2.shape and x1.dim() == 4 and x1.shape[1] == 1
                and x1.shape[0] > 0 and min(x1.shape[2:]) >= max(p, 3)):
            from .similarity import fused_gncc

            if self.sobel.sigma and self.sobel.sigma > 0:   # pre-blur as a HIP kernel pair, then the plain Sobel pair
                # (this module's own 3x3 weights without its blur: no submodule is created in forward -- state_dict()
                #  stays what __init__ made it, and nothing is copied to the device under a graph capture)
                plain = lambda img: F.conv2d(img, self.sobel.weight.to(img), padding=1)  # noqa: E731
                return fused_gncc(x1, x2, p, self.eps, plain, self.sobel.sigma)
            return fused_gncc(x1, x2, p, self.eps, self.sobel)
        return super().forward(self.sobel(x1), self.sobel(x2))


class DoubleGeodesicSE3(torch.nn.Module):
    """(angular, translational, double) geodesic distances between two batches of poses, in mm:
    angular = sdd/2 * rotation angle, double = sqrt(angular^2 + translational^2 + eps)."""

    def __init__(self, sdd: float, eps: float = 1e-6):
        super().__init__()
        self.sdd = sdd
        self.eps = eps

    def forward(self, pose_1: RigidTransform, pose_2: RigidTransform):
        R = pose_1.matrix[..., :3, :3].transpose(-1, -2) @ pose_2.matrix[..., :3, :3]
        # rotation angle as atan2(|axis part|, cos): exactly ~0 for identical rotations (the reference's
        # Evaluator relies on that, metrics/evaluator.py:15,33) and with finite gradients everywhere,
        # unlike acos of a clamped trace
        cos = (R.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
        sin = 0.5 * torch.sqrt((R[..., 2, 1] - R[..., 1, 2]) ** 2 + (R[..., 0, 2] - R[..., 2, 0]) ** 2
                               + (R[..., 1, 0] - R[..., 0, 1]) ** 2 + 1e-24)
        angular = 0.5 * self.sdd * torch.atan2(sin, cos)
        trans = (pose_1.matrix[..., :3, 3] - pose_2.matrix[..., :3, 3]).norm(dim=-1)
        return angular, trans, (angular.square() + trans.square() + self.eps).sqrt()


class Equalize(torch.nn.Module):
    """Differentiable (soft-histogram) histogram equalisation of images in [0, 1]
    (/root/reference/src/xvr/utils/preprocess.py:34-66): Gaussian-kernel histogram over n_bins, its
    normalised CDF, and each pixel mapped to the CDF averaged with its own bin weights."""

    def __init__(self, n_bins: int = 256, tau: float = 0.01, eps: float = 1e-10):
        super().__init__()
        self.n_bins, self.tau, self.eps = n_bins, tau, eps

    def forward(self, x):
        """float32 CUDA images [B, 1, H, W]: HIP kernels (xvr_sim_equalize_forward / _backward), no [pixels x bins] matrix.  The
        line-by-line torch restatement they are checked against is oracle/metrics_restated.py::equalize."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 1):
            raise RuntimeError("Equalize: float32 CUDA images of shape [B, 1, H, W] only (HIP kernels, no CPU path)")
        if not 2 <= self.n_bins <= 1024:
            raise ValueError("Equalize: n_bins must be in [2, 1024]")
        if x.shape[0] == 0:
            return x.clone()
        from .similarity import equalize_hip

        return equalize_hip(x, self.n_bins, self.tau, self.eps)


class _StandardizeNormalize(torch.autograd.Function):
    """Standardize (min / max over the tensor, or per image) -> Normalize as two HIP calls (xvr_sim_transform_forward /
    _backward, include/xvr_sim.h): the same float32 values as the torch lines of XrayTransforms.forward; the backward
    carries the gradient through the min and the max as torch's does."""

    @staticmethod
    def forward(ctx, x, per_image, mean, std):
        from . import _lib
        from .renderers import _ptr, _stream

        lib = _lib.load()
        B, n = x.shape[0], x[0].numel()
        xc = x.contiguous()
        if xc.data_ptr() % 16:   # (a contiguous view at an odd offset: the kernels want 16-byte aligned images)
            xc = xc.clone()
        y = torch.empty_like(xc)
        state = torch.empty(lib.xvr_sim_transform_state_bytes(B), dtype=torch.uint8, device=x.device)
        _lib.check(lib.xvr_sim_transform_forward(_ptr(xc), B, n, int(per_image), float(mean), float(std), 1e-6, _ptr(y), _ptr(state),
                                                 _stream()), "xvr_sim_transform_forward")
        ctx.save_for_backward(xc, state)
        ctx.cfg = (B, n, int(per_image), float(mean), float(std))
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from .renderers import _ptr, _stream

        xc, state = ctx.saved_tensors
        B, n, per_image, mean, std = ctx.cfg
        gc = g.contiguous()
        gx = torch.empty_like(xc)
        _lib.check(_lib.load().xvr_sim_transform_backward(_ptr(xc), _ptr(gc), B, n, per_image, mean, std, 1e-6, _ptr(gx), _ptr(state),
                                                          _stream()), "xvr_sim_transform_backward")
        return gx, None, None, None


class XrayTransforms(torch.nn.Module):
    """Standardize (global min-max) -> [Equalize] -> Resize((h, w)) -> Normalize(mean, std), applied to
    every rendered DRR each iteration (/root/reference/src/xvr/utils/preprocess.py:5-31; call sites
    /root/reference/src/xvr/registrar/base.py:213-218,250, /root/reference/src/xvr/model/trainer.py:207,216)."""

    def __init__(self, height: int, width: int | None = None, mean: float = 0.15, std: float = 0.1, equalize: bool = False,
                 per_image: bool = False):
        super().__init__()
        self.equalize = Equalize() if equalize else None
        self.height, self.width = height, height if width is None else width
        self.mean, self.std = mean, std
        # per_image: every image of the batch standardised by its OWN min / max (the reference's transform takes them
        # over the whole tensor, which is the same thing for the single image it is ever given); batched multi-start
        # needs the images of a batch to be independent problems
        self.per_image = per_image

    def forward(self, x):
        """float32 CUDA images [B, C, H, W].  Standardize -> Normalize is one pair of HIP calls (_StandardizeNormalize) when nothing
        sits between them; with ``equalize`` the chain is Standardize (HIP) -> Equalize + Normalize (HIP, one pass).  A Resize -- the
        full-resolution X-ray once per pyramid stage, never a DRR rendered at the stage's size -- is torch's antialiased bilinear
        interpolation between the HIP steps.  oracle/metrics_restated.py::xray_transforms holds the torch lines all of this is
        checked against."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise RuntimeError("XrayTransforms: float32 CUDA images of shape [B, C, H, W] only (HIP kernels, no CPU path)")
        if x.numel() == 0:
            return x.clone()
        if x.shape[0] > 65535:
            raise ValueError("XrayTransforms: at most 65535 images per call")
        resize = tuple(x.shape[-2:]) != (self.height, self.width)
        if self.equalize is None and not resize:
            return _StandardizeNormalize.apply(x, self.per_image, self.mean, self.std)
        x = _StandardizeNormalize.apply(x, self.per_image, 0.0, 1.0)          # Standardize alone: ((x - lo) / r - 0) * 1, the same bits
        if self.equalize is not None:
            from .similarity import equalize_hip

            if x.shape[1] != 1:
                raise RuntimeError("XrayTransforms(equalize=True): one-channel images only")
            if not resize:
                return equalize_hip(x, self.equalize.n_bins, self.equalize.tau, self.equalize.eps, self.mean, self.std)
            x = self.equalize(x)
        x = F.interpolate(x, size=(self.height, self.width), mode="bilinear", antialias=True, align_corners=False)
        return (x - self.mean) / self.std
