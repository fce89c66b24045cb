"""Fused image similarity of the registration loop: XrayTransforms + mNCC + gradient-NCC, value and
gradient in one C-ABI call (include/xvr_sim.h, xvr_amd/csrc/sim_kernels.hip).

Replaces, per iteration, ``pred = transform(pred); loss = imagesim(img, pred); loss.backward()``
(/root/reference/src/xvr/registrar/base.py:250-252) for the default similarity
``beta * MultiscaleNCC([None, p1]) + (1 - beta) * GradientNCC(p2, sigma=0)``
(/root/reference/src/xvr/registrar/base.py:115-123); ``EqualizedSimilarity`` does the same for ``equalize=True``.  The box-filter
torch formulation in ``xvr_amd.metrics`` remains the general path (patches > 15) and the cross-check.
"""

from __future__ import annotations

import ctypes

import torch

from . import _lib
from .metrics import Sobel
from .renderers import _ptr, _stream, _timed


class _FusedNCC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, moving, fixed, fixed_sobel, spec, workspace):
        lib = _lib.load()
        B, _, H, W = moving.shape
        mov = moving.contiguous()
        loss = torch.empty(B, device=moving.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(mov) if need else None
        rc = _timed("ncc_forward_backward", lib.xvr_sim_ncc_forward_backward, _ptr(fixed), _ptr(fixed_sobel), _ptr(mov),
                    B, H, W, ctypes.byref(spec), _ptr(loss), _ptr(grad), _ptr(workspace), workspace.numel() * 4, _stream())
        _lib.check(rc, "xvr_sim_ncc_forward_backward")
        ctx.save_for_backward(grad)
        ctx.per_image = bool(spec.per_image)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        # The kernels return d(sum_b loss[b]) / d moving: Standardize's min/max couple the images of a
        # batch, so per-image upstream weights are only exact for one image or a uniform weight --
        # unless every image is standardised on its own (per_image), where they are independent.
        if ctx.per_image:
            return grad * gout.reshape(-1, 1, 1, 1), None, None, None, None
        if gout.numel() == 1 or gout.stride(0) == 0:
            return grad * gout.reshape(-1)[0], None, None, None, None
        if not bool((gout == gout[0]).all()):
            raise NotImplementedError("FusedSimilarity: non-uniform per-image loss weights in a batch; "
                                      "use xvr_amd.metrics (plain torch) for that")
        return grad * gout[0], None, None, None, None


class FusedSimilarity(torch.nn.Module):
    """``sim(moving_raw) -> [B]`` against a fixed, already transformed target image."""

    def __init__(self, fixed: torch.Tensor, mncc_patch_size=9, gncc_patch_size=11, beta=0.5, mean=0.15, std=0.1,
                 eps=1e-5, per_image=False):
        super().__init__()
        if not fixed.is_cuda or fixed.dtype != torch.float32 or fixed.dim() != 4 or fixed.shape[1] != 1:
            raise RuntimeError("FusedSimilarity needs a float32 CUDA target of shape [B,1,H,W] (HIP kernels, no CPU path)")
        self.register_buffer("fixed", fixed.contiguous())
        self.register_buffer("fixed_sobel", Sobel(0.0).to(fixed.device)(fixed).contiguous())
        self.spec = _lib.CSimSpec(mean, std, 1e-6, eps, beta, int(mncc_patch_size), int(gncc_patch_size), int(bool(per_image)), 0)
        B, _, H, W = fixed.shape
        nbytes = _lib.load().xvr_sim_workspace_bytes(B, H, W)
        self.register_buffer("workspace", torch.empty((nbytes + 3) // 4, device=fixed.device, dtype=torch.float32), persistent=False)

    @staticmethod
    def supported(height, width, mncc_patch_size, gncc_patch_size, sigma, equalize) -> bool:
        big = max(mncc_patch_size, gncc_patch_size)
        return ((not equalize) and (not sigma) and 2 <= min(mncc_patch_size, gncc_patch_size) and big <= 15
                and min(height, width) >= big and torch.cuda.is_available())

    def forward(self, moving: torch.Tensor) -> torch.Tensor:
        if moving.shape != self.fixed.shape:
            raise ValueError(f"moving {tuple(moving.shape)} and fixed {tuple(self.fixed.shape)} differ")
        return _FusedNCC.apply(moving, self.fixed, self.fixed_sobel, self.spec, self.workspace)


class _ChainNCC(torch.autograd.Function):
    """autograd face of EqualizedSimilarity.evaluate (the gradient is computed with the value, as in _FusedNCC)."""

    @staticmethod
    def forward(ctx, moving, sim):
        B = moving.shape[0]
        loss = torch.empty(B, device=moving.device, dtype=torch.float32)
        grad = torch.empty_like(sim.fixed)
        mc = moving.contiguous()
        if mc.data_ptr() % 16:   # (a contiguous view at an odd offset: xvr_sim_transform_forward wants 16-byte aligned images)
            mc = mc.clone()
        sim.evaluate(mc, loss, grad)
        ctx.save_for_backward(grad)
        ctx.per_image = sim.per_image
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        if ctx.per_image:
            return grad * gout.reshape(-1, 1, 1, 1), None
        if gout.numel() == 1 or gout.stride(0) == 0 or bool((gout == gout[0]).all()):
            return grad * gout.reshape(-1)[0], None
        raise NotImplementedError("EqualizedSimilarity: non-uniform per-image loss weights in a batch standardised as one tensor")


class EqualizedSimilarity(torch.nn.Module):
    """``sim(moving_raw [B,1,H,W]) -> [B]`` for ``equalize=True`` (sigma = 0, patches <= 15), every step a HIP call and no autograd
    tape between them (/root/reference/src/xvr/registrar/base.py:213-218,250-252 with XrayTransforms(equalize=True),
    /root/reference/src/xvr/utils/preprocess.py:5-66):

        Standardize                 xvr_sim_transform_forward (mean 0, std 1)
        Equalize + Normalize        xvr_sim_equalize_forward  (normalised output in the same pass)
        beta mNCC + (1 - beta) gNCC xvr_sim_ncc_forward_backward (pre_transformed: value and d / d transformed image)
        ... and back                xvr_sim_equalize_backward, xvr_sim_transform_backward (the min / max terms included)

    ``evaluate(img, loss, grad_img)`` is what RegistrationStage enqueues per iteration (16 launches, capturable);
    calling the module gives the same through autograd."""

    def __init__(self, fixed, mncc_patch_size=9, gncc_patch_size=11, beta=0.5, mean=0.15, std=0.1, eps=1e-5, per_image=False,
                 n_bins=256, tau=0.01, eq_eps=1e-10):
        super().__init__()
        if not fixed.is_cuda or fixed.dtype != torch.float32 or fixed.dim() != 4 or fixed.shape[1] != 1:
            raise RuntimeError("EqualizedSimilarity needs a float32 CUDA target of shape [B,1,H,W] (HIP kernels, no CPU path)")
        lib = _lib.load()
        self.register_buffer("fixed", fixed.contiguous())
        self.register_buffer("fixed_sobel", Sobel(0.0).to(fixed.device)(fixed).contiguous())
        self.spec = _lib.CSimSpec(mean, std, 1e-6, eps, beta, int(mncc_patch_size), int(gncc_patch_size), 0, 1)
        self.per_image, self.mean, self.std = bool(per_image), float(mean), float(std)
        self.n_bins, self.tau, self.eq_eps = int(n_bins), float(tau), float(eq_eps)
        B, _, H, W = fixed.shape
        f = dict(device=fixed.device, dtype=torch.float32)
        words = lambda nbytes: torch.empty((nbytes + 3) // 4, **f)   # noqa: E731
        self.workspace = words(lib.xvr_sim_workspace_bytes(B, H, W))
        self.tf_state = words(lib.xvr_sim_transform_state_bytes(B))
        self.eq_ws = words(lib.xvr_sim_equalize_workspace_bytes(B, self.n_bins))
        self.x_std, self.y_eq, self.S, self.y_out, self.g_y, self.g_std = (torch.empty_like(self.fixed) for _ in range(6))

    @staticmethod
    def supported(height, width, mncc_patch_size, gncc_patch_size, sigma, equalize) -> bool:
        big = max(mncc_patch_size, gncc_patch_size)
        return (bool(equalize) and (not sigma) and 2 <= min(mncc_patch_size, gncc_patch_size) and big <= 15
                and min(height, width) >= big and torch.cuda.is_available())

    def evaluate(self, img, loss, grad_img):
        if img.data_ptr() % 16 or not img.is_contiguous():
            raise ValueError("EqualizedSimilarity.evaluate: a contiguous, 16-byte aligned image buffer (clone a misaligned view)")
        lib, s = _lib.load(), _stream()
        B, _, H, W = self.fixed.shape
        n, pi = H * W, int(self.per_image)
        cf = ctypes.c_float
        _lib.check(_timed("transform_forward", lib.xvr_sim_transform_forward, _ptr(img), B, n, pi, cf(0.0), cf(1.0), cf(1e-6),
                          _ptr(self.x_std), _ptr(self.tf_state), s), "xvr_sim_transform_forward")
        _lib.check(_timed("equalize_forward", lib.xvr_sim_equalize_forward, _ptr(self.x_std), B, n, self.n_bins, cf(self.tau), cf(self.eq_eps),
                          cf(self.mean), cf(self.std), _ptr(self.y_eq), _ptr(self.S), _ptr(self.y_out), _ptr(self.eq_ws),
                          self.eq_ws.numel() * 4, s), "xvr_sim_equalize_forward")
        _lib.check(_timed("ncc_forward_backward", lib.xvr_sim_ncc_forward_backward, _ptr(self.fixed), _ptr(self.fixed_sobel), _ptr(self.y_out),
                          B, H, W, ctypes.byref(self.spec), _ptr(loss), _ptr(self.g_y), _ptr(self.workspace), self.workspace.numel() * 4, s),
                   "xvr_sim_ncc_forward_backward")
        _lib.check(_timed("equalize_backward", lib.xvr_sim_equalize_backward, _ptr(self.x_std), _ptr(self.y_eq), _ptr(self.S), _ptr(self.g_y),
                          B, n, self.n_bins, cf(self.tau), cf(self.eq_eps), cf(self.std), _ptr(self.g_std), _ptr(self.eq_ws),
                          self.eq_ws.numel() * 4, s), "xvr_sim_equalize_backward")
        _lib.check(_timed("transform_backward", lib.xvr_sim_transform_backward, _ptr(img), _ptr(self.g_std), B, n, pi, cf(0.0), cf(1.0),
                          cf(1e-6), _ptr(grad_img), _ptr(self.tf_state), s), "xvr_sim_transform_backward")

    def forward(self, moving):
        if moving.shape != self.fixed.shape:
            raise ValueError(f"moving {tuple(moving.shape)} and fixed {tuple(self.fixed.shape)} differ")
        return _ChainNCC.apply(moving, self)


class GeneralSimilarity(torch.nn.Module):
    """``sim(moving_raw [B,1,H,W]) -> [B]`` for the configurations the single fused call does not cover -- ``equalize``,
    ``sigma > 0``, patches beyond 15 -- as the reference composes them (/root/reference/src/xvr/registrar/base.py:115-123,
    250-251): XrayTransforms (optionally per image) then beta * mNCC + (1 - beta) * gNCC.  The NCC terms and the Gaussian
    pre-blur run in the HIP kernels (through ``xvr_amd.metrics``' dispatch), so do Standardize / Equalize / Normalize
    (``XrayTransforms``); autograd strings them together.  Plugs into ``RegistrationStage`` wherever a ``FusedSimilarity`` does;
    ``equalize`` without a pre-blur has the tape-free ``EqualizedSimilarity``."""

    def __init__(self, fixed, transform, mncc_patch_size=9, gncc_patch_size=11, sigma=0.0, beta=0.5):
        super().__init__()
        from .metrics import GradientNormalizedCrossCorrelation2d, MultiscaleNormalizedCrossCorrelation2d

        self.register_buffer("fixed", fixed.contiguous())
        self.transform = transform
        self.beta = beta
        self.sim1 = MultiscaleNormalizedCrossCorrelation2d([None, mncc_patch_size], [0.5, 0.5])
        self.sim2 = GradientNormalizedCrossCorrelation2d(gncc_patch_size, sigma).to(fixed.device)

    def forward(self, moving):
        y = self.transform(moving)
        return self.beta * self.sim1(self.fixed, y) + (1 - self.beta) * self.sim2(self.fixed, y)


class _FusedMNCC(torch.autograd.Function):
    """beta = 1: 0.5 NCC(x, y) + 0.5 patch-NCC_p(x, y) per image; beta = 0: patch-NCC_p of the Sobel pairs
    (sx, sy = Sobel(x), Sobel(y), constants of the call), for already transformed images.  NCC is symmetric, so
    the gradient w.r.t. either argument is the same kernel with the roles swapped."""

    @staticmethod
    def forward(ctx, x, y, patch, eps, beta=1.0, sx=None, sy=None):
        lib = _lib.load()
        xc, yc = x.contiguous(), y.contiguous()
        B, _, H, W = xc.shape
        spec = _lib.CSimSpec(0.0, 1.0, 0.0, float(eps), float(beta), int(patch), int(patch), 0, 1)
        sx = xc if sx is None else sx.contiguous()   # (unused by the kernels when beta = 1)
        sy = yc if sy is None else sy.contiguous()
        nbytes = lib.xvr_sim_workspace_bytes(B, H, W)
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        loss = torch.empty(B, device=x.device, dtype=torch.float32)
        gx = gy = None
        if ctx.needs_input_grad[1] or not ctx.needs_input_grad[0]:
            gy = torch.empty_like(yc) if ctx.needs_input_grad[1] else None
            rc = _timed("mncc_forward_backward", lib.xvr_sim_ncc_forward_backward, _ptr(xc), _ptr(sx), _ptr(yc), B, H, W,
                        ctypes.byref(spec), _ptr(loss), _ptr(gy), _ptr(ws), ws.numel() * 4, _stream())
            _lib.check(rc, "xvr_sim_ncc_forward_backward")
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(xc)
            rc = _timed("mncc_forward_backward", lib.xvr_sim_ncc_forward_backward, _ptr(yc), _ptr(sy), _ptr(xc), B, H, W,
                        ctypes.byref(spec), _ptr(loss), _ptr(gx), _ptr(ws), ws.numel() * 4, _stream())
            _lib.check(rc, "xvr_sim_ncc_forward_backward")
        ctx.save_for_backward(gx, gy)
        return loss

    @staticmethod
    def backward(ctx, gout):
        gx, gy = ctx.saved_tensors
        g = gout.reshape(-1, 1, 1, 1)
        return (gx * g if gx is not None else None), (gy * g if gy is not None else None), None, None, None, None, None


class _Blur5(torch.autograd.Function):
    """5-tap separable Gaussian blur with reflect padding (xvr_sim_gaussian_blur5); backward = its exact transpose."""

    @staticmethod
    def forward(ctx, x, sigma):
        ctx.sigma = float(sigma)
        return _Blur5._run(x, ctx.sigma, 0)

    @staticmethod
    def backward(ctx, g):
        return _Blur5._run(g, ctx.sigma, 1), None

    @staticmethod
    def _run(x, sigma, adjoint):
        lib = _lib.load()
        xc = x.contiguous()
        H, W = xc.shape[-2:]
        out, scratch = torch.empty_like(xc), torch.empty_like(xc)
        rc = _timed("gaussian_blur5", lib.xvr_sim_gaussian_blur5, _ptr(xc), _ptr(out), _ptr(scratch), xc.numel() // (H * W), H, W,
                    ctypes.c_float(sigma), adjoint, _stream())
        _lib.check(rc, "xvr_sim_gaussian_blur5")
        return out


class _EqualizeHIP(torch.autograd.Function):
    """xvr_sim_equalize_forward / _backward: the soft-histogram equalisation without the [pixels x bins] matrix; the result is
    (equalised - out_mean) / out_std, i.e. the Normalize that follows Equalize in XrayTransforms rides in the same pass."""

    @staticmethod
    def forward(ctx, x, n_bins, tau, eps, out_mean, out_std):
        lib = _lib.load()
        xc = x.contiguous()
        B = xc.shape[0]
        n = xc.numel() // B
        y, S, y_out = torch.empty_like(xc), torch.empty_like(xc), torch.empty_like(xc)
        nbytes = lib.xvr_sim_equalize_workspace_bytes(B, int(n_bins))
        if nbytes == 0:
            raise ValueError("Equalize (HIP): n_bins must be in [2, 1024]")
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        rc = _timed("equalize_forward", lib.xvr_sim_equalize_forward, _ptr(xc), B, n, int(n_bins), ctypes.c_float(tau), ctypes.c_float(eps),
                    ctypes.c_float(out_mean), ctypes.c_float(out_std), _ptr(y), _ptr(S), _ptr(y_out), _ptr(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "xvr_sim_equalize_forward")
        ctx.save_for_backward(xc, y, S, ws)
        ctx.args = (B, n, int(n_bins), float(tau), float(eps), float(out_std))
        return y_out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xc, y, S, ws = ctx.saved_tensors
        B, n, K, tau, eps, out_std = ctx.args
        gx = torch.empty_like(xc)
        rc = _timed("equalize_backward", lib.xvr_sim_equalize_backward, _ptr(xc), _ptr(y), _ptr(S), _ptr(g.contiguous()), B, n, K,
                    ctypes.c_float(tau), ctypes.c_float(eps), ctypes.c_float(out_std), _ptr(gx), _ptr(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "xvr_sim_equalize_backward")
        return gx, None, None, None, None, None


def equalize_hip(x, n_bins: int = 256, tau: float = 0.01, eps: float = 1e-10, out_mean: float = 0.0, out_std: float = 1.0):
    """``Equalize`` of /root/reference/src/xvr/utils/preprocess.py:34-66 as HIP kernels (float32 CUDA [B,1,H,W] in [0, 1]), followed
    by ``(y - out_mean) / out_std`` in the same pass (the identity by default)."""
    return _EqualizeHIP.apply(x, int(n_bins), float(tau), float(eps), float(out_mean), float(out_std))


def gaussian_blur5(x, sigma: float):
    """The pre-blur of ``GradientNormalizedCrossCorrelation2d(p, sigma > 0)`` as a HIP kernel pair (float32 CUDA, H, W >= 3)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("gaussian_blur5 needs a float32 CUDA tensor (HIP kernel, no CPU path)")
    return _Blur5.apply(x, float(sigma))


def fused_mncc(x, y, patch_size: int, eps: float = 1e-5):
    """``MultiscaleNormalizedCrossCorrelation2d([None, p], [0.5, 0.5])(x, y)`` -> [B] through the HIP kernels."""
    return _FusedMNCC.apply(x, y, int(patch_size), float(eps))


def fused_gncc(x, y, patch_size: int, eps: float, sobel, sigma: float = 0.0):
    """``GradientNormalizedCrossCorrelation2d(p, sigma)(x, y)`` -> [B] through the HIP kernels.  The kernels take
    the Sobel pair of the image they treat as fixed ready-made (one conv2d each, only for a side that needs it).
    ``sigma > 0``: both images go through the 5-tap Gaussian pre-blur first (``gaussian_blur5``, differentiable);
    ``sobel`` must then be the plain 3x3 pair."""
    if sigma and sigma > 0:
        x, y = gaussian_blur5(x, sigma), gaussian_blur5(y, sigma)
    with torch.no_grad():
        sx = sobel(x)
        sy = sobel(y) if x.requires_grad else None
    return _FusedMNCC.apply(x, y, int(patch_size), float(eps), 0.0, sx, sy)
