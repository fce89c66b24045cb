"""Loss of the pose-regressor training step (the consumer of the two renders per step).

Restates /root/reference/src/xvr/model/loss.py:5-89: ``PoseRegressionLoss`` = weight_ncc * (1 - mNCC)
+ weight_dice * Dice + weight_geo * double-geodesic (+ optional multiview consistency), with the in-tree
``DiceLoss`` / ``DiceMetric`` (background = channel 0 excluded, nan-mean over structures).
"""

from __future__ import annotations

import torch

from .metrics import DoubleGeodesicSE3, MultiscaleNormalizedCrossCorrelation2d


def _hip_poses(*poses) -> bool:
    return all(p.matrix.is_cuda and p.matrix.dtype == torch.float32 for p in poses)


class _Geodesic(torch.autograd.Function):
    """DoubleGeodesicSE3(sdd)(a, b) in one HIP launch (xvr_pose_geodesic); differentiable w.r.t. b through the
    double geodesic, which is all the training loss uses (angular / translational parts are returned for logging)."""

    @staticmethod
    def forward(ctx, a, b, sdd, eps):
        from . import _lib
        from .renderers import _ptr, _stream

        lib = _lib.load()
        N = a.shape[0]
        a_c, b_c = a.contiguous(), b.contiguous()
        out = torch.empty(3, N, device=a.device, dtype=torch.float32)
        gb = torch.empty(N, 12, device=a.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if N > 0:   # (an empty batch -- every sample dropped by `keep`, trainer.py:202-204 -- is three empty [0] results, no launch)
            _lib.check(lib.xvr_pose_geodesic(_ptr(a_c), _ptr(b_c), N, float(sdd), float(eps), _ptr(out), _ptr(gb), _stream()),
                       "xvr_pose_geodesic")
        ctx.save_for_backward(gb)
        # the SAME tensor objects must be marked and returned (marking a temporary view has no effect): the fused
        # kernel carries the gradient of the combined distance only, so the angular and translational terms are
        # honestly non-differentiable here instead of silently dropping an incoming gradient
        ang, trans, dist = out.unbind(0)
        ctx.mark_non_differentiable(ang, trans)
        return ang, trans, dist

    @staticmethod
    def backward(ctx, _g_ang, _g_trans, g_d):
        (gb,) = ctx.saved_tensors
        if gb is None:
            return None, None, None, None
        grad = torch.zeros(gb.shape[0], 4, 4, device=gb.device, dtype=gb.dtype)
        g = gb * g_d.reshape(-1, 1)
        grad[:, :3, :3] = g[:, :9].reshape(-1, 3, 3)
        grad[:, :3, 3] = g[:, 9:]
        return None, grad, None, None


class _Multiview(torch.autograd.Function):
    """Multiview consistency over all pose pairs (xvr_pose_multiview_forward / _backward)."""

    @staticmethod
    def forward(ctx, true_m, pred_m, sdd, eps):
        from . import _lib
        from .renderers import _ptr, _stream

        lib = _lib.load()
        B = true_m.shape[0]
        t_c, p_c = true_m.contiguous(), pred_m.contiguous()
        mvc = torch.empty(B * (B - 1) // 2, device=true_m.device, dtype=torch.float32)
        _lib.check(lib.xvr_pose_multiview_forward(_ptr(t_c), _ptr(p_c), B, float(sdd), float(eps), _ptr(mvc), _stream()),
                   "xvr_pose_multiview_forward")
        ctx.save_for_backward(t_c, p_c)
        ctx.cfg = (float(sdd), float(eps))
        return mvc

    @staticmethod
    def backward(ctx, g_mvc):
        from . import _lib
        from .renderers import _ptr, _stream

        lib = _lib.load()
        t_c, p_c = ctx.saved_tensors
        B = t_c.shape[0]
        gp = torch.empty(B, 12, device=t_c.device, dtype=torch.float32)
        _lib.check(lib.xvr_pose_multiview_backward(_ptr(t_c), _ptr(p_c), _ptr(g_mvc.contiguous()), B, *ctx.cfg, _ptr(gp), _stream()),
                   "xvr_pose_multiview_backward")
        grad = torch.zeros(B, 4, 4, device=gp.device, dtype=gp.dtype)
        grad[:, :3, :3] = gp[:, :9].reshape(B, 3, 3)
        grad[:, :3, 3] = gp[:, 9:]
        return None, grad, None, None


class DiceMetric(torch.nn.Module):
    """2D Dice between two multi-channel BOOLEAN label maps [B, C, ...] (what the trainer passes: ``mask = img > 0``,
    /root/reference/src/xvr/model/trainer.py:292), background (channel 0) excluded, no reduction: [B, C - 1], NaN where a
    structure is absent from both maps.  One HIP launch of integer counts (xvr_sim_dice_bool) -- the float32 values of the
    reference's float formulation (/root/reference/src/xvr/model/loss.py:63-89; oracle/loss_restated.py is the checker)."""

    def forward(self, y_pred, y_true):
        if not (y_pred.is_cuda and y_true.is_cuda and y_pred.dtype == torch.bool and y_true.dtype == torch.bool):
            raise RuntimeError("DiceMetric: boolean CUDA label maps only (HIP kernel, no CPU path)")
        if y_pred.shape != y_true.shape or y_pred.dim() < 3:
            raise ValueError(f"DiceMetric: label maps {tuple(y_pred.shape)} and {tuple(y_true.shape)} must be equal [B, C, ...] shapes")
        B, C = y_pred.shape[:2]
        n = y_pred.shape[2:].numel()
        dice = torch.empty(B, C, device=y_pred.device, dtype=torch.float32)
        if B > 0 and C > 0 and n > 0:   # (every sample dropped by `keep`: an empty [0, C - 1] result, as the torch lines give)
            from . import _lib
            from .renderers import _ptr, _stream

            a, b = y_pred.contiguous(), y_true.contiguous()
            _lib.check(_lib.load().xvr_sim_dice_bool(_ptr(a), _ptr(b), B, C, n, _ptr(dice), _stream()), "xvr_sim_dice_bool")
        else:
            dice.fill_(float("nan"))
        return dice[:, 1:]


class DiceLoss(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.dice = DiceMetric()

    def forward(self, img1, img2):
        return 1 - self.dice(img1, img2).nanmean(dim=1).nan_to_num()


class PoseRegressionLoss(torch.nn.Module):
    # the pose terms (double geodesic, multiview consistency over all pairs) are one HIP launch each instead of ~200 tiny torch
    # launches; the torch formulation they are checked against lives in oracle/loss_restated.py

    def __init__(self, sdd: float, weight_ncc: float = 1e0, weight_geo: float = 1e-2, weight_dice: float = 1e0,
                 weight_mvc: float = 1e-3):
        super().__init__()
        self.imagesim = MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])
        self.diceloss = DiceLoss()
        self.geodesic = DoubleGeodesicSE3(sdd)
        self.weight_ncc, self.weight_geo = weight_ncc, weight_geo
        self.weight_dice, self.weight_mvc = weight_dice, weight_mvc

    def forward(self, img, mask, pose, pred_img, pred_mask, pred_pose):
        """-> (loss [B], mncc, dgeo, rgeo, tgeo, dice, mvc), the reference's tuple (loss.py:25-42).  Every term is a HIP call:
        fused mNCC, boolean Dice, one launch for the geodesics, one for the multiview term (true poses carry no gradient)."""
        if not _hip_poses(pose, pred_pose) or pose.matrix.requires_grad:
            raise RuntimeError("PoseRegressionLoss: float32 CUDA poses, the true pose without a gradient (HIP kernels, no CPU path)")
        if len(pose) != len(pred_pose):
            raise ValueError("PoseRegressionLoss: the two batches of poses differ in length")
        if len(pose) == 0:
            # every sample of the step was dropped by `keep` (trainer.py:202-204): empty [0] terms, as the reference's per-image
            # torch lines give (its DiceMetric's `.view(0, C, -1)` raises instead, loss.py:73, which the trainer swallows per
            # step, trainer.py:171-175); no kernel is launched on nothing.  The loss stays attached to the prediction (a zero-size sum is 0 with a
            # zero gradient), so `loss.mean().backward()` of the caller behaves as it does upstream.
            z = pred_pose.matrix.sum(dim=(-1, -2)) * 0 + pred_img.sum(dim=tuple(range(1, pred_img.dim()))) * 0
            e = z.detach()
            return z, e, e, e, e, e, e
        mncc = self.imagesim(img, pred_img)
        dice = self.diceloss(mask, pred_mask)
        rgeo, tgeo, dgeo = _Geodesic.apply(pose.matrix, pred_pose.matrix, self.geodesic.sdd, self.geodesic.eps)
        mvc = self.multiview_consistency(pose, pred_pose)
        terms = (self.weight_ncc * (1 - mncc), self.weight_dice * dice, self.weight_geo * dgeo)
        loss = terms[0] + terms[1] + terms[2]
        if self.weight_mvc > 0:
            loss = loss + self.weight_mvc * mvc.mean()
        return loss, mncc, dgeo, rgeo, tgeo, dice, mvc

    def multiview_consistency(self, true_pose, pred_pose):
        """Double geodesic between the relative poses of every pair (i < j), true against predicted: [B (B - 1) / 2]."""
        if len(true_pose) != len(pred_pose):
            raise ValueError("multiview_consistency: the two batches of poses differ in length")
        if not _hip_poses(true_pose, pred_pose) or true_pose.matrix.requires_grad:
            raise RuntimeError("multiview_consistency: float32 CUDA poses, the true pose without a gradient (HIP kernels, no CPU path)")
        if len(true_pose) < 2:
            return torch.zeros(1, device=true_pose.matrix.device)
        return _Multiview.apply(true_pose.matrix, pred_pose.matrix, self.geodesic.sdd, self.geodesic.eps)
