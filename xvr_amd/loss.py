"""Loss of the pose-regressor training step (the consumer of the two renders per step).

Restates /root/reference/src/xvr/model/loss.py:5-89: ``PoseRegressionLoss`` = weight_ncc * (1 - mNCC)
+ weight_dice * Dice + weight_geo * double-geodesic (+ optional multiview consistency), with the in-tree
``DiceLoss`` / ``DiceMetric`` (background = channel 0 excluded, nan-mean over structures).
"""

from __future__ import annotations

import torch

from .metrics import DoubleGeodesicSE3, MultiscaleNormalizedCrossCorrelation2d


class DiceMetric(torch.nn.Module):
    """2D Dice between two multi-channel label maps, background (channel 0) excluded; reduction none."""

    def forward(self, y_pred, y_true):
        y_pred = y_pred.reshape(y_pred.shape[0], y_pred.shape[1], -1).to(torch.float32)
        y_true = y_true.reshape(y_true.shape[0], y_true.shape[1], -1).to(torch.float32)
        intersection = (y_pred * y_true).sum(dim=2)
        dice = (2.0 * intersection) / (y_pred.sum(dim=2) + y_true.sum(dim=2))
        return dice[:, 1:]


class DiceLoss(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.dice = DiceMetric()

    def forward(self, img1, img2):
        return 1 - self.dice(img1, img2).nanmean(dim=1).nan_to_num()


class PoseRegressionLoss(torch.nn.Module):
    def __init__(self, sdd: float, weight_ncc: float = 1e0, weight_geo: float = 1e-2, weight_dice: float = 1e0,
                 weight_mvc: float = 1e-3):
        super().__init__()
        self.imagesim = MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])
        self.diceloss = DiceLoss()
        self.geodesic = DoubleGeodesicSE3(sdd)
        self.weight_ncc, self.weight_geo = weight_ncc, weight_geo
        self.weight_dice, self.weight_mvc = weight_dice, weight_mvc

    def forward(self, img, mask, pose, pred_img, pred_mask, pred_pose):
        mncc = self.imagesim(img, pred_img)
        dice = self.diceloss(mask, pred_mask)
        rgeo, tgeo, dgeo = self.geodesic(pose, pred_pose)
        loss = self.weight_ncc * (1 - mncc) + self.weight_dice * dice + self.weight_geo * dgeo
        mvc = self.multiview_consistency(pose, pred_pose)
        if self.weight_mvc > 0:
            loss = loss + self.weight_mvc * mvc.mean()
        return loss, mncc, dgeo, rgeo, tgeo, dice, mvc

    def multiview_consistency(self, true_pose, pred_pose):
        assert (B := len(true_pose)) == len(pred_pose)
        idx, jdx = torch.triu_indices(B, B, offset=1)
        if len(idx) == 0:
            return torch.zeros(1, device=true_pose.matrix.device)
        _, _, dgeo_relative = self.geodesic(true_pose[jdx] @ true_pose[idx].inverse(),
                                            pred_pose[jdx] @ pred_pose[idx].inverse())
        return dgeo_relative
