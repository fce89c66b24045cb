"""Vectors produced by the REFERENCE's own files (src/xvr/model/sampler.py, src/xvr/model/loss.py), executed in
the build container over the diffdrr compat shim (tests/golden/make_golden_xvr.py).  They pin xvr's own
logic -- the pose sampling recipe, the Dice definition, the loss assembly and the multiview pairing."""
import sys
from pathlib import Path

import numpy as np
import torch

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "xvr_reference_sampler_loss.npz")


def test_get_random_pose_matches_the_reference_sampler():
    from xvr_amd.training import get_random_pose

    ranges = dict(zip((str(k) for k in GOLD["sampler_keys"]), GOLD["sampler_ranges"].tolist()))
    ranges["batch_size"] = int(ranges["batch_size"])
    torch.manual_seed(123)
    mine = get_random_pose(**ranges).matrix.numpy()
    assert np.allclose(mine, GOLD["sampler_matrix"], atol=1e-5)
    # sanity of the recipe itself: sources sit ty mm from the isocentre, looking back at it
    assert (np.linalg.norm(mine[:, :3, 3], axis=1) >= 450 - 1e-3).all()


def test_pose_regression_loss_matches_the_reference_loss_module():
    from xvr_amd.loss import DiceMetric, PoseRegressionLoss
    from xvr_amd.pose import RigidTransform

    t = lambda k: torch.from_numpy(GOLD[k])  # noqa: E731
    fn = PoseRegressionLoss(1020.0, weight_mvc=1e-3)
    res = fn(t("in_img"), t("in_mask"), RigidTransform(t("in_pose")), t("in_pred_img"), t("in_pred_mask"),
             RigidTransform(t("in_pred_pose")))
    for k, v in zip(("loss", "mncc", "dgeo", "rgeo", "tgeo", "dice", "mvc"), res):
        assert np.allclose(v.detach().numpy(), GOLD[f"loss_{k}"], rtol=1e-5, atol=1e-6), k
    d = DiceMetric()(t("in_mask").float(), t("in_pred_mask").float()).numpy()
    assert np.allclose(d, GOLD["dice_metric"], equal_nan=True)
    assert np.isnan(GOLD["dice_metric"][0, 1])  # the structure that is empty in both masks


def test_compat_shim_exposes_the_diffdrr_names_xvr_imports():
    saved = {k: v for k, v in sys.modules.items() if k == "diffdrr" or k.startswith("diffdrr.")}
    try:
        from xvr_amd.compat import install_as_diffdrr

        assert install_as_diffdrr(force=True)
        from diffdrr.data import read, transform_hu_to_density  # noqa: F401
        from diffdrr.drr import DRR  # noqa: F401
        from diffdrr.metrics import (DoubleGeodesicSE3, GradientNormalizedCrossCorrelation2d,  # noqa: F401
                                     MultiscaleNormalizedCrossCorrelation2d)
        from diffdrr.pose import RigidTransform, convert, make_matrix  # noqa: F401
        from diffdrr.registration import N_ANGULAR_COMPONENTS, Registration  # noqa: F401

        import xvr_amd.drr

        assert DRR is xvr_amd.drr.DRR and N_ANGULAR_COMPONENTS["quaternion_adjugate"] == 10
    finally:
        for k in [k for k in sys.modules if k == "diffdrr" or k.startswith("diffdrr.")]:
            del sys.modules[k]
        sys.modules.update(saved)
