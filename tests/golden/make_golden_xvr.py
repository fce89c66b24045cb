"""Golden vectors from the REFERENCE's own code, generated in this container.

Two files of /root/reference import nothing but torch and ``diffdrr``: ``src/xvr/model/sampler.py``
(``get_random_pose``) and ``src/xvr/model/loss.py`` (``PoseRegressionLoss``, ``DiceLoss``, ``DiceMetric``,
multiview consistency).  With ``xvr_amd.compat.install_as_diffdrr()`` standing in for the absent diffdrr
package they execute here, unmodified, loaded by path (their package ``__init__`` pulls timm / torchio,
which are not installed).  What these vectors pin is therefore xvr's OWN logic -- the sampling recipe
(uniform ranges, circle shift, degrees, ZXY), the Dice definition, the loss assembly and the pairing of
poses in the multiview term -- on top of this package's pose algebra and metrics.

    python tests/golden/make_golden_xvr.py        (needs /root/reference; the .npz it writes is committed)
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference/src/xvr/model")
sys.path.insert(0, str(ROOT))

from xvr_amd.compat import install_as_diffdrr  # noqa: E402

install_as_diffdrr(force=True)


def load(name):
    spec = importlib.util.spec_from_file_location(f"xvr_ref_{name}", REF / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sampler, loss = load("sampler"), load("loss")
    out = {}
    ranges = dict(alphamin=135.0, alphamax=225.0, betamin=-45.0, betamax=45.0, gammamin=-15.0, gammamax=15.0,
                  txmin=-150.0, txmax=150.0, tymin=450.0, tymax=1000.0, tzmin=-150.0, tzmax=150.0, batch_size=7)
    torch.manual_seed(123)
    out["sampler_matrix"] = sampler.get_random_pose(**ranges).matrix.numpy()
    out["sampler_ranges"] = np.array([ranges[k] for k in sorted(ranges)], dtype=np.float64)
    out["sampler_keys"] = np.array(sorted(ranges))

    g = torch.Generator().manual_seed(7)
    B, C, H = 5, 4, 24
    img = torch.rand(B, 1, H, H, generator=g)
    pred_img = 0.8 * img + 0.2 * torch.rand(B, 1, H, H, generator=g)
    mask = torch.rand(B, C, H, H, generator=g) > 0.6
    pred_mask = torch.rand(B, C, H, H, generator=g) > 0.5
    mask[0, 2] = False
    pred_mask[0, 2] = False  # an empty structure in both: nan in the metric, ignored by the loss
    from xvr_amd.pose import convert

    rot = (torch.rand(B, 3, generator=g) - 0.5)
    xyz = torch.tensor([[0.0, 700.0, 0.0]]).repeat(B, 1) + 30 * (torch.rand(B, 3, generator=g) - 0.5)
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    pred = convert(rot + 0.05 * torch.randn(B, 3, generator=g), xyz + 8 * torch.randn(B, 3, generator=g),
                   parameterization="euler_angles", convention="ZXY")
    fn = loss.PoseRegressionLoss(1020.0, weight_mvc=1e-3)
    res = fn(img, mask, pose, pred_img, pred_mask, pred)
    for k, v in zip(("loss", "mncc", "dgeo", "rgeo", "tgeo", "dice", "mvc"), res):
        out[f"loss_{k}"] = v.detach().numpy()
    out["dice_metric"] = loss.DiceMetric()(mask.float(), pred_mask.float()).numpy()
    for k, v in dict(img=img, pred_img=pred_img, mask=mask, pred_mask=pred_mask, pose=pose.matrix, pred_pose=pred.matrix).items():
        out[f"in_{k}"] = v.numpy()
    np.savez_compressed(Path(__file__).resolve().parent / "xvr_reference_sampler_loss.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
