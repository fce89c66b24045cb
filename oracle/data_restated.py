"""ORACLE (test infrastructure, never shipped, never on the product path).

``diffdrr.data.transform_hu_to_density`` as xvr calls it before every training render
(/root/reference/src/xvr/model/trainer.py:124,196-197), restated in stock torch ops from SURVEY.md Appendix A (A9): air
(<= -800 HU) is set to the minimum soft-tissue value, bone (> 350 HU) is scaled by the multiplier, then min-max normalised.
** PARITY UNPINNED ** for the thresholds (diffdrr 0.6.0 is absent, see oracle/diffdrr_restated.py); the checker of
xvr_drr_hu_stats / xvr_drr_hu_to_density.  Only ``tests/`` may import this.
"""
import torch


def transform_hu_to_density(volume: torch.Tensor, bone_attenuation_multiplier: float) -> torch.Tensor:
    volume = volume.to(torch.float32)
    air = volume <= -800
    bone = volume > 350
    soft = ~(air | bone)
    soft_min = volume[soft].min() if soft.any() else volume.min()
    density = torch.where(air, soft_min, volume)
    density = torch.where(bone, volume * bone_attenuation_multiplier, density)
    density = density - density.min()
    return density / density.max().clamp_min(torch.finfo(torch.float32).tiny)
