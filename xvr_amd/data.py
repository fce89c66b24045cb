"""Minimal CT container for the render path (what xvr gets from ``diffdrr.data.read``).

Reference call sites: /root/reference/src/xvr/renderer/load.py:29 (``read(volume, mask, labels,
orientation, **read_kwargs)``), /root/reference/src/xvr/model/utils.py:40,155.  torchio / nibabel are
not in this image, so a ``Subject`` here is a plain container built from tensors (file readers are a
"next" row of SURVEY.md section 8f); it carries exactly the fields the DRR module consumes:
``volume`` (HU or density) [D0,D1,D2], ``affine`` 4x4 voxel->world (mm), ``density``, ``mask``
(float label map), ``orientation``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


def transform_hu_to_density(volume: torch.Tensor, bone_attenuation_multiplier: float) -> torch.Tensor:
    """Piecewise HU -> density map used before every training render
    (/root/reference/src/xvr/model/trainer.py:124,196-197): air (<= -800 HU) is set to the minimum
    soft-tissue value, bone (> 350 HU) is scaled, then the result is min-max normalised to [0, 1]."""
    volume = volume.to(torch.float32)
    air = volume <= -800
    bone = volume > 350
    soft = ~(air | bone)
    soft_min = volume[soft].min() if soft.any() else volume.min()
    density = torch.where(air, soft_min, volume)
    density = torch.where(bone, volume * bone_attenuation_multiplier, density)
    density = density - density.min()
    return density / density.max().clamp_min(torch.finfo(torch.float32).tiny)


@dataclass
class Subject:
    volume: torch.Tensor                 # [D0, D1, D2]
    affine: torch.Tensor                 # 4x4 voxel index -> world mm
    density: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None  # [D0, D1, D2] labels
    orientation: Optional[str] = "AP"

    def get_center(self):
        """World coordinates of the volume's isocentre (as torchio's Image.get_center)."""
        size = torch.tensor(self.volume.shape, dtype=torch.float64)
        c = (size - 1) / 2
        A = self.affine.to(torch.float64)
        return tuple((A[:3, :3] @ c + A[:3, 3]).tolist())


def read(volume, labelmap=None, labels=None, orientation="AP", bone_attenuation_multiplier=1.0,
         affine=None, spacing=(1.0, 1.0, 1.0), center_volume=True, hu=False) -> Subject:
    """Build a Subject from tensors.  ``hu=True`` converts HU to density first; otherwise ``volume``
    is taken to be a density already.  ``center_volume`` puts the isocentre at the world origin."""
    if not torch.is_tensor(volume):
        raise NotImplementedError("file readers (NIfTI) are not part of this round; pass a tensor")
    volume = volume.to(torch.float32)
    if volume.dim() != 3:
        raise ValueError("volume must be [D0, D1, D2]")
    if affine is None:
        affine = torch.diag(torch.tensor([*spacing, 1.0], dtype=torch.float32))
    affine = torch.as_tensor(affine, dtype=torch.float32).clone()
    if center_volume:
        c = (torch.tensor(volume.shape, dtype=torch.float32) - 1) / 2
        affine[:3, 3] = -(affine[:3, :3] @ c)
    mask = None
    if labelmap is not None:
        mask = labelmap.to(torch.float32)
        if labels is not None:
            keep = torch.zeros_like(mask, dtype=torch.bool)
            for lab in labels:
                keep |= mask == float(lab)
            mask = torch.where(keep, mask, torch.zeros_like(mask))
    density = transform_hu_to_density(volume, bone_attenuation_multiplier) if hu else volume
    if mask is not None and labels is not None:
        density = torch.where(mask > 0, density, torch.zeros_like(density))
    return Subject(volume=volume, affine=affine, density=density, mask=mask, orientation=orientation)


def make_phantom(size=64, n_ellipsoids=12, n_labels=0, seed=0, noise=0.02, device="cpu"):
    """Seeded synthetic density phantom (sum of random ellipsoids + smoothed noise, clamped to [0,1])
    -- the stand-in for a CT everywhere in tests and bench ("Synthetic inputs").
    Returns (density[D,D,D], labels[D,D,D] or None)."""
    g = torch.Generator().manual_seed(seed)
    D = (size, size, size) if isinstance(size, int) else tuple(size)
    dev = torch.device(device)
    ax = [torch.arange(d, dtype=torch.float32, device=dev) for d in D]
    vol = torch.zeros(D, dtype=torch.float32, device=dev)
    lab = torch.zeros(D, dtype=torch.float32, device=dev) if n_labels else None
    for i in range(n_ellipsoids):
        c = [(0.2 + 0.6 * torch.rand(1, generator=g).item()) * d for d in D]
        rad = [(0.04 + 0.12 * torch.rand(1, generator=g).item()) * d for d in D]
        rho = 0.2 + 0.8 * torch.rand(1, generator=g).item()
        q = (((ax[0] - c[0]) / rad[0]) ** 2)[:, None, None] + (((ax[1] - c[1]) / rad[1]) ** 2)[None, :, None] \
            + (((ax[2] - c[2]) / rad[2]) ** 2)[None, None, :]
        inside = q <= 1.0
        vol += rho * inside
        if n_labels:
            lab = torch.where(inside, torch.full_like(lab, float(1 + i % (n_labels - 1))), lab)
    if noise > 0:
        nz = torch.randn(D, generator=g).to(dev)
        nz = torch.nn.functional.avg_pool3d(nz[None, None], 3, stride=1, padding=1)[0, 0]
        vol += noise * nz
    # a soft body envelope so that most rays see non-zero density
    q = sum((((ax[i] - (D[i] - 1) / 2) / (0.45 * D[i])) ** 2).reshape([-1 if j == i else 1 for j in range(3)]) for i in range(3))
    vol += 0.15 * (q <= 1.0)
    return vol.clamp_(0, 1), lab
