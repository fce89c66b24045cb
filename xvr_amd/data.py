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


def _hu_to_density_hip(volume: torch.Tensor, mult: float) -> torch.Tensor:
    """One streaming HIP pass (xvr_drr_hu_to_density).  The per-class statistics do not depend on the
    multiplier: they are reduced once and remembered ON the tensor object (keyed by its version counter --
    never by address: the caching allocator hands the same address to unrelated tensors)."""
    import ctypes

    from . import _lib
    from .renderers import _ptr, _stream, _timed

    lib = _lib.load()
    vol, stats = _hu_stats(volume)
    out = torch.empty_like(vol)
    rc = _timed("hu_to_density", lib.xvr_drr_hu_to_density, _ptr(vol), ctypes.c_longlong(vol.numel()), _ptr(stats),
                ctypes.c_float(mult), _ptr(out), _stream())
    _lib.check(rc, "xvr_drr_hu_to_density")
    return out


def _hu_stats(volume: torch.Tensor):
    """(contiguous volume, its per-class statistics) -- reduced once per tensor version and remembered on the tensor object."""
    import ctypes

    from . import _lib
    from .renderers import _ptr, _stream

    vol = volume if volume.is_contiguous() else volume.contiguous()
    cached = getattr(volume, "_xvr_hu_stats", None)
    if cached is not None and cached[0] == volume._version and vol is volume:
        return vol, cached[1]
    stats = torch.empty(16, dtype=torch.int32, device=vol.device)
    _lib.check(_lib.load().xvr_drr_hu_stats(_ptr(vol), ctypes.c_longlong(vol.numel()), _ptr(stats), _stream()), "xvr_drr_hu_stats")
    if vol is volume:
        try:
            volume._xvr_hu_stats = (volume._version, stats)
        except AttributeError:
            pass
    return vol, stats


class HUDensity:
    """``transform_hu_to_density(volume, multiplier, lazy=True)``: the density of a training step that has NOT been written yet.

    The reference maps the whole CT to a density volume before the two renders of every step
    (/root/reference/src/xvr/model/trainer.py:196-197): 512 MiB written and read back at 512^3.  The renders' own copy of the
    volume -- labels in the mantissa bits, y-pair tiles (xvr_drr_pack_labels_ytiles) -- is rebuilt every step anyway, so the map
    is applied on the way into it (xvr_drr_pack_hu_labels_ytiles: same bits) and the density volume never exists.  Whatever else
    asks for it -- an unmasked render, a small launch, a voxel gradient -- calls ``materialize()``, the plain HIP pass (cached)."""

    def __init__(self, hu: torch.Tensor, stats: torch.Tensor, multiplier: float):
        self.hu, self.stats, self.multiplier = hu, stats, float(multiplier)
        self._dense = None
        self._hu_version = hu._version      # the statistics belong to THIS content of the tensor

    def check_fresh(self):
        """The HU tensor must not have been modified in place since ``transform_hu_to_density(..., lazy=True)`` reduced its
        statistics: the renders' packing pass would apply a map normalised with stale min / max while a fresh ``materialize()``
        recomputes them -- two different densities from one object.  Raises instead; call ``transform_hu_to_density`` again."""
        if self.hu._version != self._hu_version:
            raise RuntimeError("HUDensity: the HU volume was modified in place after transform_hu_to_density(..., lazy=True); its "
                               "statistics are stale -- map it again")

    shape = property(lambda self: self.hu.shape)
    device = property(lambda self: self.hu.device)
    dtype = property(lambda self: self.hu.dtype)
    is_cuda = property(lambda self: self.hu.is_cuda)
    requires_grad = False

    def dim(self):
        return self.hu.dim()

    def materialize(self) -> torch.Tensor:
        self.check_fresh()
        if self._dense is None:
            self._dense = _hu_to_density_hip(self.hu, self.multiplier)
        return self._dense


def transform_hu_to_density(volume: torch.Tensor, bone_attenuation_multiplier: float, lazy: bool = False):
    """Piecewise HU -> density map used before every training render
    (/root/reference/src/xvr/model/trainer.py:124,196-197): air (<= -800 HU) is set to the minimum
    soft-tissue value, bone (> 350 HU) is scaled, then the result is min-max normalised to [0, 1].
    One fused streaming HIP pass over a float32 CUDA volume; there is no CPU path and no autograd through it (xvr's volume is
    a buffer, model/utils.py:162,169).  The torch lines it is checked against: oracle/data_restated.py."""
    if not (volume.is_cuda and volume.dtype == torch.float32):
        raise RuntimeError("transform_hu_to_density: a float32 CUDA volume (HIP kernel, no CPU path)")
    if volume.requires_grad:
        raise NotImplementedError("transform_hu_to_density: the HU map is not differentiable here (the volume is a buffer upstream)")
    if lazy:   # (see HUDensity: the masked renders of a training step take the map inside their packing pass)
        vol, stats = _hu_stats(volume)
        return HUDensity(vol, stats, float(bone_attenuation_multiplier))
    return _hu_to_density_hip(volume, float(bone_attenuation_multiplier))


def _density_at_load(volume: torch.Tensor, bone_attenuation_multiplier: float) -> torch.Tensor:
    """The ONE-OFF density of ``read()``: file loading is host code (as the reference's torchio / diffdrr.data.read is, and as
    this module's NIfTI decode is), so a CT that arrives as a host tensor gets its density on the host, once, before anything
    is on the GPU; a CUDA tensor takes the HIP pass.  Never on the per-step path -- that is transform_hu_to_density above."""
    if volume.is_cuda:
        return _hu_to_density_hip(volume.to(torch.float32), float(bone_attenuation_multiplier))
    v = volume.to(torch.float32)
    bone = v > 350
    soft = (v > -800) & ~bone
    floor_ = v[soft].min() if bool(soft.any()) else v.min()
    d = torch.where(bone, v * bone_attenuation_multiplier, torch.where(v <= -800, floor_, v))
    d = d - d.min()
    return d / d.max().clamp_min(torch.finfo(torch.float32).tiny)


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


def read_nifti(path):
    """Minimal NIfTI-1 reader (.nii / .nii.gz, single-file, numpy only -- nibabel/torchio are not in
    this image): returns (data [D0,D1,D2] float32 with scl_slope/inter applied, affine 4x4 float64),
    re-oriented to RAS+ like torchio's ToCanonical (the reference loads CTs through torchio,
    /root/reference/src/xvr/model/utils.py:52-56)."""
    import gzip
    import struct

    import numpy as np

    path = str(path)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    for endian in ("<", ">"):
        if struct.unpack(endian + "i", raw[:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad NIfTI-1 magic {raw[344:348]!r}")
    dim = struct.unpack(endian + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(endian + "hh", raw[70:74])
    pixdim = struct.unpack(endian + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(endian + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(endian + "hh", raw[252:256])
    dtypes = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4"}
    if datatype not in dtypes:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    shape = tuple(int(d) for d in dim[1:1 + max(int(dim[0]), 3)]) + (1,) * (3 - min(int(dim[0]), 3))
    shape3 = shape[:3]
    if any(d != 1 for d in shape[3:]):
        raise ValueError(f"{path}: only 3-D volumes are supported, got dims {shape}")
    count = shape3[0] * shape3[1] * shape3[2]
    data = np.frombuffer(raw, dtype=np.dtype(endian + dtypes[datatype]), count=count, offset=int(vox_offset))
    data = data.reshape(shape3, order="F").astype(np.float32)
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            data = data * np.float32(slope) + np.float32(inter)
    if sform_code > 0:
        affine = np.eye(4)
        affine[0] = struct.unpack(endian + "4f", raw[280:296])
        affine[1] = struct.unpack(endian + "4f", raw[296:312])
        affine[2] = struct.unpack(endian + "4f", raw[312:328])
    elif qform_code > 0:
        b, c, d = struct.unpack(endian + "3f", raw[256:268])
        off = struct.unpack(endian + "3f", raw[268:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        affine = np.eye(4)
        affine[:3, :3] = R @ np.diag([pixdim[1], pixdim[2], pixdim[3] * qfac])
        affine[:3, 3] = off
    else:
        affine = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    # canonicalise to RAS+: permute / flip axes so that the affine's rotation part is closest to identity
    lin = affine[:3, :3]
    perm = [int(np.argmax(np.abs(lin[i]))) for i in range(3)]  # world axis i <- voxel axis perm[i]
    if sorted(perm) != [0, 1, 2]:
        perm = list(np.argsort(-np.abs(lin), axis=1)[:, 0])
        if sorted(perm) != [0, 1, 2]:
            raise ValueError(f"{path}: cannot canonicalise an affine this oblique")
    data = np.transpose(data, perm)
    P = np.zeros((4, 4))
    for new, old in enumerate(perm):
        P[old, new] = 1.0
    P[3, 3] = 1.0
    affine = affine @ P
    for ax in range(3):
        if affine[ax, ax] < 0:
            data = np.flip(data, axis=ax)
            F = np.eye(4)
            F[ax, ax] = -1.0
            F[ax, 3] = data.shape[ax] - 1
            affine = affine @ F
    return np.ascontiguousarray(data), affine


def read(volume, labelmap=None, labels=None, orientation="AP", bone_attenuation_multiplier=1.0,
         affine=None, spacing=(1.0, 1.0, 1.0), center_volume=True, hu=None) -> Subject:
    """Build a Subject from tensors or from NIfTI files (``volume`` / ``labelmap`` as paths, like the
    reference's ``read(volume, mask, labels, orientation)``, /root/reference/src/xvr/renderer/load.py:29).
    ``hu`` converts HU to density first (default: True for files, False for tensors, which are taken
    to be densities already).  ``center_volume`` puts the isocentre at the world origin."""
    if not torch.is_tensor(volume):
        data, aff = read_nifti(volume)
        volume = torch.from_numpy(data)
        affine = torch.from_numpy(aff).to(torch.float32) if affine is None else affine
        hu = True if hu is None else hu
        if labelmap is not None and not torch.is_tensor(labelmap):
            ldata, _ = read_nifti(labelmap)
            labelmap = torch.from_numpy(ldata)
    hu = bool(hu)
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
    density = _density_at_load(volume, bone_attenuation_multiplier) if hu else volume
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
