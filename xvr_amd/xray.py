"""Pixel pipeline of a raw X-ray before registration: what ``xvr register`` applies to the DICOM's pixel array
(/root/reference/src/xvr/io/xray.py:93-129, called from /root/reference/src/xvr/registrar/base.py:128-141) -- collimator
border trim, unit-range rescale, optional background removal (mode intensity), optional exponential -> linear conversion,
and the reduction of a multi-frame acquisition to one image.

Independently written as a small configurable pipeline object that runs wherever its input lives (the X-ray is prepared once
per registration, on the device when the caller has put it there); DICOM parsing itself is out of scope (pydicom is not in
this image).  Pinned by vectors the reference's own function produced in the build container
(tests/golden/make_golden_xray.py -> tests/golden/xvr_reference_xray.npz, tests/test_xray.py).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Union

import torch

FrameReduction = Union[str, int, Callable, None]


def _centred_window(size: int, keep: int) -> slice:
    """The ``keep`` central elements of an axis of length ``size``.  An odd surplus cannot be split evenly: the extra element
    goes where torchvision's crop (which the reference calls) puts it -- first index = round-half-to-even of surplus / 2."""
    if not 0 < keep <= size:
        raise ValueError(f"cannot keep {keep} of {size} pixels")
    first = int(round((size - keep) / 2.0))
    return slice(first, first + keep)


@dataclass(frozen=True)
class XrayPreparation:
    """``prep(pixels) -> image``; ``pixels`` is [1, 1, H, W] or, multi-frame, [1, 1, T, H, W] (float).

    trim                 pixels removed from the height and from the width IN TOTAL (the window stays centred): edge artefacts
                         of the collimator.  The registrar's pyramid compensates for it (registrar.parse_scales).
    subtract_background  the most frequent intensity becomes the top of the range, everything brighter is clipped to it
    linearize            I = I0 exp(-line integral)  ->  line integral up to a constant: log(max) - log(.) of the image shifted by 1
    frames               multi-frame only: "max" | "sum" | a frame index | a callable on the 5-D tensor | None (keep 5-D)
    """

    trim: int = 0
    subtract_background: bool = False
    linearize: bool = True
    frames: FrameReduction = "max"

    def __call__(self, pixels: torch.Tensor) -> torch.Tensor:
        if pixels.dim() not in (4, 5) or not pixels.is_floating_point():
            raise ValueError("XrayPreparation: a float tensor [1, 1, H, W] or [1, 1, T, H, W]")
        x = pixels
        if self.trim:
            h, w = x.shape[-2:]
            x = x[..., _centred_window(h, h - self.trim), _centred_window(w, w - self.trim)]
        lo, hi = torch.aminmax(x)
        x = (x - lo) / (hi - lo + 1e-6)
        if self.subtract_background:
            x = (x - torch.mode(x.reshape(-1)).values).clamp(-1, 0) + 1
        if self.linearize:
            x = x + 1
            x = x.max().log() - x.log()
        return self._one_image(x) if x.dim() == 5 else x

    def _one_image(self, x: torch.Tensor) -> torch.Tensor:
        how = self.frames
        if how is None:
            return x
        if isinstance(how, bool):
            raise ValueError(f"XrayPreparation: frames={how!r} is neither a reduction nor a frame index")
        if isinstance(how, int):
            return x.select(2, how)
        if callable(how):
            return how(x)
        if how == "max":
            return x.amax(dim=2)
        if how == "sum":
            return x.sum(dim=2)
        raise ValueError(f"XrayPreparation: unknown frame reduction {how!r}")
