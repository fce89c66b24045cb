"""Numerical convention of a render (host side).

diffdrr==0.6.0, which defines these constants for the reference, is not available here
(SURVEY.md F2-F4): every one of them is therefore an explicit, documented knob rather than a
hard-coded guess.  Defaults are the self-consistent geometry
described in DESIGN.md ("Semantics"); the recalled upstream variants are one keyword away.
"""

from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class RenderSpec:
    renderer: str = "trilinear"  # "trilinear" | "siddon"
    voxel_shift: float = 0.5     # integer coordinate = voxel corner (0.0) or centre (0.5)
    eps: float = 1e-8            # added to (target - source)
    eps_in_xyz: bool = True      # False: eps only guards the alpha divisions.  The sample points move by alpha * eps <= 1e-8
                                 # voxels, below fp32 resolution: the kernels are the same for both settings (a knob of the
                                 # pin grid, tools/pin_against_diffdrr.py)
    align_corners: bool = False  # grid_sample's flag
    norm_dims_offset: int = 0    # xyz normalised with shape + offset (recalled upstream: +1 siddon, -1 trilinear)
    # trilinear
    n_points: int = 500
    near: float = 0.0
    far: float = 1.0
    step_mode: str = "n_points"  # out = L * sum / n_points   |  "n_minus_1": / (n_points - 1)
    clip_to_volume: bool | str = False  # True: rescale alphas per ray to [alphamin, alphamax]; "batch": ONE window for the whole
                                        # call, [min alphamin, max alphamax] over its rays (computed on the device)
    # siddon (the HIP traversal is per ray by construction)
    per_ray_clamp: bool = True
    filter_intersections_outside_volume: bool = True

    def with_(self, **kw) -> "RenderSpec":
        return replace(self, **kw)

    def validate(self) -> None:
        if self.renderer not in ("trilinear", "siddon"):
            raise ValueError(f"renderer must be 'trilinear' or 'siddon', got {self.renderer!r}")
        if self.step_mode not in ("n_points", "n_minus_1"):
            raise ValueError(f"unknown step_mode {self.step_mode!r}")
        if self.n_points < 1 or (self.step_mode == "n_minus_1" and self.n_points < 2):
            raise ValueError("n_points too small")
        if self.clip_to_volume not in (False, True, "batch"):
            raise ValueError(f"clip_to_volume must be False, True or 'batch', got {self.clip_to_volume!r}")
        if not self.far >= self.near:
            raise ValueError("far must be >= near")
        if self.renderer == "siddon" and self.align_corners and self.norm_dims_offset == -1:
            # index = rint(a x + b) with a = (S - 1) / (S - 1) = 1 and b = voxel_shift: the MIDPOINT of every fully crossed plane cell
            # (x = c - voxel_shift + 1/2) maps to exactly c + 1/2, for every cell, every size and either shift.  Which of two voxels
            # a whole segment is credited with is then decided by the last bit of float32 arithmetic -- in torch as in any kernel --
            # for most segments of most rays: the render is not a function of its inputs in any useful sense, and the fuzz soak of
            # rounds 3 and 4 kept finding seeds where two correct traversals disagree on > 1 % of the voxels (seed 52075 and its
            # class).  Refused rather than rendered.  One structural tie per axis (dims = shape + 1 on an
            # even-sized axis: its middle cell) is served; tests/conftest.py::has_structural_tie.
            raise ValueError("siddon with align_corners=True and norm_dims_offset=-1 is degenerate: the midpoint of every fully crossed "
                             "cell lies exactly on a rounding boundary of the index map (see xvr_amd/spec.py)")
        if self.renderer == "siddon" and not self.per_ray_clamp:
            raise NotImplementedError(
                "per_ray_clamp=False (the literal batch-filtered sort formulation) is an oracle-only "
                "mode: a per-ray traversal cannot reproduce a batch-dependent column filter"
            )

    def index_map(self, shape):
        """(a, b) per axis with sampling index = a * x + b (see oracle/diffdrr_restated.py::index_map)."""
        a, b = [], []
        for S in shape:
            dims = S + self.norm_dims_offset
            if self.align_corners:
                a.append((S - 1) / dims)
                b.append(self.voxel_shift * (S - 1) / dims)
            else:
                a.append(S / dims)
                b.append(self.voxel_shift * S / dims - 0.5)
        return a, b
