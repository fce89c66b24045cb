"""C-arm detector geometry: pose -> (source, target) ray endpoints in world mm.

Mirrors what xvr reaches as ``drr.detector``:

* ``source, target = self.drr.detector(pose, None)``      /root/reference/src/xvr/model/trainer.py:283
* ``drr.detector.height / .width / .sdd / .delx``          /root/reference/src/xvr/registrar/base.py:214-215,
                                                           /root/reference/src/xvr/metrics/evaluator.py:15,21
* rebuilt by ``set_intrinsics_`` / ``rescale_detector_``   /root/reference/src/xvr/registrar/base.py:155,212

Geometry (restated, diffdrr==0.6.0 is not in the tree -- SURVEY.md Appendix A): the source starts at
the origin and the H x W pixel grid, unit-spaced and centred, sits on the plane z = 1; the calibration
matrix diag(dely, delx, sdd) + principal point (y0, x0) scales it to millimetres; ``reorient`` then the
camera pose place it in the world.  Rows of the grid vary slowest (ray r = row * W + col).
"""

from __future__ import annotations

import torch

from .pose import RigidTransform

REORIENT = {
    "AP": [[1.0, 0, 0, 0], [0, 0, -1.0, 0], [0, 1.0, 0, 0], [0, 0, 0, 1.0]],
    "PA": [[1.0, 0, 0, 0], [0, 0, 1.0, 0], [0, 1.0, 0, 0], [0, 0, 0, 1.0]],
    None: [[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]],
}


def make_reorient(orientation):
    if orientation not in REORIENT:
        raise ValueError(f"orientation must be 'AP', 'PA' or None, got {orientation!r}")
    return torch.tensor(REORIENT[orientation], dtype=torch.float32)


class Detector(torch.nn.Module):
    def __init__(self, sdd, height, width, delx, dely, x0, y0, reorient, reverse_x_axis=False):
        super().__init__()
        self.height = int(height)
        self.width = int(width)
        self.reverse_x_axis = bool(reverse_x_axis)
        source, target = self._initialize_carm()
        self.register_buffer("source", source)
        self.register_buffer("target", target)
        self.register_buffer("_reorient", torch.as_tensor(reorient, dtype=torch.float32))
        self.register_buffer(
            "_calibration",
            torch.tensor(
                [[dely, 0, 0, y0], [0, delx, 0, x0], [0, 0, sdd, 0], [0, 0, 0, 1]], dtype=torch.float32
            ),
        )

    # -- read-only intrinsics (plain floats, like the reference's properties) --
    @property
    def sdd(self):
        return self._calibration[2, 2].item()

    @property
    def delx(self):
        return self._calibration[1, 1].item()

    @property
    def dely(self):
        return self._calibration[0, 0].item()

    @property
    def x0(self):
        return -self._calibration[1, -1].item()

    @property
    def y0(self):
        return -self._calibration[0, -1].item()

    @property
    def reorient(self) -> RigidTransform:
        return RigidTransform(self._reorient)

    @property
    def calibration(self) -> RigidTransform:
        return RigidTransform(self._calibration)

    def _initialize_carm(self):
        h_off = 1.0 if self.height % 2 else 0.5
        w_off = 1.0 if self.width % 2 else 0.5
        t = torch.arange(-self.height // 2, self.height // 2, dtype=torch.float32) + h_off
        s = torch.arange(-self.width // 2, self.width // 2, dtype=torch.float32) + w_off
        if self.reverse_x_axis:
            s = -s
        coefs = torch.cartesian_prod(t, s).reshape(-1, 2)
        target = torch.cat([coefs, torch.ones(len(coefs), 1)], dim=-1)[None]  # plane z = 1
        source = torch.zeros(1, 1, 3)
        return source, target

    def forward(self, extrinsic: RigidTransform, calibration: RigidTransform | None = None):
        """-> source [B,1,3], target [B,H*W,3] (world mm)."""
        calib = self.calibration if calibration is None else calibration
        target = calib(self.target)
        pose = self.reorient.compose(extrinsic)  # reorient first, then the camera pose
        return pose(self.source), pose(target)
