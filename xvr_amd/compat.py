"""``install_as_diffdrr()``: make ``import diffdrr...`` resolve to this package, so that xvr's render-path
imports (``from diffdrr.drr import DRR``, ``from diffdrr.pose import convert`` ...) land on the MI355X path.

Scope: the RENDER-PATH names only.  ``diffdrr.visualization`` (``plot_drr``,
``plot_mask``), ``diffdrr.utils.resample`` and ``diffdrr.data.load_example_ct`` are NOT provided -- plotting and
example data are outside SURVEY.md section 8 -- so the reference modules that import them at module scope
(``registrar/base.py:12``, ``model/trainer.py:8``, ``model/inference.py:3``) do not import over this shim as they
stand; ``renderer/load.py``, ``model/sampler.py`` and ``model/loss.py`` do (tests/test_reference_goldens.py).
Mapped:

    diffdrr.drr            DRR
    diffdrr.data           read, transform_hu_to_density
    diffdrr.pose           RigidTransform, convert, make_matrix
    diffdrr.registration   Registration, N_ANGULAR_COMPONENTS
    diffdrr.metrics        MultiscaleNormalizedCrossCorrelation2d, GradientNormalizedCrossCorrelation2d,
                           NormalizedCrossCorrelation2d, DoubleGeodesicSE3
    diffdrr.renderers      Siddon, Trilinear
    diffdrr.detector       Detector

A real ``diffdrr`` that is already importable is left alone unless ``force=True``.
"""

from __future__ import annotations

import importlib.util
import sys
import types


def install_as_diffdrr(force: bool = False) -> bool:
    if not force and "diffdrr" not in sys.modules and importlib.util.find_spec("diffdrr") is not None:
        return False  # the real package is installed: do not shadow it
    from . import data, detector, drr, metrics, pose, registration, renderers

    pkg = types.ModuleType("diffdrr")
    pkg.__path__ = []  # mark as a package
    pkg.__doc__ = "compatibility alias installed by xvr_amd.compat (MI355X-native render path)"
    table = {
        "drr": (drr, ["DRR"]),
        "data": (data, ["read", "transform_hu_to_density", "Subject"]),
        "pose": (pose, ["RigidTransform", "convert", "make_matrix"]),
        "registration": (registration, ["Registration", "N_ANGULAR_COMPONENTS"]),
        "metrics": (metrics, ["MultiscaleNormalizedCrossCorrelation2d", "GradientNormalizedCrossCorrelation2d",
                              "NormalizedCrossCorrelation2d", "DoubleGeodesicSE3"]),
        "renderers": (renderers, ["Siddon", "Trilinear"]),
        "detector": (detector, ["Detector"]),
    }
    sys.modules["diffdrr"] = pkg
    for sub, (mod, names) in table.items():
        m = types.ModuleType(f"diffdrr.{sub}")
        for n in names:
            setattr(m, n, getattr(mod, n))
        m.__all__ = list(names)
        sys.modules[f"diffdrr.{sub}"] = m
        setattr(pkg, sub, m)
    return True
