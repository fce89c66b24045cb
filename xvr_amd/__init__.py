"""xvr_amd: MI355X-native differentiable-DRR render path behind the DRR / Registration surface of xvr."""

from .spec import RenderSpec  # noqa: F401
from .pose import RigidTransform, convert, make_matrix  # noqa: F401

__version__ = "0.1.0"
