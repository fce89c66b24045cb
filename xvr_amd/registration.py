"""``Registration``: holds the pose being optimised as two ``nn.Parameter``s and renders it.

Drop-in for ``diffdrr.registration.Registration`` as xvr uses it:
``Registration(drr, rot, xyz, parameterization, convention)``, ``.rotation``, ``.translation``,
``.pose``, ``.drr``, ``reg()``  (/root/reference/src/xvr/registrar/base.py:169,201,212,224-225,249).
"""

from __future__ import annotations

import torch

from .drr import DRR
from .pose import N_ANGULAR_COMPONENTS, convert  # noqa: F401  (re-exported like the reference module)


class Registration(torch.nn.Module):
    def __init__(self, drr: DRR, rotation: torch.Tensor, translation: torch.Tensor,
                 parameterization: str, convention: str | None = None):
        super().__init__()
        self.drr = drr
        self.rotation = torch.nn.Parameter(rotation.detach().clone())
        self.translation = torch.nn.Parameter(translation.detach().clone())
        self.parameterization = parameterization
        self.convention = convention

    @property
    def pose(self):
        return convert(self.rotation, self.translation,
                       parameterization=self.parameterization, convention=self.convention)

    def forward(self, **kwargs):
        # (the DRR turns Euler parameters into its camera with one HIP launch; other parameterisations go
        #  through convert() as before)
        return self.drr(self.rotation, self.translation, parameterization=self.parameterization,
                        convention=self.convention, **kwargs)
