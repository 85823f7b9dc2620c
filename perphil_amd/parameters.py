"""
``DPPParameters`` — mirror of reference ``src/perphil/models/dpp/parameters.py:5-53``:
same fields, defaults (k1 = beta = mu = 1, scale_contrast = 1e2, k2 = k1/scale_contrast when None),
coercion to ``Constant`` and the derived ``eta = sqrt(beta (k1 + k2) / (k1 k2))``.
"""
from __future__ import annotations

import attr

from . import fd


@attr.define
class DPPParameters:
    k1: float | fd.Constant = 1.0
    k2: float | fd.Constant | None = None
    beta: float | fd.Constant = 1.0
    mu: float | fd.Constant = 1.0
    scale_contrast: float = 1e2

    def __attrs_post_init__(self):
        # every coefficient ends up as a Constant (so that eta and the forms can do arithmetic on them); the
        # micro-scale permeability defaults to the macro-scale one divided by the contrast
        as_constant = lambda v: v if isinstance(v, fd.Constant) else fd.Constant(v)   # noqa: E731
        self.k1 = as_constant(self.k1)
        self.k2 = as_constant(self.k1 / self.scale_contrast if self.k2 is None else self.k2)
        for name in ("beta", "mu"):
            setattr(self, name, as_constant(getattr(self, name)))

    @property
    def eta(self) -> fd.Constant:
        return fd.sqrt(self.beta * (self.k1 + self.k2) / (self.k1 * self.k2))
