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
        if not isinstance(self.k1, fd.Constant):
            self.k1 = fd.Constant(self.k1)
        if self.k2 is None:
            self.k2 = self.k1 / self.scale_contrast
        if not isinstance(self.k2, fd.Constant):
            self.k2 = fd.Constant(self.k2)
        if not isinstance(self.beta, fd.Constant):
            self.beta = fd.Constant(self.beta)
        if not isinstance(self.mu, fd.Constant):
            self.mu = fd.Constant(self.mu)

    @property
    def eta(self) -> fd.Constant:
        return fd.sqrt(self.beta * (self.k1 + self.k2) / (self.k1 * self.k2))
