"""
``dpp_form`` / ``dpp_delayed_form`` / ``dpp_splitted_form`` — mirror of reference
``src/perphil/forms/dpp.py:95-247``.  The reference returns UFL forms that Firedrake compiles; here
the "forms" are small descriptors of the same operators which the HIP library assembles:

    a((p1,p2),(q1,q2)) = (k1/mu) grad p1.grad q1 + (beta/mu)(p1-p2) q1          dpp.py:57
                       + (k2/mu) grad p2.grad q2 - (beta/mu)(p1-p2) q2          dpp.py:89
    L = 0                                                                       dpp.py:58,90,130

i.e.  A = [[a K + b M, -b M], [-b M, c K + b M]]  with a = k1/mu, b = beta/mu, c = k2/mu.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

from . import fd
from .parameters import DPPParameters


def _require_mixed(W) -> None:
    if not hasattr(W, "num_sub_spaces") or W.num_sub_spaces() != 2:
        raise ValueError(f"Expected a 2-field MixedFunctionSpace, got {type(W)}")


@dataclass(frozen=True)
class DPPBilinearForm:
    """Monolithic two-pressure operator on W (4 integrals, rank 2)."""
    space: fd.MixedFunctionSpace
    k1: float
    k2: float
    beta: float
    mu: float
    rank: int = 2
    num_integrals: int = 4

    @property
    def coefficients(self) -> Tuple[float, float, float]:
        return self.k1 / self.mu, self.beta / self.mu, self.k2 / self.mu


@dataclass(frozen=True)
class ZeroLinearForm:
    """L = 0 (zero forcing; the right-hand side comes from Dirichlet lifting only)."""
    space: object
    rank: int = 1


@dataclass(frozen=True)
class ScalarBlockForm:
    """One Picard block: (coef_K K + coef_M M) p_new = coef_M M p_old   (dpp.py:196-203)."""
    space: fd.FunctionSpace
    coef_K: float
    coef_M: float
    field: int
    delayed: Optional[fd.Function] = None
    rank: int = 2


@dataclass(frozen=True)
class DPPResidualForm:
    """Residual F(p1,p2) of the split formulation (dpp.py:240-246)."""
    space: fd.MixedFunctionSpace
    k1: float
    k2: float
    beta: float
    mu: float
    rank: int = 1


def dpp_form(W, model_params: DPPParameters) -> Tuple[DPPBilinearForm, ZeroLinearForm]:
    """(a, L) of the coupled system; raises ValueError unless W is a 2-field mixed space."""
    _require_mixed(W)
    a = DPPBilinearForm(W, float(model_params.k1), float(model_params.k2), float(model_params.beta),
                        float(model_params.mu))
    return a, ZeroLinearForm(W)


def dpp_delayed_form(macro_function_space, micro_function_space, model_params: DPPParameters,
                     macro_pressure_initial_values, micro_pressure_initial_values):
    """((a_macro, L_macro), (a_micro, L_micro)) with the other scale's pressure delayed."""
    k1, k2 = float(model_params.k1), float(model_params.k2)
    beta, mu = float(model_params.beta), float(model_params.mu)
    a_macro = ScalarBlockForm(macro_function_space, k1 / mu, beta / mu, 0)
    L_macro = ScalarBlockForm(macro_function_space, 0.0, beta / mu, 0, delayed=micro_pressure_initial_values, rank=1)
    a_micro = ScalarBlockForm(micro_function_space, k2 / mu, beta / mu, 1)
    L_micro = ScalarBlockForm(micro_function_space, 0.0, beta / mu, 1, delayed=macro_pressure_initial_values, rank=1)
    return (a_macro, L_macro), (a_micro, L_micro)


def dpp_splitted_form(W, model_params: DPPParameters) -> Tuple[DPPResidualForm, fd.Function]:
    """(F, fields) for the fixed-point (Picard) solve."""
    _require_mixed(W)
    fields = fd.Function(W)
    F = DPPResidualForm(W, float(model_params.k1), float(model_params.k2), float(model_params.beta),
                        float(model_params.mu))
    return F, fields
