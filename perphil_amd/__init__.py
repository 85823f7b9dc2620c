"""
perphil_amd — MI355X-native implementation of perphil's DPP hot path (CG-1 assembly of the
two-pressure system + Krylov / Picard solve) behind perphil's own Python surface.

reference module                              ->  here
``perphil.mesh.builtin.create_mesh``              ``perphil_amd.create_mesh``
``perphil.forms.spaces.create_function_spaces``   ``perphil_amd.create_function_spaces``
``perphil.forms.dpp.dpp_form`` (+delayed/split)   ``perphil_amd.forms``
``perphil.models.dpp.parameters.DPPParameters``   ``perphil_amd.DPPParameters``
``perphil.solvers.solver.solve_dpp`` / Solution   ``perphil_amd.solve_dpp`` / ``Solution``
``perphil.solvers.parameters``                    ``perphil_amd.solver_parameters``
``perphil.utils.manufactured_solutions``          ``perphil_amd.manufactured_solutions``
``firedrake`` objects crossing the boundary       ``perphil_amd.fd``

Importing the package does not touch the GPU; the HIP library is loaded on first use of
``perphil_amd.solver`` / ``perphil_amd._ffi`` and there is no CPU fallback.
"""
from . import fd, solver_parameters  # noqa: F401
from .parameters import DPPParameters  # noqa: F401
from .mesh import create_mesh  # noqa: F401
from .spaces import create_function_spaces  # noqa: F401
from .forms import dpp_form, dpp_delayed_form, dpp_splitted_form  # noqa: F401
from .manufactured_solutions import exact_expressions, exact_expressions_3d, interpolate_exact  # noqa: F401

__all__ = [
    "fd", "solver_parameters", "DPPParameters", "create_mesh", "create_function_spaces", "dpp_form",
    "dpp_delayed_form", "dpp_splitted_form", "exact_expressions", "exact_expressions_3d", "interpolate_exact",
    "solve_dpp", "solve_dpp_nonlinear", "Solution",
]


def __getattr__(name):
    # solver pulls in the HIP library: import lazily so that `import perphil_amd` works for the
    # host-only pieces (option dictionaries, parameters) exactly like `import perphil` does
    # without Firedrake (reference src/perphil/__init__.py:1-16)
    if name in ("solve_dpp", "solve_dpp_nonlinear", "Solution", "solver", "_ffi"):
        import importlib

        mod = importlib.import_module(".solver" if name != "_ffi" else "._ffi", __name__)
        return mod if name in ("solver", "_ffi") else getattr(mod, name)
    raise AttributeError(name)
