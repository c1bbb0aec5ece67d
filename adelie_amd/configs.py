"""Global numeric knobs — mirrors ``adelie.configs`` (reference ``adelie/configs.py:4-27``,
``adelie_core/configs.hpp:6-21``, ``py_configs.cpp:6-49``).  Only the two constants that change
numerics on the hot path are forwarded to the native library."""
from . import _abi


class Configs:
    hessian_min_def = 1e-24
    dbeta_tol_def = 1e-12
    min_bytes_def = 1 << 17
    project_def = True
    max_solver_value_def = 1e100
    pb_symbol_def = "\033[1;32m█\033[0m"

    hessian_min = hessian_min_def
    dbeta_tol = dbeta_tol_def
    min_bytes = min_bytes_def
    project = project_def
    max_solver_value = max_solver_value_def
    pb_symbol = pb_symbol_def


def set_configs(name: str, value=None):
    """Sets a configuration; ``value=None`` restores the default (reference ``configs.py:4-27``)."""
    if not hasattr(Configs, name + "_def"):
        raise RuntimeError(f"unknown config: {name}")
    if value is None:
        value = getattr(Configs, name + "_def")
    setattr(Configs, name, value)
    if name in ("hessian_min", "dbeta_tol"):
        b = _abi.hip_backend()
        b.check(b.fn("set_config")(name.encode(), float(value)))
