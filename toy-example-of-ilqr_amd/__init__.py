"""MI355X-native batched CILQR solver (package directory: ``toy-example-of-ilqr_amd``; import it
as ``cilqr_amd`` through the shim at the repo root, or with importlib).

Scope: the CILQR solve path of PuYuuu/toy-example-of-iLQR — ``CILQRSolver::solve`` and everything
it calls — as hand-written HIP for gfx950 behind a C-ABI (include/cilqr_amd.h)."""
from . import _lib
from ._lib import (CilqrError, CilqrLibraryMissing, CilqrParams, END_NAMES, RESULT_DTYPE, STATUS_NAMES,
                   TRACE_DTYPE, library_info)
from .config import GlobalConfig, copy_params, params_from_config, params_to_dict
from .scenario import ReferenceLine, RoutingLine, Scenario, build_scenario
from .solver import BatchedCILQR, CILQRSolver, SceneTable
from . import workloads

__all__ = ["BatchedCILQR", "CILQRSolver", "SceneTable", "GlobalConfig", "params_from_config", "copy_params",
           "params_to_dict", "ReferenceLine", "RoutingLine", "Scenario", "build_scenario", "workloads",
           "CilqrParams", "CilqrError", "CilqrLibraryMissing", "RESULT_DTYPE", "TRACE_DTYPE", "STATUS_NAMES",
           "END_NAMES", "library_info"]
