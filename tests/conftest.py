import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "not_yet_run_on_hardware: GPU tests of round 6 whose kernels were only ever run on the CPU emulator "
                            "(tests/emu; the GPU pool was closed): collected LAST, so that `pytest -m gpu -x` has proven every test that "
                            "has hardware history before it reaches them")


def pytest_collection_modifyitems(config, items):
    last = [it for it in items if it.get_closest_marker("not_yet_run_on_hardware")]
    if last:
        items[:] = [it for it in items if not it.get_closest_marker("not_yet_run_on_hardware")] + last


@pytest.fixture(scope="session")
def built():
    """Build the native pieces once per session (hipcc cross-compiles without a GPU)."""
    import importlib
    build = importlib.import_module("toy-example-of-ilqr_amd.build")
    build.build_all()
    import oracle
    oracle.ensure_built()
    return True


@pytest.fixture(scope="session")
def pkg(built):
    import cilqr_amd
    return cilqr_amd


@pytest.fixture(scope="session")
def orc_det(built):
    from oracle import Oracle
    return Oracle("det")


@pytest.fixture(scope="session")
def orc_libm(built):
    from oracle import Oracle
    return Oracle("libm")


def oracle_scene(sc, tick=0):
    """oracle.Scene from a package Scenario (same arrays for both sides)."""
    from oracle import Scene
    return Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, tick)


@pytest.fixture(scope="session")
def scenarios(pkg):
    out = {}
    for name in ("two_straight", "two_borrow", "three_straight", "three_bend"):
        cfg = pkg.GlobalConfig.get_instance(name)
        out[name] = (cfg, pkg.build_scenario(cfg, name))
    return out


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def reference_layout_yaml(flat):
    """A scenario in the layout of the reference's config/*.yaml (block maps, inline lists, "- [..]" rows, comments),
    re-created from one of our flattened scenario files — what include/cilqr_config.hpp's YAML reader must handle."""
    nested = {"max_simulation_time": flat["max_simulation_time"], "delta_t": flat["delta_t"], "lqr": {}, "iteration": {},
              "vehicle": {}, "laneline": {"reference": {}}, "visualization": {}}
    for k, v in flat.items():
        parts = k.split("/")
        if len(parts) == 2 and parts[0] in nested:
            nested[parts[0]][parts[1]] = v
        elif len(parts) == 3:
            nested["laneline"]["reference"][parts[2]] = v
    nested["initial_condition"] = flat["initial_condition"]

    def scalar(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return f'"{v}"'
        return repr(v)

    def emit(d, ind=0):
        out = []
        for k, v in d.items():
            pad = " " * ind
            if isinstance(v, dict):
                out.append(f"{pad}{k}:   # section")
                out += emit(v, ind + 2)
            elif isinstance(v, list) and v and isinstance(v[0], list):
                out.append(f"{pad}{k}:")
                out.append(f"{pad}  # [x, y, v, yaw]")
                out += [f"{pad}  - [{', '.join(repr(e) for e in row)}]" for row in v]
            elif isinstance(v, list):
                out.append(f"{pad}{k}: [{', '.join(repr(e) for e in v)}]")
            else:
                out.append(f"{pad}{k}: {scalar(v)}")
        return out

    return "\n".join(["# generated for the test"] + emit(nested)) + "\n"
