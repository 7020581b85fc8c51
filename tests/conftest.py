import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A GPU test that stops making progress must fail by name, not wedge the run: ten minutes per test
    # (the slowest, full-size C5 against its golden tiles, takes about ten seconds) unless the command
    # line sets its own --timeout.  pytest-timeout is in the image; without it the marker is inert.
    if not config.pluginmanager.hasplugin("timeout") or getattr(config.option, "timeout", None):
        return
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/rt_oracle.h)."""
    from oracle import rt_oracle_py
    rt_oracle_py.lib()
    return rt_oracle_py


@pytest.fixture(scope="session")
def constant_sky():
    import compute_raytracer_amd as rt
    from compute_raytracer_amd.scene_raytracing import CONSTANT_SKY_RGBA
    return rt.CubemapMaterial.constant(CONSTANT_SKY_RGBA)
