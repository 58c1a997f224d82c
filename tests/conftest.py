import importlib
import os
import sys

import pytest
import torch  # noqa: F401  -- first: torch ships its own HIP runtime; a process that loads /opt/rocm's copy (through
#                             librt_mi355.so) before torch's ends with two runtimes and torch then finds no GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def rt():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def scene_path():
    return lambda name: os.path.join(ROOT, "scenes", name + ".npz")


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    """One context for the whole GPU session (a fresh box pays the HIP init once)."""
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
