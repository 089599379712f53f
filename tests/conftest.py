import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # DRFE_TEST_LIB=<path>: run the suite against a variant build of libdrfe.so (kernel experiments)
    if os.environ.get("DRFE_TEST_LIB"):
        import dr_slam_amd.lib as _L
        _L.LIB_PATH = os.path.abspath(os.environ["DRFE_TEST_LIB"])
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def frames_room():
    """Four consecutive 640x480 frames of the seed-2 room_boxes sequence."""
    from dr_slam_amd import synth
    return list(synth.sequence(2, 4))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.lib()
    return orc
