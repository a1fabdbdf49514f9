import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def vga_seq():
    """Three 640x480 BGR frames of one synthetic sequence (seed 1001)."""
    from visual_odometry_amd import synth
    return synth.make_sequence(synth.CONFIG_CAMERAS["vga"], 3, seed=1001, channels=3)


@pytest.fixture(scope="session")
def ctx():
    from visual_odometry_amd import capi
    c = capi.Context(device=0, max_width=1920, max_height=1080, max_features=4000, max_batch=8)
    yield c
    c.close()
