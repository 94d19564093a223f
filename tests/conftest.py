import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sk25():
    from cheetah_pose_estimation_amd import skeleton
    return skeleton.build_skeleton("phantom", 25)


@pytest.fixture(scope="session")
def sk24():
    from cheetah_pose_estimation_amd import skeleton
    return skeleton.build_skeleton("phantom", 24)


@pytest.fixture(scope="session")
def cams6():
    from cheetah_pose_estimation_amd import synth
    return synth.make_cameras(6)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle: test infrastructure only (oracle/)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_handle_factory():
    """Builds product handles (HIP path through the C ABI); fails loudly when the library is missing."""
    from cheetah_pose_estimation_amd import _lib
    made = []

    def make(sk, cams, opts=None, priors=None):
        h = _lib.Handle(sk, cams, opts, priors)
        made.append(h)
        return h
    yield make
    for h in made:
        h.close()
