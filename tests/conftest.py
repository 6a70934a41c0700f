import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden2():
    """round-2 fixtures (tests/golden/make_golden_v2.py): mpmath-checked gradient / Matern prediction vectors, G6/G7"""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))


@pytest.fixture(scope="session")
def ref_inputs():
    """the reference's own example inputs (data files copied into tests/golden/ref_inputs)"""
    import numpy as np
    from madaiemulator_amd import synth
    d = os.path.join(ROOT, "tests", "golden", "ref_inputs")
    X1, Y1 = synth.read_input_model_file(os.path.join(d, "uni-simple.input_model_file.dat"))
    X2, Y2 = synth.read_input_model_file(os.path.join(d, "uni-2d-param.input_model_file.dat"))
    X3, Y3 = synth.read_input_model_file(os.path.join(d, "multi-simple.input_model_file.dat"))
    q1 = np.array(open(os.path.join(d, "uni-simple.sample_locations.dat")).read().split(), float).reshape(-1, 1)
    q2 = np.array(open(os.path.join(d, "uni-2d-param.sample_locations.dat")).read().split(), float).reshape(-1, 2)
    return dict(uni=(X1, Y1[:, 0]), twod=(X2, Y2[:, 0]), multi=(X3, Y3), q_uni=q1, q_2d=q2)


@pytest.fixture(scope="session")
def gpu_ctx():
    """one device context for the whole GPU session; fails loudly if the HIP library is missing"""
    from madaiemulator_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
