import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def O():
    """The C oracle (test infrastructure; compiled on demand with gcc)."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def R():
    from oracle import np_restatement

    return np_restatement


@pytest.fixture(scope="session")
def golden():
    return {name[:-4]: np.load(os.path.join(GOLDEN, name)) for name in os.listdir(GOLDEN) if name.endswith(".npz")}


@pytest.fixture(scope="session")
def cameras(golden):
    return {k: golden["cameras"][k].tobytes() for k in golden["cameras"].files}


@pytest.fixture(scope="session")
def hip_built():
    """Build (if stale) and load the HIP C-ABI library; never falls back to anything else."""
    import __graft_entry__ as g

    g.build_hip()
    from vokselis_amd import _native

    return _native.lib()


def adversarial_volumes(n=32):
    """Same closed-form volumes as oracle/gen_golden.py (kept in sync by test_golden_volumes_rebuild)."""
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    vols = {
        "all0": np.zeros((n, n, n), np.uint8),
        "all25": np.full((n, n, n), 25, np.uint8),
        "all26": np.full((n, n, n), 26, np.uint8),
        "all255": np.full((n, n, n), 255, np.uint8),
        "impulse": np.zeros((n, n, n), np.uint8),
        "ramp_x": ((x * 255) // (n - 1)).astype(np.uint8),
        "ramp_y": ((y * 255) // (n - 1)).astype(np.uint8),
        "ramp_z": ((z * 255) // (n - 1)).astype(np.uint8),
        "checker": (((x + y + z) & 1) * 255).astype(np.uint8),
    }
    vols["impulse"][n // 2, n // 3, n // 4] = 255
    return vols


@pytest.fixture(scope="session")
def golden_volumes(O):
    vols = adversarial_volumes(32)
    vols["standin"] = O.volume_standin_u8(32)
    vols["fog"] = O.volume_fog_u8(32)
    return vols
