import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 GPU and the built libfriture_hip.so")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    return load


@pytest.fixture(scope="session")
def hip():
    """The initialised HIP library; errors (not skips) when the library or the GPU is missing."""
    from friture_amd import _lib
    return _lib.init()


@pytest.fixture
def option():
    """frt_set_option for the duration of a test (the library never reads the environment): option(name, value) forces one of
    two product code paths of an entry point; every option is back on its shape rule when the test ends."""
    from friture_amd import _lib
    touched = []

    def set_option(name, value):
        touched.append(name)
        _lib.set_option(name, value)
    yield set_option
    for name in touched:
        _lib.set_option(name, -1)


def rel_max(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def synth(kind: str, n: int, seed: int) -> np.ndarray:
    """Seeded synthetic audio of SURVEY.md §8d as float32 PCM."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    if kind == "noise":
        x = 0.25 * rng.standard_normal(n)
    elif kind == "tone":
        x = 0.5 * np.sin(2 * np.pi * 1000.0 * t / 48000.0) + 1e-3 * rng.standard_normal(n)
    elif kind == "chirp":
        dur = n / 48000.0
        k = np.log(20000.0 / 20.0) / dur
        x = 0.5 * np.sin(2 * np.pi * 20.0 * (np.exp(k * t / 48000.0) - 1.0) / k)
    else:
        raise ValueError(kind)
    return x.astype(np.float32)
