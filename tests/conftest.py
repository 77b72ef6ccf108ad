import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _reload_knobs():
    """the library reads its per-launch tile knobs (MI355_IGEMM8, MI355_IGEMM_BIG, MI355_STEM_DIRECT, ...) from the environment once:
    tests that flip them have the cache re-read (mi355_reload_knobs)"""
    try:
        from sota_imagenet_amd import native

        native.lib().mi355_reload_knobs()
    except Exception:
        pass  # (library not built: the tests that need it fail on their own)


@pytest.fixture(autouse=True)
def _fresh_knobs():
    _reload_knobs()  # the previous test's monkeypatch has been undone by now
    yield


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch whose setenv / delenv also refresh the library's knob cache"""
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set_(name, value, prepend)
        if name.startswith("MI355_"):
            _reload_knobs()

    def delenv(name, raising=True):
        del_(name, raising)
        if name.startswith("MI355_"):
            _reload_knobs()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
