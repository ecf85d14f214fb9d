import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def load_package():
    """The package directory is named after the reference repo (medical-image-codec_amd); the
    hyphen makes it a non-identifier, so it is loaded by path under a legal module name."""
    name = "medical_image_codec_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "medical-image-codec_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def mic():
    return load_package()


@pytest.fixture(scope="session")
def synth():
    load_package()
    import importlib
    return importlib.import_module("medical_image_codec_amd.synth")


@pytest.fixture(scope="session")
def mico():
    from oracle import mico as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def gpu_ready(mic):
    name = mic.device_name()
    if not name:
        pytest.fail("libmic_hip.so found no usable gfx950 device (no CPU fallback exists)")
    return name
