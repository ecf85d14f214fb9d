"""The C ABI from plain C (Go is absent here, so the boundary is exercised by a C driver): the driver
compiles against include/mic_hip.h and links libmic_hip.so with gcc; on a GPU box it runs a PICS round trip and a
batch round trip through host buffers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-image-codec_amd")


def _build(tmp_path):
    lib = os.path.join(PKG, "libmic_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libmic_hip.so not built")
    exe = str(tmp_path / "driver")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_driver", "driver.c"),
           "-L", PKG, "-lmic_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_driver_compiles_and_links(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_driver_round_trips(tmp_path, gpu_ready):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "round trip ok" in r.stdout
