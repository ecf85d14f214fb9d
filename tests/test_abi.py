"""CPU tests of the drop-in boundary: libmic_hip.so loads without a GPU and exports every
symbol include/mic_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mic_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_in_tree(mic):
    assert os.path.exists(mic.LIB_PATH), "run __graft_entry__.build()"


def test_every_declared_symbol_is_exported(mic):
    L = ctypes.CDLL(mic.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_covers_header(mic):
    assert sorted(mic.ABI_SYMBOLS) == _declared()


def test_status_codes_match_header(mic):
    src = open(os.path.join(ROOT, "include", "mic_hip.h")).read()
    for name, val in re.findall(r"#define\s+(MIC_(?:OK|ERR_[A-Z_]+))\s+(-?\d+)", src):
        assert getattr(mic, name) == int(val), name
    dev = open(os.path.join(ROOT, "medical-image-codec_amd", "csrc", "mic_dev.h")).read()
    for name, val in re.findall(r"#define\s+MICD_(OK|ERR_[A-Z_]+)\s+(-?\d+)", dev):
        assert getattr(mic, "MIC_" + name) == int(val), name


def test_product_package_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "medical-image-codec_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "mic_oracle" not in txt and "libmic_oracle" not in txt and "from oracle" not in txt, os.path.join(d, f)


def test_version_string_without_gpu(mic):
    assert b"mic-hip" in mic.lib().mic_hip_version()
