#!/usr/bin/env python3
"""Generates tests/golden/* from the reference's OWN C codec, compiled in place from
/root/reference/ojph (oracle/_ref/libmic_ref.so, `make -C oracle ref`).  Run in the build
container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs (all data, no reference source):
  CT_512_512_image.bin, MR_256_256_image.bin   the two 16-bit test images the reference's
                                               own tests hold (fseu16_test.go:28-53)
  MR_256_256_{2,4,8}state.mic                  full reference C streams for MR
  golden.json                                  size / FNV-1a64 / head / tail of the reference
                                               C stream for 10 images x {2,4,8}-state
"""
import ctypes as C
import json
import os
import resource
import shutil
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import mico  # noqa: E402  (only for the FNV helper)

NEMA = {"CT1": (512, 512), "CT2": (512, 512), "MR1": (512, 512), "MR2": (1024, 1024),
        "MR3": (512, 512), "MR4": (512, 512), "NM1": (256, 1024), "XA1": (1024, 1024)}


def load_nema(name):
    """Pixel data of a NEMA *_UNC file without a DICOM library: last
    (7FE0,0010) tag, u32 length at +8, pixels at +12."""
    b = open(f"{REF}/testdata/compsamples_refanddir/IMAGES/REF/{name}_UNC", "rb").read()
    i = b.rfind(bytes([0xE0, 0x7F, 0x10, 0x00]))
    ln = int.from_bytes(b[i + 8:i + 12], "little")
    w, h = NEMA[name]
    return np.frombuffer(b[i + 12:i + 12 + ln], dtype="<u2")[: w * h].reshape(h, w).copy()


def main():
    mico.build(ref=True)
    R = C.CDLL(mico.REF_PATH)
    imgs = {"CT": np.fromfile(f"{REF}/testdata/CT_512_512_image.bin", dtype="<u2").reshape(512, 512),
            "MR": np.fromfile(f"{REF}/testdata/MR_256_256_image.bin", dtype="<u2").reshape(256, 256)}
    for k in NEMA:
        imgs[k] = load_nema(k)
    shutil.copy(f"{REF}/testdata/CT_512_512_image.bin", HERE)
    shutil.copy(f"{REF}/testdata/MR_256_256_image.bin", HERE)
    gold = {"generator": "tests/golden/make_golden.py", "source": "reference C codec ojph/mic_compress_c.c (oracle/_ref)",
            "streams": {}}
    fns = {2: "mic_compress_two_state", 4: "mic_compress_four_state", 8: "mic_compress_eight_state"}
    for name, img in imgs.items():
        h, w = img.shape
        img = np.ascontiguousarray(img)
        for ns, fn in fns.items():
            cap = img.size * 4 + 4096
            out = np.zeros(cap, np.uint8)
            n = C.c_size_t()
            rc = getattr(R, fn)(C.c_void_p(img.ctypes.data), w, h, C.c_void_p(out.ctypes.data), C.c_size_t(cap), C.byref(n))
            assert rc == 0, (name, ns, rc)
            blob = out[: n.value].tobytes()
            gold["streams"][f"{name}/{ns}"] = {
                "width": w, "height": h, "max_value": int(img.max()), "nstates": ns, "size": len(blob),
                "fnv1a64": f"{mico.fnv1a64(blob):016x}", "head": blob[:32].hex(), "tail": blob[-32:].hex(),
                "pixels_fnv1a64": f"{mico.fnv1a64(img.tobytes()):016x}"}
            if name == "MR":
                open(os.path.join(HERE, f"MR_256_256_{ns}state.mic"), "wb").write(blob)
    json.dump(gold, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(gold["streams"]), "stream records")


if __name__ == "__main__":
    # the reference C keeps ~1 MiB of tables on the stack
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    threading.stack_size(256 * 1024 * 1024)
    t = threading.Thread(target=main)
    t.start(); t.join()
