"""Debug: bare FSE decode through the lane-per-state kernels against the oracle's token streams."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
mic = entry.load_package()
from oracle import mico
mico.lib()
img = np.fromfile(os.path.join(ROOT, "tests/golden/MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
tok = mico.delta_rle_compress(img, int(img.max()))
for ns in (2, 4, 8, 108):
    for n in (len(tok),):
        t = tok[:n]
        rc, blob = mico.fse_compress(t, ns)
        if rc:
            print(ns, n, "oracle rc", rc); continue
        try:
            got = mic.fse_decompress_u16_auto(blob, n + 64)
        except Exception as e:
            print(ns, n, "ERR", e); continue
        if len(got) != n or not np.array_equal(got, t):
            bad = np.nonzero(got[:min(n, len(got))] != t[:min(n, len(got))])[0]
            print(ns, n, "MISMATCH len", len(got), "first bad", bad[:8], "tl?", blob[6] & 15 if ns != 1 else None)
            print(" got ", got[32:64]); print(" want", t[32:64])
        else:
            print(ns, n, "ok")
