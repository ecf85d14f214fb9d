import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import numpy as np
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
from oracle import mico
rng = np.random.default_rng(5)
img = synth.xr_like(cols=420, rows=300, depth=12, seed=3, noise=2.0)
n = img.size
rc, good = mico.wavelet_v2_compress(img, 4095, 4)
cases = [good[:k] for k in (11, 13, 20, len(good) // 2, len(good) - 1)]
for _ in range(40):
    b = bytearray(good)
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(11, len(b))); b[i] ^= 1 << int(rng.integers(0, 8))
    cases.append(bytes(b))
for ci, c in enumerate(cases):
    rc_o, want = mico.wavelet_v2_decompress(c)
    try:
        got, _, _ = mic.wavelet_v2_decompress(c); rc_g = 0
    except mic.MicError as e:
        rc_g, got = e.code, None
    flag = "" if (rc_g == 0) == (rc_o == 0) and (rc_o != 0 or np.array_equal(got, want)) else "  <<<<<< MISMATCH"
    if flag or rc_o == 0 or ci in (8, 11, 27):
        rcf, tk = (mico.fse_decompress_auto(c[11:], n * 6 + 32) if len(c) > 13 else (-1, None))
        print(ci, "oracle", rc_o, "gpu", rc_g, "fse", rcf, "ntok", None if tk is None else tk.size, "outlen", None if tk is None or tk.size < 3 else (int(tk[1]) << 16) + int(tk[2]),
              "zeros", None if tk is None else int((tk == 0).sum()), flag)
