"""frame-level token mutations: GPU vs oracle"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import numpy as np
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
from oracle import mico
rng = np.random.default_rng(9)
tot = ok = bad = 0
for seed, noise, (h, w) in ((3, 2.0, (200, 320)), (4, 60.0, (130, 257)), (5, 10.0, (64, 96))):
    img = synth.xr_like(cols=w, rows=h, depth=12, seed=seed, noise=noise)
    tok = mico.delta_rle_compress(img, 4095)
    for k in range(60):
        t = tok.copy()
        for _ in range(int(rng.integers(1, 5))):
            i = int(rng.integers(1, t.size))
            t[i] = (0, 1, 2, int(t[0]), int(t[0]) // 2 + 1, int(rng.integers(0, int(t[0]) + 1)))[int(rng.integers(0, 6))]
        if k % 8 == 2: t = t[: int(rng.integers(3, t.size))]
        rc, stream = mico.fse_compress(t, 2)
        if rc: continue
        rc_o, want = mico.decompress_single_frame(stream, w, h)
        try:
            got = mic.decompress_single_frame(stream, w, h); rc_g = 0
        except mic.MicError as e:
            rc_g, got = e.code, None
        tot += 1; ok += rc_o == 0
        if (rc_g == 0) != (rc_o == 0) or (rc_o == 0 and not np.array_equal(got, want)):
            bad += 1
            print("MISMATCH seed", seed, "k", k, "oracle", rc_o, "gpu", rc_g, "zeros", int((t == 0).sum()), "ntok", t.size)
print("cases", tot, "oracle ok", ok, "mismatches", bad)
