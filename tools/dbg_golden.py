"""Diagnostic: the golden MR / CT frames coded several times in one process -- a stream that changes from call to call is a race."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
mic = load_package()
from oracle import mico
G = os.path.join(ROOT, "tests", "golden")
gold = json.load(open(os.path.join(G, "golden.json")))
imgs = {"MR": np.fromfile(os.path.join(G, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256),
        "CT": np.fromfile(os.path.join(G, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)}
for name, img in imgs.items():
    for ns in (2, 4, 8):
        rec = gold["streams"][f"{name}/{ns}"]
        hs = []
        for rep in range(6):
            blob = mic.compress_single_frame(img, rec["width"], rec["height"], rec["max_value"], ns)
            hs.append(f"{mico.fnv1a64(blob):016x}")
        rc, want = mico.compress_single_frame(img, rec["max_value"], ns)
        diff = [i for i in range(min(len(blob), len(want))) if blob[i] != want[i]]
        print(name, ns, "want", rec["fnv1a64"], "got", sorted(set(hs)), "ok" if set(hs) == {rec["fnv1a64"]} else "BAD", "first diffs", diff[:6], "n diff", len(diff), "len", len(blob))
import importlib
synth = importlib.import_module("medical_image_codec_amd.synth")
tok = mico.delta_rle_compress(synth.xr_like(cols=500, rows=180, depth=12, seed=17), 4095)
for fl in (1, 2, 4, 8, 108):
    rc, want = mico.fse_compress(tok, fl)
    bad = 0
    for rep in range(8):
        bad += mic.fse_compress_u16(tok, fl) != want
    print("fse stage", fl, "bad", bad, "of 8")
for depth, nm in ((16, "ct16"), (12, "xr12")):
    img = synth.xr_like(cols=700, rows=300, depth=depth, seed=9)
    for ns in (2, 4, 8):
        rc, want = mico.compress_single_frame(img, (1 << depth) - 1, ns)
        bad = sum(mic.compress_single_frame(img, 700, 300, (1 << depth) - 1, ns) != want for _ in range(6))
        print(nm, ns, "bad", bad, "of 6")
