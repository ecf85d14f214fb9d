"""Diagnostic: the bare FSE stage coded many times at several lengths -- counts the calls whose stream differs from the oracle's.
usage: python tools/dbg_fse1.py [flavours, e.g. 1,2] [rounds]"""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
mic = load_package()
from oracle import mico
synth = importlib.import_module("medical_image_codec_amd.synth")
tok = mico.delta_rle_compress(synth.xr_like(cols=500, rows=180, depth=12, seed=17), 4095)
fls = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for fl in fls:
    bad = {}; calls = 0
    for n in (tok.size, 80000, 65536, 60000, 49152, 40000, 20000):
        t = tok[:n].copy()
        rc, want = mico.fse_compress(t, fl)
        for r in range(rounds):
            got = mic.fse_compress_u16(t, fl); calls += 1
            if got != want:
                d = [i for i in range(min(len(got), len(want))) if got[i] != want[i]]
                bad.setdefault(n, []).append((len(d), d[0] if d else -1))
    print("flavour", fl, "calls", calls, "bad", sum(len(v) for v in bad.values()), {k: v[:3] for k, v in bad.items()})
