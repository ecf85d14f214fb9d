"""Diagnostic: the bare FSE stage coded many times at several lengths -- counts the calls whose stream differs from the oracle's,
says where the differences lie, and collects the status codes a TE_CHECK build (tools/te_variants.sh) reports.
usage: python tools/dbg_fse1.py [flavours, e.g. 1,2] [rounds]"""
import os, sys, importlib, collections
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
    bad = {}; calls = 0; codes = collections.Counter()
    for n in (tok.size, 80000, 65536, 60000, 49152, 40000, 20000):
        t = tok[:n].copy()
        rc, want = mico.fse_compress(t, fl)
        w = np.frombuffer(want, dtype=np.uint8)
        for r in range(rounds):
            calls += 1
            try:
                got = mic.fse_compress_u16(t, fl)
            except mic.MicError as e:
                codes[e.code] += 1
                continue
            if got != want:
                g = np.frombuffer(got, dtype=np.uint8); m = min(g.size, w.size)
                d = np.nonzero(g[:m] != w[:m])[0]
                # (n differing bytes, first, last, stream length): a thread's range is about len / 512 bytes
                bad.setdefault(n, []).append((int(d.size), int(d[0]) if d.size else -1, int(d[-1]) if d.size else -1, len(want), len(got)))
    print("flavour", fl, "calls", calls, "bad", sum(len(v) for v in bad.values()), "codes", dict(codes), {k: v[:3] for k, v in bad.items()}, flush=True)
