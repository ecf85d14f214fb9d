"""Diagnostic (build with EXTRA_FLAGS=-DMIC_STAMP): per-phase ticks of the encode kernels on many SMALL units (256 x 256, 8-bit:
the shape of a MIC3 plane)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H = 256, 2048
img = synth.xr_like(cols=W, rows=H, depth=8, seed=1, noise=2.0)
F = int(os.environ.get("FRAMES", "1024"))
d_px = torch.from_numpy(np.stack([img] * F).view(np.int16)).cuda()
units = [(f * W * H + y0 * W, W, 256, 255, 2) for f in range(F) for y0 in range(0, H, 256)]
sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
for _ in range(2):
    sess.set_timing(1)
    sess.encode_enqueue(d_px.data_ptr(), cu); t = sess.last_timings(); d_blobs, offs, st, ns = sess.encode_finish(); assert (st == 0).all()
print({k: round(v, 3) for k, v in t if v > 0.02}, "ratio", d_px.numel() * 2 / int(offs[-1]))
buf = (C.c_uint32 * 32)()
names = ["tok.A symbols", "tok.B facts", "tok.C counts", "tok.D write", "tok.E carry", "tab.normalise", "tab.ncount", "tab.ctable",
         "tans.walk", "tans.fixup", "tans.bits", "tans.pack", "fix.rounds", "fix.rewalkers", "fix.maxgroups", "tok.slow_wavetiles"]
for i in (0, 1, len(units) // 2, len(units) - 1):
    mic.lib().mic_hip_debug_unit(sess._h, i, buf)
    print(f"unit {i}: ntok={buf[0]} tl={buf[2]} " + " ".join(f"{n}={buf[16 + k]}" for k, n in enumerate(names)))
